"""Serving engine of the model worker: what `modelcompose/serve/model_worker.py:123-194` asks of the model -
`generate_stream(params)`: tokenise the prompt with its placeholders, run generation with a streamer and a stop string, yield the growing
text as NUL-terminated JSON chunks - on top of a CONTINUOUS-BATCHING scheduler instead of one `model.generate` thread per request.

The reference serialises requests behind a semaphore (`--limit-model-concurrency`, model_worker.py:214-225): each request owns the GPU for
its whole generation.  Decoding is HBM-bound - a step streams the 13 GB of weights whatever the number of rows - so here every request owns
one ROW of a fixed-size decode batch instead:

  * `ContinuousBatcher` keeps one KV cache [layer][max_batch][Hkv][Smax][D] and the per-row decode state (token to feed, cached length);
  * at every iteration it (1) admits waiting requests into free rows: the request's prompt is prefilled as its own launch sequence
    (encoders, splice, routed LocalLoRA prefill - the path `generate()` takes) into a private cache whose keys are then copied into the
    row, (2) runs ONE decode step of the runtime for all rows (rows without a request idle on a one-token context), (3) hands every
    active row's new token to its request (streamer / stop criteria / EOS / max_new_tokens) and retires finished rows;
  * requests therefore join and leave at token granularity; a long generation never blocks a short one.

Per-row results equal `model.generate(...)` of that request alone in the same decode-kernel class (tests/test_serve_gpu.py).
The HTTP shell (FastAPI app, controller registration, heart beat: model_worker.py:33-121, 196-260) is control plane and out of scope;
`ModelWorker` mirrors the two methods the shell calls, `generate_stream` and `get_status`."""
from __future__ import annotations

import json
import queue
import threading
import time
from dataclasses import dataclass, field
from typing import Callable, Dict, Iterator, List, Optional

import numpy as np
import torch

from .. import _lib
from ..constants import DEFAULT_IMAGE_TOKEN, IMAGE_TOKEN_INDEX

server_error_msg = "**NETWORK ERROR DUE TO HIGH TRAFFIC. PLEASE REGENERATE OR REFRESH THIS PAGE.**"      # modelcompose/utils.py:11


@dataclass
class GenerationRequest:
    input_ids: torch.Tensor                              # (1, L_text) with sentinels
    modal_inputs: Optional[dict] = None
    max_new_tokens: int = 256
    do_sample: bool = False
    temperature: float = 1.0
    top_p: float = 1.0
    top_k: int = 50
    seed: Optional[int] = None
    stopping_criteria: Optional[list] = None             # callables (ids (1, n), scores) -> bool, transformers style
    on_token: Optional[Callable[[int], None]] = None     # called with every new token id (the streamer hook)
    # filled by the scheduler
    new_ids: List[int] = field(default_factory=list)
    finished: threading.Event = field(default_factory=threading.Event)
    error: Optional[BaseException] = None
    t_submit: float = 0.0
    t_first: float = 0.0
    t_done: float = 0.0


class ContinuousBatcher:
    """Iteration-level scheduler over `max_batch` decode rows of one model (see the module docstring)."""

    def __init__(self, model, max_batch: int = 16, max_seq_len: Optional[int] = None):
        self.model, self.B = model, int(max_batch)
        cfg = model.config
        self.Smax = ((max_seq_len or cfg.max_position_embeddings) + 63) // 64 * 64
        dev = model.device
        shape = (cfg.num_hidden_layers, self.B, cfg.num_key_value_heads, self.Smax, cfg.head_dim)
        with torch.inference_mode(False):
            self.kc = torch.zeros(shape, dtype=_lib.storage_dtype(), device=dev)
            self.vc = torch.zeros(shape, dtype=_lib.storage_dtype(), device=dev)
            self.kv_lens = torch.ones(self.B, dtype=torch.int32, device=dev)          # idle rows: a one-token context
            self.next_ids = torch.zeros(self.B, dtype=torch.int64, device=dev)
            self.out = torch.zeros(self.B, 1, dtype=torch.int64, device=dev)
        self.lens = np.ones(self.B, dtype=np.int64)                                    # host mirror of kv_lens
        self.rows: List[Optional[GenerationRequest]] = [None] * self.B
        self.waiting: "queue.Queue[GenerationRequest]" = queue.Queue()
        self.slot = model._new_slot()
        self._ws = None
        self._thread: Optional[threading.Thread] = None
        self._stop = threading.Event()
        self.steps = 0
        self.tokens_out = 0
        self.dead: Optional[BaseException] = None                    # set when the background loop has died

    # ------------------------------------------------------------------ client side
    def submit(self, req: GenerationRequest) -> GenerationRequest:
        if self.dead is not None:
            raise RuntimeError(f"the batching engine has stopped: {self.dead!r}")
        req.t_submit = time.perf_counter()
        self.waiting.put(req)
        return req

    def queue_length(self) -> int:
        return self.waiting.qsize() + sum(r is not None for r in self.rows)

    # ------------------------------------------------------------------ scheduler
    def _admit(self, row: int, req: GenerationRequest):
        """Prefill `req` alone (the path generate() takes for its prompt) and move its keys into cache row `row`."""
        m = self.model
        ids = req.input_ids.to(m.device)
        mi = req.modal_inputs or {}
        feats, _ = m.encode_modal_inputs(mi, m.prefix_tokens, m.suffix_tokens)
        plan = m._plan(ids, None, None, mi, feats)
        L = int(plan.valid_lens[0])
        if L + req.max_new_tokens > self.Smax:
            raise ValueError(f"prompt of {L} tokens + {req.max_new_tokens} new tokens exceeds the worker's context of {self.Smax}")
        sampling = None
        if req.do_sample:
            seed = req.seed if req.seed is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
            sampling = (float(req.temperature), int(req.top_k or 0), float(req.top_p), int(seed))
        # the admission cache is sized ONCE, at the worker's context (reserve = Smax - L): a prompt whose ceil64(L) differs from the last one's
        # does not reallocate and zero a full-depth KV cache
        reserve = max(0, min(self.Smax, m.config.max_position_embeddings) - int(plan.Lmax))
        st = m._prefill(plan, feats, reserve, want_logits=True, slot=("adm", self.slot))
        if sampling is not None:
            from .. import ops
            first = ops.sample_step(st["logits"], sampling[0], sampling[1], sampling[2], seed=sampling[3], step=-1)
        else:
            first = st["next_ids"]
        # the prefill's cache is [layer][1][Hkv][Smax'][D]: copy the L cached keys into the batch cache's row
        self.kc[:, row, :, :L].copy_(st["kc"][:, 0, :, :L])
        self.vc[:, row, :, :L].copy_(st["vc"][:, 0, :, :L])
        self.next_ids[row:row + 1].copy_(first.reshape(1))
        self.kv_lens[row:row + 1].fill_(L)
        self.lens[row] = L
        req._sampling = sampling
        req._prompt_len = int(ids.shape[1])
        self.rows[row] = req
        self._emit(row, int(first.reshape(1).item()))

    def _emit(self, row: int, tok: int):
        req = self.rows[row]
        req.new_ids.append(tok)
        if not req.t_first:
            req.t_first = time.perf_counter()
        self.tokens_out += 1
        if req.on_token is not None:
            req.on_token(tok)
        done = tok == self.model.config.eos_token_id or len(req.new_ids) >= req.max_new_tokens
        if not done and req.stopping_criteria:
            seq = torch.cat([req.input_ids.reshape(1, -1).cpu(), torch.tensor([req.new_ids], dtype=torch.int64)], 1)
            done = any(bool(c(seq, None)) for c in req.stopping_criteria)
        if done:
            self._retire(row)

    def _retire(self, row: int, error: Optional[BaseException] = None):
        req = self.rows[row]
        self.rows[row] = None
        self.kv_lens[row:row + 1].fill_(1)
        self.lens[row] = 1
        self.next_ids[row:row + 1].zero_()
        if req is not None:
            req.error = error
            req.t_done = time.perf_counter()
            req.finished.set()

    def step(self) -> int:
        """One scheduler iteration; returns the number of active rows after it."""
        m = self.model
        # (1) admissions
        for row in range(self.B):
            if self.rows[row] is None and not self.waiting.empty():
                try:
                    req = self.waiting.get_nowait()
                except queue.Empty:
                    break
                try:
                    with m._lock:                                 # the model's one-shot handle state is ours for the whole admission
                        self._admit(row, req)
                except Exception as e:                            # a bad request must not take the engine down (KeyboardInterrupt / SystemExit do)
                    self.rows[row] = req
                    self._retire(row, e)
        active = [r for r in range(self.B) if self.rows[r] is not None]
        if not active:
            return 0
        # (2) one decode step for every row: the token in next_ids[row] is appended at position lens[row]
        st = self._state()
        sampled = [r for r in active if getattr(self.rows[r], "_sampling", None) is not None]
        with m._lock:
            lg = m._decode(st, 1, self.out, 0, want_logits=bool(sampled))
        self.kv_lens += 1                                           # every row advanced (idle rows are reset below)
        self.lens += 1
        toks = self.out[:, 0].clone()
        if sampled:
            from .. import ops
            for r in sampled:                                       # per-request sampling parameters: one small launch per sampled row
                T, K, P, seed = self.rows[r]._sampling
                toks[r:r + 1] = ops.sample_step(lg[0, r:r + 1], T, K, P, seed=seed, step=len(self.rows[r].new_ids) - 1)
            self.next_ids.copy_(toks)
        toks = toks.cpu().tolist()
        self.steps += 1
        for r in range(self.B):
            if self.rows[r] is None:
                self.kv_lens[r:r + 1].fill_(1)
                self.lens[r] = 1
        # (3) deliver
        for r in active:
            try:
                self._emit(r, int(toks[r]))
            except Exception as e:                                # a request's own callback / stopping criterion failed: retire that row only
                if self.rows[r] is not None:
                    self._retire(r, e)
        return sum(r is not None for r in self.rows)

    def _state(self):
        """The dict _decode() reads (what _prefill() returns for a batch), over the shared cache and the current row lengths."""
        m = self.model
        if self._ws is None:
            import ctypes as C
            from .. import _lib
            nbytes = C.c_int64(0)
            _lib.check(_lib.lib().mc_llm_workspace_bytes(m._handle, self.B, self.B, 1, C.byref(nbytes)), "mc_llm_workspace_bytes")
            with torch.inference_mode(False):
                self._ws = torch.empty(nbytes.value, dtype=torch.uint8, device=m.device)

        class _P:                                                   # the two fields of a SplicePlan that _decode() reads
            pass
        p = _P()
        p.B, p.valid_lens = self.B, self.lens.copy()
        return {"plan": p, "kv_lens": self.kv_lens, "kc": self.kc, "vc": self.vc, "Smax": self.Smax, "ws": self._ws,
                "next_ids": self.next_ids, "slot": self.slot}

    # ------------------------------------------------------------------ background loop
    def start(self):
        if self._thread is None:
            self._stop.clear()
            self._thread = threading.Thread(target=self._loop, name="mc-batcher", daemon=True)
            self._thread.start()
        return self

    def shutdown(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(30)
            if not self._thread.is_alive():                       # a loop still inside a long step keeps its handle: start() must not spawn a second one
                self._thread = None

    def close(self):
        """shutdown() + give the model back the engine's buffer slots (decode state, admission KV cache)."""
        self.shutdown()
        if self._thread is None:
            self.model._release_slot(self.slot)

    def _loop(self):
        try:
            dev = self.model.device
            torch.cuda.set_device(dev.index if dev.index is not None else torch.cuda.current_device())
            with torch.no_grad():
                while not self._stop.is_set():
                    if self.step() == 0 and self.waiting.empty():
                        time.sleep(0.001)
        except BaseException as e:                                   # the engine itself failed: no request may wait for it forever
            self.dead = e
            for row in range(self.B):
                if self.rows[row] is not None:
                    self._fail(self.rows[row], e)
                    self.rows[row] = None
            while not self.waiting.empty():
                try:
                    self._fail(self.waiting.get_nowait(), e)
                except queue.Empty:
                    break

    @staticmethod
    def _fail(req, e):
        req.error = e
        req.t_done = time.perf_counter()
        req.finished.set()

    def run_until_idle(self, max_iters: int = 1 << 30):
        """Synchronous driver (tests, offline use): iterate until no request is waiting or active."""
        with torch.no_grad():
            for _ in range(max_iters):
                if self.step() == 0 and self.waiting.empty():
                    return


class ModelWorker:
    """The model-facing half of modelcompose/serve/model_worker.py's ModelWorker: `generate_stream(params)` with the reference's
    parameter names and chunk format (:123-194), `get_status()` (:116-121).  `tokenizer` must offer what the reference uses of it:
    __call__(text).input_ids, decode(ids, skip_special_tokens=True), bos_token_id."""

    def __init__(self, model, tokenizer, image_processor=None, model_name="modelcompose-hip", max_batch=16, max_seq_len=None, start=True):
        self.model, self.tokenizer, self.image_processor, self.model_name = model, tokenizer, image_processor, model_name
        self.is_multimodal = True
        self.engine = ContinuousBatcher(model, max_batch=max_batch, max_seq_len=max_seq_len)
        if start:
            self.engine.start()

    def get_queue_length(self):
        return self.engine.queue_length()

    def get_status(self):
        return {"model_names": [self.model_name], "speed": 1, "queue_length": self.get_queue_length()}

    def generate_stream(self, params: Dict) -> Iterator[bytes]:
        from ..mm_utils import KeywordsStoppingCriteria, process_images, tokenizer_image_token
        tokenizer, model = self.tokenizer, self.model
        prompt = params["prompt"]
        ori_prompt = prompt
        images = params.get("images", None)
        modal_inputs = {}
        num_image_tokens = 0
        if images is not None and len(images) > 0 and self.is_multimodal:
            if len(images) != prompt.count(DEFAULT_IMAGE_TOKEN):
                raise ValueError("Number of images does not match number of <image> tokens in prompt")
            if isinstance(images, torch.Tensor):
                pixels = images
            else:
                pixels = process_images(images, self.image_processor, model.config)
            if isinstance(pixels, list):
                pixels = torch.stack(pixels, 0)
            modal_inputs["vision"] = pixels.to(model.device, dtype=_lib.storage_dtype())
            enc = model.get_model().get_modal_encoder("vision")
            num_image_tokens = prompt.count(DEFAULT_IMAGE_TOKEN) * int(getattr(enc, "num_patches", 0))
        temperature = float(params.get("temperature", 1.0))
        top_p = float(params.get("top_p", 1.0))
        max_context_length = getattr(model.config, "max_position_embeddings", 2048)
        max_new_tokens = min(int(params.get("max_new_tokens", 256)), 1024)
        stop_str = params.get("stop", None)
        do_sample = temperature > 0.001
        input_ids = tokenizer_image_token(prompt, tokenizer, IMAGE_TOKEN_INDEX, return_tensors="pt").unsqueeze(0)
        criteria = [KeywordsStoppingCriteria([stop_str], tokenizer, input_ids)] if stop_str else None
        max_new_tokens = min(max_new_tokens, max_context_length - input_ids.shape[-1] - num_image_tokens)
        if max_new_tokens < 1:
            yield json.dumps({"text": ori_prompt + "Exceeds max token length. Please start a new conversation, thanks.", "error_code": 0}).encode() + b"\0"
            return
        q: "queue.Queue[Optional[int]]" = queue.Queue()
        req = GenerationRequest(input_ids=input_ids, modal_inputs=modal_inputs, max_new_tokens=max_new_tokens, do_sample=do_sample,
                                temperature=temperature, top_p=top_p, seed=params.get("seed"), stopping_criteria=criteria, on_token=q.put)
        self.engine.submit(req)
        generated, ids = ori_prompt, []
        eos = model.config.eos_token_id
        while True:
            try:
                tok = q.get(timeout=0.05)
            except queue.Empty:
                if req.finished.is_set() and q.empty():
                    break
                if self.engine.dead is not None:
                    raise RuntimeError(f"the batching engine has stopped: {self.engine.dead!r}")
                continue
            if tok != eos:
                ids.append(tok)
            text = tokenizer.decode(ids, skip_special_tokens=True)
            generated = ori_prompt + text
            if stop_str and generated.endswith(stop_str):
                generated = generated[:-len(stop_str)]
            yield json.dumps({"text": generated, "error_code": 0}).encode() + b"\0"
        if req.error is not None:
            raise req.error

    def generate_stream_gate(self, params):
        try:
            for x in self.generate_stream(params):
                yield x
        except Exception as e:                                       # model_worker.py:180-194: every failure becomes an error chunk
            print("Caught Error", e)
            yield json.dumps({"text": server_error_msg, "error_code": 1}).encode() + b"\0"
