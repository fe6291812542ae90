"""One stage-2 finetune step (forward + backward + gradient all-reduce + AdamW) of the LocalLoRA model on the HIP path.

Host-side mirror of what `modelcompose/train/train_multimodal.py` asks HF Trainer / DeepSpeed / autograd to do for one batch:
  forward   MultimodalLlamaForCausalLM.forward with labels (multimodal_llama.py:676-745): splice -> 32 x decoder layer in the
            branch form y = x W^T + s_a B_a (A_a x) with the adapter chosen per token (:120-160, :262-268) -> lm_head ->
            shifted CrossEntropyLoss (:722-733)
  backward  gradients of the trainable set of train_multimodal.py:436-465 (lora_strategy 'modal+language': every lora_A/lora_B,
            the modal projectors, prefix/suffix tokens); base weights, norms, embeddings, lm_head and encoders are frozen
  exchange  data-parallel gradient all-reduce (the reference: DeepSpeed ZeRO-2/3 over NCCL; here torch.distributed = RCCL over
            xGMI, bucketed and overlapped with the backward pass), then AdamW on fp32 master weights.
Every tensor op is a kernel of libmc_hip.so; torch provides memory, streams and the collective.

Restructuring that keeps the function:
  * rows stay in sequence order (row = b*L + t); the per-token adapter mask-sum (:262-268) is applied to the rank-r activations:
    T = x [A_0; A_1; ..]^T  ([M, n_adapters*r]), row m keeps the r columns of its adapter (zero elsewhere), y += s T [B_0 | B_1 | ..]^T.
    One pair of rank-r GEMMs serves all adapters, and the same mask routes the gradients.
  * nothing is recomputed: activations of all layers stay resident (≈0.4 GB per layer at B=4, L=682 — the reference checkpoints
    per layer, multimodal_llama.py:567-583, because it targets 80 GB parts).
  * lora_dropout (nn.Dropout on the LoRA input, multimodal_llama.py:133-148; 0.05 in run_finetune_*_damc.sh): counter-based Philox masks
    keyed by (seed, step, layer, linear, element) - regenerated in the backward pass instead of stored; one mask per (linear, token): a
    token only uses its own adapter's branch, so the reference's independent per-adapter masks are the same distribution.
  * padded / ragged batches (collator: multimodal_dataset.py:148-214; splice padding: multimodal_arch.py:390-430): rows keep the
    reference's padded [B, Lmax] layout - pad slots are zero embeddings, masked as attention keys by a per-sample length, ignored by
    the loss (-100) - so they contribute exactly zero to every gradient, as in the reference.
  * projectors: MLP / linear (vision, video, point) and the Q-Former of the audio recipe (train/qformer.py).
Limits of this version (raise, never fall back): right padding only."""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .. import _lib, ops
from ..constants import IGNORE_INDEX
from ..model.config import MultimodalConfig, adapter_plan, infer_modals
from ..model.multimodal_llama import MultimodalLlamaForCausalLM
from ..model.encoders_extra import HipQformerProjector
from ..model.projector import HipMlpProjector
from .buckets import bucket_ranges

BF16 = _lib.storage_dtype()      # the library's 16-bit storage element: bf16, or fp16 with MC_STORAGE_DTYPE=fp16 (_lib.set_storage_dtype)
F32 = torch.float32
# linears that read the same activation share one rank-projection GEMM: their A matrices are stored stacked ("A_in")
GROUPS = (("attn_in", (("self_attn", "q_proj"), ("self_attn", "k_proj"), ("self_attn", "v_proj"))),
          ("attn_out", (("self_attn", "o_proj"),)),
          ("mlp_in", (("mlp", "gate_proj"), ("mlp", "up_proj"))),
          ("mlp_out", (("mlp", "down_proj"),)))


class _Param:
    __slots__ = ("name", "off", "shape", "n")

    def __init__(self, name, off, shape):
        self.name, self.off, self.shape, self.n = name, off, tuple(shape), int(np.prod(shape))


class MultimodalTrainStep:
    def __init__(self, model: MultimodalLlamaForCausalLM, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 bucket_layers: int = 4, process_group=None, overlap_wgrad: bool = True, force_exchange: bool = False, dropout_seed: int = 0,
                 mm_projector_lr: Optional[float] = None, mm_language_lr: Optional[float] = None):
        """lr / mm_projector_lr / mm_language_lr: the reference's optimizer groups (llava_trainer.py:210-290; run_finetune_vision_damc.sh:28
        `--mm_projector_lr 2e-5 --mm_language_lr 1e-5`).  With mm_projector_lr set, every `modal_projectors.*` tensor takes that rate; with
        mm_language_lr ALSO set, the `lora_A.default` / `lora_B.default` tensors join the projector group - at mm_projector_lr: the reference
        uses mm_language_lr only as the switch (:212-213), its value never reaches the optimizer; reproduced as is.  Everything else (modal
        adapters, prefix / suffix tokens) takes `lr`.  mm_language_lr without mm_projector_lr changes nothing (:210)."""
        cfg = model.config
        self.p = float(getattr(cfg, "lora_dropout", 0.0) or 0.0)
        if not 0.0 <= self.p < 1.0:
            raise ValueError(f"dropout probability has to be between 0 and 1, but got {self.p}")
        self.dropout_seed = int(dropout_seed)
        if cfg.reset_scaling_weights is not None:
            raise NotImplementedError("training a composed (reset_scaling_weights) checkpoint is not what the reference's stage-2 scripts do")
        if cfg.num_key_value_heads != cfg.num_attention_heads:
            raise NotImplementedError("attention backward is MHA only (Vicuna)")
        self.model, self.cfg, self.dev = model, cfg, model.device
        self.names, scaling, _, _ = adapter_plan(cfg)                       # ['default', <modal>...]
        self.nA, self.r = len(self.names), cfg.lora_r
        self.scale = float(scaling["default"])
        if any(abs(scaling[n] - self.scale) > 0 for n in self.names):
            raise NotImplementedError("per-adapter scaling differs")
        self.R = self.nA * self.r
        if self.R % 64:
            raise ValueError(f"n_adapters * lora_r = {self.R} must be a multiple of 64 for the MFMA GEMMs")
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.mm_projector_lr = None if mm_projector_lr is None else float(mm_projector_lr)
        self.mm_language_lr = None if mm_language_lr is None else float(mm_language_lr)
        self._accum_n = 0                                          # micro-batches whose gradients wait in self.Gacc (gradient accumulation)
        self.Gacc = None
        self._present_acc: List[str] = []
        self.bucket_layers, self.pg = bucket_layers, process_group
        self.world = 1
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.world = dist.get_world_size(process_group)
        except Exception:
            self.world = 1
        # force_exchange: run the bucketed all-reduce even in a world of one process (exercises the RCCL path on a single GPU)
        self._exchange = self.world > 1 or (force_exchange and torch.distributed.is_available() and torch.distributed.is_initialized())
        self.rank = 0
        if self.world > 1:
            import torch.distributed as dist
            self.rank = dist.get_rank(process_group)
        self.step_count = 0
        # weight-gradient (TN) GEMMs run on a second HIP stream: they only feed the gradient buffer, so they overlap the main stream's
        # input-gradient GEMMs, which at B*L = 2728 rows fill only 176 of the 256 CUs
        if overlap_wgrad:
            from ..model.multimodal_llama import _shared_streams       # one side stream per device, shared by every instance (library per-stream state)
            self._wstream = _shared_streams(self.dev, "train_side", 1)[0]
        else:
            self._wstream = None
        self._build_frozen()
        self._build_trainable()
        self._frozen_names = self._check_selection()
        self._buckets = bucket_ranges(self.layer_end, cfg.num_hidden_layers, bucket_layers, self.n_params)
        self._segs = None                                          # chunk table of the two-rate AdamW (built on first use)

    def reference_param_names(self):
        """The trainable set of this step under the reference's parameter names (train_multimodal.py:436-465 with lora_strategy
        'modal+language': every lora_A / lora_B of every adapter, the projectors, the non-default prefix / suffix tokens)."""
        out = []
        for l in range(self.cfg.num_hidden_layers):
            for _, lins in GROUPS:
                for blk, lin in lins:
                    for n in self.names:
                        out += [f"model.layers.{l}.{blk}.{lin}.lora_A.{n}.weight", f"model.layers.{l}.{blk}.{lin}.lora_B.{n}.weight"]
        for name in self.params:
            if name.startswith("model.modal_projectors.") or name.startswith("prefix_tokens.") or name.startswith("suffix_tokens."):
                out.append(name)
        for q in self.qformers.values():
            out += list(getattr(q, "reference_names", lambda: [])())
        return out

    def _check_selection(self):
        """The caller's requires_grad selection (MultimodalLlamaForCausalLM.named_parameters() -> ParamRef.requires_grad, the sequence of
        train_multimodal.py:436-465) against what this step trains.  No selection made: the step's own set.  Supported selections: the
        step's set minus whole non-LoRA tensors (freeze_mm_mlp_adapter: projector tensors frozen - they keep learning rate 0); anything
        else - base weights trainable, a subset of the LoRA adapters ('same' / 'modal' strategies) - is refused, never silently trained
        differently.  Returns the frozen names."""
        sel = getattr(self.model, "_requires_grad", None)
        if not sel:
            return set()
        want = set(self.model.trainable_names())
        have = set(self.reference_param_names())
        # prefix / suffix tokens of the text adapter are selected by the reference's name test but are never spliced (grad None)
        extra = {n for n in want - have if not (n.startswith("prefix_tokens.") or n.startswith("suffix_tokens."))}
        qf_modals = set(self.qformers)
        extra = {n for n in extra if not any(n.startswith(f"model.modal_projectors.{m}.") for m in qf_modals)}
        if extra:
            raise NotImplementedError(f"requires_grad selects {len(extra)} tensors this step does not train (e.g. {sorted(extra)[:3]}): only the "
                                      f"LocalLoRA stage-2 set (lora_A / lora_B, modal projectors, prefix / suffix tokens) has a backward")
        frozen = have - want
        lora_frozen = [n for n in frozen if ".lora_" in n]
        if lora_frozen:
            raise NotImplementedError(f"requires_grad freezes {len(lora_frozen)} LoRA tensors (e.g. {sorted(lora_frozen)[:2]}): lora_strategy "
                                      f"'same' / 'modal' (a subset of the adapters) is not implemented - 'modal+language' trains every adapter")
        return frozen

    def lr_of(self, name: str) -> float:
        """Learning rate of a parameter under the REFERENCE's name (lora_A.{adapter}.weight, modal_projectors.*, prefix_tokens.*): the
        group llava_trainer.py:210-290 puts it in; 0 for a tensor the caller's requires_grad selection froze."""
        if name in getattr(self, "_frozen_names", ()):
            return 0.0
        if self.mm_projector_lr is None:
            return self.lr
        if "modal_projectors" in name or "mm_projector" in name:
            return self.mm_projector_lr
        if self.mm_language_lr is not None and ("lora_A.default" in name or "lora_B.default" in name):
            return self.mm_projector_lr                            # :212-213 extends the projector group; its rate is mm_projector_lr
        return self.lr

    def _lora_segments(self):
        """mc_adamw_seg records over [0, aux_lo): chunks of <= 64 Ki elements, the `default` adapter's rows of every A_in and columns of
        every B_cat flagged as the alternate-rate group."""
        if self._segs is None:
            CH = 1 << 16
            recs = []
            covered = 0
            for name, p_ in self.params.items():
                if p_.off >= self.aux_lo:
                    continue
                if p_.off > covered:                               # alignment padding between tensors: plain elements (zero gradient)
                    recs.append((covered, p_.off - covered, 0, 0, 0))
                if name.endswith(".A_in"):
                    K = p_.shape[1]
                    period, width = self.R * K, self.r * K
                elif name.endswith(".B_cat"):
                    period, width = self.R, self.r
                else:
                    raise AssertionError(name)
                assert period % 4 == 0 and width % 4 == 0 and p_.off % 4 == 0
                for c0 in range(0, p_.n, CH):
                    n = min(CH, p_.n - c0)
                    recs.append((p_.off + c0, n, c0 % period, period, width))
                covered = p_.off + p_.n
            if covered < self.aux_lo:
                recs.append((covered, self.aux_lo - covered, 0, 0, 0))
            for off, n, *_ in recs:
                assert off % 4 == 0 and (n % 4 == 0 or off + n == self.aux_lo), (off, n)
            arr = np.zeros(len(recs), dtype=np.dtype([("off", "<i8"), ("n", "<i4"), ("idx0", "<i4"), ("period", "<i4"), ("width", "<i4")]))
            for i, rcd in enumerate(recs):
                arr[i] = rcd
            self._segs = (torch.from_numpy(arr.view(np.uint8).copy()).to(self.dev), len(recs))
        return self._segs

    # ------------------------------------------------------------------ weights
    def _build_frozen(self):
        cfg, dev, raw = self.cfg, self.dev, self.model._raw
        Hd, I, Ln = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
        t = lambda k: raw[k].to(dev, BF16)
        self.layers = []
        for l in range(Ln):
            p = f"model.layers.{l}"
            wqkv = torch.cat([t(f"{p}.self_attn.{n}_proj.weight") for n in "qkv"], 0)
            wgu = torch.cat([t(f"{p}.mlp.gate_proj.weight"), t(f"{p}.mlp.up_proj.weight")], 0)
            wo, wd = t(f"{p}.self_attn.o_proj.weight"), t(f"{p}.mlp.down_proj.weight")
            L = dict(qkv=ops.pack_weight(wqkv), qkvT=ops.pack_weight(wqkv.t().contiguous()), o=ops.pack_weight(wo),
                     oT=ops.pack_weight(wo.t().contiguous()), gu=ops.pack_weight(wgu), guT=ops.pack_weight(wgu.t().contiguous()),
                     down=ops.pack_weight(wd), downT=ops.pack_weight(wd.t().contiguous()),
                     g_in=t(f"{p}.input_layernorm.weight").contiguous(), g_post=t(f"{p}.post_attention_layernorm.weight").contiguous())
            self.layers.append(L)
        self.g_final = t("model.norm.weight").contiguous()
        lm = t("lm_head.weight")
        self.lm_head, self.lm_headT = ops.pack_weight(lm), ops.pack_weight(lm.t().contiguous())
        self.embed = self.model.model.embed_tokens if self.model.model.embed_tokens is not None else t("model.embed_tokens.weight").contiguous()
        if self.model.model.embed_tokens is None:                 # a from_pretrained() model that was never finalize()d (the train() caller's path):
            self.model.model.embed_tokens = self.embed             # the splice helpers read the table from the model object
        D = cfg.head_dim
        inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, D, 2, dtype=F32) / D))
        ang = torch.outer(torch.arange(cfg.max_position_embeddings, dtype=F32), inv)
        self.cos, self.sin = ang.cos().to(dev).contiguous(), ang.sin().to(dev).contiguous()

    def _build_trainable(self):
        """Flat fp32 master buffer in backward order (last layer first) so gradient buckets complete early."""
        cfg, raw = self.cfg, self.model._raw
        Hd, I, Ln = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
        params: List[_Param] = []
        init: List[torch.Tensor] = []
        off = 0
        self.layer_end = {}                                       # layer -> end offset of its block in the flat buffer

        def add(name, tensor):
            nonlocal off
            if off % 4:                                            # 16-byte aligned slices: AdamW runs per parameter group (see optimizer_step)
                pad = 4 - off % 4
                init.append(torch.zeros(pad, dtype=F32, device=self.dev))
                off += pad
            params.append(_Param(name, off, tensor.shape))
            init.append(tensor.reshape(-1).to(self.dev, F32))
            off += tensor.numel()
        self._register = add

        for l in reversed(range(Ln)):
            for gname, lins in reversed(GROUPS):
                a_rows = []
                for blk, lin in lins:
                    pre = f"model.layers.{l}.{blk}.{lin}"
                    a = [raw.get(f"{pre}.lora_A.{n}.weight") for n in self.names]
                    b = [raw.get(f"{pre}.lora_B.{n}.weight") for n in self.names]
                    if any(x is None for x in a + b):
                        raise ValueError(f"{pre}: lora_A/lora_B missing for one of the adapters {self.names}")
                    a_rows += [x.float() for x in a]
                    add(pre + ".B_cat", torch.cat([x.float() for x in b], 1))        # [N, nA*r]
                add(f"model.layers.{l}.{gname}.A_in", torch.cat(a_rows, 0))           # [n_linears*nA*r, K], linear-major, adapter, rank
            self.layer_end[l] = off
        self.proj_modals = []
        self.aux_lo = off                                          # projector / prefix / suffix parameters live in [aux_lo, n_params)
        n_before = len(params)
        self.qformers = {}
        for m, proj in self.model.model.modal_projectors.items():
            if isinstance(proj, HipQformerProjector):               # audio stage-2 recipe (run_finetune_audio_damc.sh:37-38)
                from .qformer import QformerTrainable
                self.qformers[m] = QformerTrainable(self, m, proj, raw)
                continue
            if not isinstance(proj, HipMlpProjector):
                raise NotImplementedError(f"projector of modality '{m}' is neither an MLP / linear nor a Q-Former projector: its backward is not implemented")
            self.proj_modals.append(m)
            keys = ["weight"] if proj.depth == 0 else [f"{2 * i}.weight" for i in range(proj.depth)]
            for k in keys:
                add(f"model.modal_projectors.{m}.{k}", raw_or_fail(raw, f"model.modal_projectors.{m}.{k}"))
                add(f"model.modal_projectors.{m}.{k.replace('weight', 'bias')}", raw_or_fail(raw, f"model.modal_projectors.{m}.{k.replace('weight', 'bias')}"))
        for which in ("prefix_tokens", "suffix_tokens"):
            d = getattr(self.model, which) or {}
            for m in d:
                if m == "default":
                    continue                                       # never spliced: no gradient (reference: grad None)
                add(f"{which}.{m}", d[m].float())
        self.params = {p.name: p for p in params}
        self.n_params = off
        # per-modality parameter groups outside the decoder layers.  A modality absent from a batch contributes no gradient: the reference
        # leaves .grad = None there, so AdamW skips those tensors entirely (no moment decay, no step count) - mirrored in optimizer_step
        self.aux_params: Dict[str, List[_Param]] = {}
        for p_ in params[n_before:]:
            modal = p_.name.split(".")[2] if p_.name.startswith("model.modal_projectors.") else p_.name.split(".", 1)[1]
            self.aux_params.setdefault(modal, []).append(p_)
        self._aux_steps = {m: 0 for m in self.aux_params}
        self._present: List[str] = []
        dev = self.dev
        self.P = torch.cat(init)
        self.G = torch.zeros(off, dtype=F32, device=dev)
        self.m1, self.m2 = torch.zeros_like(self.G), torch.zeros_like(self.G)
        self.P16 = ops.cast_bf16(self.P)
        self._packed: Dict[str, ops.PackedWeight] = {}
        self._repack()

    def _repack(self):
        """MFMA-fragment forms of the LoRA matrices, refreshed once per optimizer step in ONE launch: A_in and B_cat as they multiply in
        the forward, and their transposes (packed straight from the row-major bf16 masters, no transposed copy) for the input-gradient
        GEMMs.  The descriptor table (source / destination addresses never change) is built on first use."""
        if not self._packed:
            descs = []
            for name, p in self.params.items():
                if not (name.endswith(".A_in") or name.endswith(".B_cat")):
                    continue
                w = self.view(self.P16, name)
                N, K = w.shape
                fw = ops.PackedWeight(torch.empty(ops.packed_elems(N, K), dtype=BF16, device=self.dev), N, K)
                tr = ops.PackedWeight(torch.empty(ops.packed_elems(K, N), dtype=BF16, device=self.dev), K, N)
                self._packed[name], self._packed[name + ".T"] = fw, tr
                descs.append((w.data_ptr(), fw.data.data_ptr(), w.stride(0), 1, N, K))
                descs.append((w.data_ptr(), tr.data.data_ptr(), 1, w.stride(0), K, N))          # W' = w^T: W'[n][k] = w[k][n]
                if self.p > 0 and name.endswith(".A_in"):
                    # with dropout every linear of a group sees its own dropped input: per-linear views of the stacked image (block rows are
                    # contiguous) for the forward, per-linear transposes for the input gradient
                    Kp = ops.ceil_to(K, 64)
                    for j in range(N // self.R):
                        self._packed[f"{name}.{j}"] = ops.PackedWeight(fw.data[j * self.R * Kp:(j + 1) * self.R * Kp], self.R, K)
                        trj = ops.PackedWeight(torch.empty(ops.packed_elems(K, self.R), dtype=BF16, device=self.dev), K, self.R)
                        self._packed[f"{name}.{j}.T"] = trj
                        descs.append((w.data_ptr() + j * self.R * w.stride(0) * 2, trj.data.data_ptr(), 1, w.stride(0), K, self.R))
            rec = np.zeros(len(descs), dtype=np.dtype([("src", "<u8"), ("dst", "<u8"), ("sn", "<i8"), ("sk", "<i8"), ("N", "<i4"), ("K", "<i4")]))
            for i, dsc in enumerate(descs):
                rec[i] = dsc
            self._pack_descs = torch.from_numpy(rec.view(np.uint8).copy()).to(self.dev)
            self._n_pack = len(descs)
        from .. import _lib
        _lib.check(_lib.lib().mc_pack_weight_batch_bf16(self._pack_descs.data_ptr(), self._n_pack, 16, ops._stream()), "mc_pack_weight_batch_bf16")

    def view(self, buf, name):
        p = self.params[name]
        return buf[p.off:p.off + p.n].view(*p.shape)

    def named_gradients(self) -> Dict[str, torch.Tensor]:
        """Gradients under the reference's parameter names (lora_A.{adapter}.weight ...)."""
        return self._named(self.G)

    def named_parameters(self) -> Dict[str, torch.Tensor]:
        """fp32 master weights under the reference's parameter names (views of the flat buffer)."""
        return self._named(self.P)

    def _named(self, buf) -> Dict[str, torch.Tensor]:
        out = {}
        for name, p in self.params.items():
            g = self.view(buf, name)
            if name.endswith(".A_in"):
                l, gname = name.split(".")[2], name.split(".")[3]
                lins = dict(GROUPS)[gname]
                for j, (blk, lin) in enumerate(lins):
                    for i, n in enumerate(self.names):
                        out[f"model.layers.{l}.{blk}.{lin}.lora_A.{n}.weight"] = g[j * self.R + i * self.r:j * self.R + (i + 1) * self.r]
            elif name.endswith(".B_cat"):
                for i, n in enumerate(self.names):
                    out[name[:-6] + f".lora_B.{n}.weight"] = g[:, i * self.r:(i + 1) * self.r]
            elif name.startswith("prefix_tokens.") or name.startswith("suffix_tokens."):
                out[name] = g.view(1, *g.shape)
            else:
                out[name] = g
        return out

    # ------------------------------------------------------------------ helpers
    def _on_side(self, fn, *used):
        """Run fn() on the side stream once everything enqueued so far on the main stream is complete; returns an event that marks its
        completion (None without a side stream: fn ran inline).  `used`: main-stream tensors fn reads."""
        if self._wstream is None:
            fn()
            return None
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self._wstream):
            self._wstream.wait_event(ready)
            fn()
            done = torch.cuda.Event()
            done.record()
        for t in used:
            t.record_stream(self._wstream)                          # the allocator must not recycle them under the side stream
        return done

    _LIN_ID = {"q_proj": 0, "k_proj": 1, "v_proj": 2, "o_proj": 3, "gate_proj": 4, "up_proj": 5, "down_proj": 6}

    def _stream_id(self, layer, lin):
        return layer * 8 + self._LIN_ID[lin]

    def dropout_keep_scale(self, layer: int, lin: str, M: int, K: int) -> torch.Tensor:
        """keep / (1 - p) of the LAST step's mask of one linear, [M, K] (tests: the oracle applies the same masks)."""
        return ops.dropout(torch.ones(M, K, dtype=BF16, device=self.dev), self.p, self._seed, self._stream_id(layer, lin)).float()

    def _lora_fwd_begin(self, x, layer, gname, row_adapter):
        """Rank projection of a group, T = mask(x [A_0; A_1; ..]^T) ([M, n_linears * R]; the routing mask zeroes, per row, the rank blocks
        of the other adapters).  It only depends on x, so it runs on the side stream next to the base GEMM of the same input."""
        box = {}

        def run():
            aname = f"model.layers.{layer}.{gname}.A_in"
            if self.p > 0:
                lins = dict(GROUPS)[gname]
                T = torch.empty(x.shape[0], len(lins) * self.R, dtype=BF16, device=self.dev)
                xds = []
                for j, (blk, lin) in enumerate(lins):
                    xd = ops.dropout(x, self.p, self._seed, self._stream_id(layer, lin))
                    ops.linear(xd, self._packed[f"{aname}.{j}"], out=T[:, j * self.R:(j + 1) * self.R], auto_split=True)
                    xds.append(xd)
                box["xd"] = xds
            else:
                T = ops.linear(x, self._packed[aname], auto_split=True)
            ops.lora_mask_rows(T, row_adapter, self.r, self.nA)
            box["T"] = T
        done = self._on_side(run, x)
        self._xd[(layer, gname)] = box.get("xd")
        return box["T"], done

    def _lora_fwd_end(self, T, done, ys, layer, gname, saved):
        """y_j += s * T_j B_j^T in place on the views ys[j] ([M, N_j]) of the base GEMM's output."""
        if done is not None:
            torch.cuda.current_stream().wait_event(done)
            T.record_stream(torch.cuda.current_stream())
        for j, (blk, lin) in enumerate(dict(GROUPS)[gname]):
            ops.linear(T[:, j * self.R:(j + 1) * self.R], self._packed[f"model.layers.{layer}.{blk}.{lin}.B_cat"], residual=ys[j], out=ys[j],
                       alpha=self.scale)
        saved[f"{layer}.{gname}.T"] = T

    def _wgrad(self, a_list, b_list, out_list, alpha=1.0):
        """out_i = alpha * a_i^T b_i into the gradient buffer, on the side stream once the operands are complete on the main stream."""
        self._on_side(lambda: ops.gemm_tn(a_list, b_list, out_list, alpha=alpha), *a_list, *b_list)

    def _join_wgrad(self):
        if self._wstream is not None:
            torch.cuda.current_stream().wait_stream(self._wstream)

    def _lora_bwd_begin(self, dys, layer, gname, row_adapter):
        """dT = mask([dy_j B_j]_j) on the side stream, next to the base input-gradient GEMM that reads the same dy."""
        lins = dict(GROUPS)[gname]
        M = dys[0].shape[0]
        box = {}

        def run():
            dT = torch.empty(M, len(lins) * self.R, dtype=BF16, device=self.dev)
            for j, (blk, lin) in enumerate(lins):
                ops.linear(dys[j], self._packed[f"model.layers.{layer}.{blk}.{lin}.B_cat.T"], out=dT[:, j * self.R:(j + 1) * self.R], auto_split=True)
            ops.lora_mask_rows(dT, row_adapter, self.r, self.nA)
            box["dT"] = dT
        done = self._on_side(run, *dys)
        return box["dT"], done

    def _lora_bwd_end(self, dT, done, dys, x, dx, layer, gname, saved):
        """dx += s * dT A_in;  dB_j = s dy_j^T T_j;  dA_in = s dT^T x.  The two weight gradients reduce over the token rows: TN GEMMs
        straight from the row-major activations (one launch for the same-shape linears of the group), on the side stream."""
        lins = dict(GROUPS)[gname]
        aname = f"model.layers.{layer}.{gname}.A_in"
        T = saved[f"{layer}.{gname}.T"]
        if done is not None:
            torch.cuda.current_stream().wait_event(done)
            dT.record_stream(torch.cuda.current_stream())
        bnames = [f"model.layers.{layer}.{blk}.{lin}.B_cat" for blk, lin in lins]
        self._wgrad(list(dys), [T[:, j * self.R:(j + 1) * self.R] for j in range(len(lins))], [self.view(self.G, bn) for bn in bnames],
                    alpha=self.scale)
        xds = self._xd.pop((layer, gname), None)
        if xds is None:
            ops.linear(dT, self._packed[aname + ".T"], residual=dx, out=dx, alpha=self.scale)
            self._wgrad([dT], [x], [self.view(self.G, aname)], alpha=self.scale)
            return
        # dropout: dx += s * (dT_j A_j) * keep_j / (1 - p) with linear j's own mask (regenerated), dA_j = s dT_j^T (dropped x_j)
        gA = self.view(self.G, aname)
        for j, (blk, lin) in enumerate(lins):
            tmp = ops.linear(dT[:, j * self.R:(j + 1) * self.R], self._packed[f"{aname}.{j}.T"])
            ops.dropout(tmp, self.p, self._seed, self._stream_id(layer, lin), out=dx, accumulate=True, alpha=self.scale)
        self._wgrad([dT[:, j * self.R:(j + 1) * self.R] for j in range(len(lins))], xds, [gA[j * self.R:(j + 1) * self.R] for j in range(len(lins))],
                    alpha=self.scale)

    # ------------------------------------------------------------------ one step
    def forward_backward(self, input_ids, labels, modal_inputs, attention_mask=None, accumulate: bool = False) -> torch.Tensor:
        """Returns the loss (fp32 scalar tensor) of this micro-batch.

        accumulate=False: the LAST (or only) micro-batch of an optimizer step - afterwards self.G holds the gradient SUM over this call and
        every accumulate=True call since the last optimizer_step(), all-reduced over the ranks; optimizer_step() divides by
        world x micro-batches (HF Trainer: each micro-batch's loss / gradient_accumulation_steps, gradients summed;
        run_finetune_vision_damc.sh:45 `--gradient_accumulation_steps 4`).
        accumulate=True: an earlier micro-batch - its gradient is added to the pending sum, nothing is exchanged (DDP no_sync)."""
        from .. import _lib
        pending = self._accum_n > 0
        exch = self._exchange
        if accumulate or pending:
            self._exchange = False                                 # exchange once, on the sum
        try:
            if self._wstream is None:
                loss = self._forward_backward(input_ids, labels, modal_inputs, attention_mask)
            else:
                # the side stream fills the CUs an under-filled base GEMM leaves idle; narrower tiles would compete with it for them
                _lib.check(_lib.lib().mc_gemm_set_option(b"tile192", 0), "mc_gemm_set_option")
                try:
                    loss = self._forward_backward(input_ids, labels, modal_inputs, attention_mask)
                finally:
                    _lib.lib().mc_gemm_set_option(b"tile192", 1)
        finally:
            self._exchange = exch
        if accumulate:
            if self.Gacc is None:
                self.Gacc = torch.zeros_like(self.G)
            if self._accum_n == 0:
                self.Gacc.copy_(self.G)
                self._present_acc = list(self._present)
            else:
                ops.axpy_f32(self.Gacc, self.G)
                self._present_acc += [m for m in self._present if m not in self._present_acc]
            self._accum_n += 1
            self._micro = self._accum_n
            return loss
        if pending:
            ops.axpy_f32(self.G, self.Gacc)                        # G = sum over the micro-batches of this optimizer step
            self._present = self._present_acc + [m for m in self._present if m not in self._present_acc]
            self._micro = self._accum_n + 1
            self._accum_n = 0
            if self._exchange:
                hs = [self._allreduce_async(lo, hi) for (_, lo, hi) in self._buckets]
                for h in hs:
                    h.wait()
        else:
            self._micro = 1
        return loss

    def _forward_backward(self, input_ids, labels, modal_inputs, attention_mask=None) -> torch.Tensor:
        model, cfg, dev = self.model, self.cfg, self.dev
        Hd, I, Hh, D, V = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.head_dim, cfg.vocab_size
        HD = Hh * D
        eps = cfg.rms_norm_eps
        # Philox key of this step's dropout masks: (step counter, seed x rank) - every rank and every step draws fresh masks
        self._seed = ((self.dropout_seed * 4096 + self.rank) << 32) | (self.step_count & 0xFFFFFFFF)
        self._xd = {}
        # ---- encoders (frozen, no gradient) + trainable projector forward with saved pre-activations
        saved: Dict[str, torch.Tensor] = {}
        feats = self._encode(modal_inputs, saved)
        plan = model._plan(input_ids, attention_mask, labels, modal_inputs, feats)
        if not plan.mask_is_suffix:
            raise NotImplementedError("attention_mask with zeros before the last attended token (left padding / holes) is not implemented: "
                                      "right-pad the batch as the reference's collator does")
        # modalities without a block in this batch get no gradient; their slices of G still hold the previous step's values (every kernel
        # overwrites, nothing accumulates), so clear them before they can reach the all-reduce
        self._present = [m for m in plan.modal_order if m in self.aux_params]
        for m, plist in self.aux_params.items():
            if m not in self._present:
                for p_ in plist:
                    self.G[p_.off:p_.off + p_.n].zero_()
        B, L = plan.B, plan.Lmax
        # padded layout [B, Lmax] as in the reference (multimodal_arch.py:390-430): slots behind a sample's spliced length are zero rows;
        # keys at or behind the attended length (those slots and right-padding text tokens) are masked by kv_lens
        ragged = not bool((plan.valid_lens == L).all())
        kv_lens = torch.from_numpy(np.ascontiguousarray(plan.valid_lens, dtype=np.int32)).to(dev) if ragged else None
        M = B * L
        Mp = ops.ceil_to(M, 64)
        bb, tt = np.divmod(np.arange(M), L)
        x = torch.empty(M, Hd, dtype=BF16, device=dev)
        model._gather_rows(plan, feats, bb, tt, x)
        i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
        row_b, row_t = i32(bb), i32(tt)
        adapter = np.zeros(M, dtype=np.int32)
        if cfg.lora_strategy in ("modal", "modal+language"):
            for i, m in enumerate(plan.modal_order):
                adapter[(plan.src_modal == i).reshape(-1)] = self.names.index(m)
        row_adapter = i32(adapter)
        st_q = (L * HD, HD, D)
        st_kv = (Hh * L * D, D, L * D)                              # cache layout [B][H][L][D]: (batch, token, head) strides
        acts = []
        # ---- forward
        for l, W in enumerate(self.layers):
            a = {"x": x}
            n1 = ops.rmsnorm(x, W["g_in"], eps)
            T, ev = self._lora_fwd_begin(n1, l, "attn_in", row_adapter)
            qkv = ops.linear(n1, W["qkv"])
            self._lora_fwd_end(T, ev, [qkv[:, j * HD:(j + 1) * HD] for j in range(3)], l, "attn_in", saved)
            q_seq = torch.empty(M, HD, dtype=BF16, device=dev)
            kc, vc = (torch.empty(B, Hh, L, D, dtype=BF16, device=dev) for _ in range(2))
            ops.rope_kv(qkv, row_b, row_t, row_t, self.cos, self.sin, q_seq, kc, vc, Hh, Hh, D, L, L)
            attn = torch.empty(M, HD, dtype=BF16, device=dev)
            lse = torch.empty(B * Hh * L, dtype=F32, device=dev)
            ops.attn_prefill_lse(q_seq, kc, vc, attn, lse, B, Hh, L, L, D, st_q, st_kv, st_kv, HD, True, kv_lens=kv_lens)
            T, ev = self._lora_fwd_begin(attn, l, "attn_out", row_adapter)
            x1 = ops.linear(attn, W["o"], residual=x)
            self._lora_fwd_end(T, ev, [x1], l, "attn_out", saved)
            n2 = ops.rmsnorm(x1, W["g_post"], eps)
            T, ev = self._lora_fwd_begin(n2, l, "mlp_in", row_adapter)
            gu = ops.linear(n2, W["gu"])
            self._lora_fwd_end(T, ev, [gu[:, :I], gu[:, I:]], l, "mlp_in", saved)
            inter = ops.silu_mul(gu, I)
            T, ev = self._lora_fwd_begin(inter, l, "mlp_out", row_adapter)
            x2 = ops.linear(inter, W["down"], residual=x1)
            self._lora_fwd_end(T, ev, [x2], l, "mlp_out", saved)
            a.update(n1=n1, q=q_seq, kc=kc, vc=vc, attn=attn, lse=lse, x1=x1, n2=n2, gu=gu, inter=inter)
            acts.append(a)
            x = x2
        nf = ops.rmsnorm(x, self.g_final, eps)
        logits = ops.linear(nf, self.lm_head, out_f32=True)
        # ---- shifted cross-entropy (:722-733): row (b, t) predicts labels[b, t + 1]
        lab = np.full((B, L), IGNORE_INDEX, dtype=np.int64)
        lab[:, :-1] = plan.labels[:, 1:]
        n_valid = int((lab != IGNORE_INDEX).sum())
        if n_valid == 0:
            raise ValueError("no target token in the batch (all labels are IGNORE_INDEX)")
        loss_rows, dlogits = ops.ce_loss(logits, torch.from_numpy(lab.reshape(-1)).to(dev), 1.0 / n_valid)
        loss = loss_rows.sum() / n_valid
        # ---- backward
        dnf = ops.linear(dlogits, self.lm_headT)
        dx = ops.rmsnorm_bwd(x, self.g_final, dnf, eps)
        handles = []
        for l in reversed(range(len(self.layers))):
            W, a = self.layers[l], acts[l]
            # down_proj
            dT, ev = self._lora_bwd_begin([dx], l, "mlp_out", row_adapter)
            d_inter = ops.linear(dx, W["downT"])
            self._lora_bwd_end(dT, ev, [dx], a["inter"], d_inter, l, "mlp_out", saved)
            dgu = ops.swiglu_bwd(a["gu"], d_inter)
            dT, ev = self._lora_bwd_begin([dgu[:, :I], dgu[:, I:]], l, "mlp_in", row_adapter)
            dn2 = ops.linear(dgu, W["guT"])
            self._lora_bwd_end(dT, ev, [dgu[:, :I], dgu[:, I:]], a["n2"], dn2, l, "mlp_in", saved)
            dx1 = ops.rmsnorm_bwd(a["x1"], W["g_post"], dn2, eps, dres=dx)
            # o_proj
            dT, ev = self._lora_bwd_begin([dx1], l, "attn_out", row_adapter)
            d_attn = ops.linear(dx1, W["oT"])
            self._lora_bwd_end(dT, ev, [dx1], a["attn"], d_attn, l, "attn_out", saved)
            # attention + RoPE
            dqkv = torch.empty(M, 3 * HD, dtype=BF16, device=dev)
            st3 = (L * 3 * HD, 3 * HD, D)
            ops.attn_bwd(a["q"], a["kc"], a["vc"], a["attn"], d_attn, a["lse"], dqkv, dqkv[:, HD:], dqkv[:, 2 * HD:], B, Hh, L, L, D,
                         st_q, st_kv, st_kv, st_q, st3, st3, st3, True, kv_lens=kv_lens)
            ops.rope_inplace(dqkv, row_t, self.cos, self.sin, 2 * Hh, D, -1.0)
            dqs = [dqkv[:, j * HD:(j + 1) * HD] for j in range(3)]
            dT, ev = self._lora_bwd_begin(dqs, l, "attn_in", row_adapter)
            dn1 = ops.linear(dqkv, W["qkvT"])
            self._lora_bwd_end(dT, ev, dqs, a["n1"], dn1, l, "attn_in", saved)
            dx = ops.rmsnorm_bwd(a["x"], W["g_in"], dn1, eps, dres=dx1)
            acts[l] = None
            if self._exchange:
                for (ready, lo, hi) in self._buckets:
                    if ready == l:
                        self._join_wgrad()
                        handles.append(self._allreduce_async(lo, hi))
        # ---- spliced feature blocks -> prefix / suffix tokens and the projectors
        self._backward_features(dx, plan, feats, saved)
        self._join_wgrad()
        if self._exchange:
            for (ready, lo, hi) in self._buckets:
                if ready == -1:
                    handles.append(self._allreduce_async(lo, hi))
            for h in handles:
                h.wait()
        return loss

    # ------------------------------------------------------------------ encoders / projectors
    def _encode(self, modal_inputs, saved):
        model, dev = self.model, self.dev
        feats = {}
        for modal in [m for m in model.modal_names if m != "default"]:
            if modal not in modal_inputs:
                continue
            enc = model.model.get_modal_encoder(modal)
            xin = modal_inputs[modal]
            f = enc(**xin) if isinstance(xin, dict) else enc(xin)   # BEATs takes audio_inputs / audio_padding_mask (multimodal_arch.py:233-235)
            f = f[0] if isinstance(f, tuple) else f
            if modal == "video":
                b, t, n, d = f.shape
                f = f.reshape(b, t * n, d)
            f = f.to(BF16).contiguous()
            nI, T, Dm = f.shape
            proj = model.model.modal_projectors[modal]
            if modal in self.qformers:
                out_q = self.qformers[modal].forward(f)              # (nI, n_queries, hidden)
                T = out_q.shape[1]
                h = out_q.reshape(nI * T, -1)
            else:
                h = f.view(nI * T, Dm)
            saved[f"proj.{modal}.in"] = h
            keys = [] if modal in self.qformers else (["weight"] if proj.depth == 0 else [f"{2 * i}.weight" for i in range(proj.depth)])
            for i, k in enumerate(keys):
                w16 = self.view(self.P16, f"model.modal_projectors.{modal}.{k}")
                b16 = self.view(self.P16, f"model.modal_projectors.{modal}.{k.replace('weight', 'bias')}")
                if i > 0:
                    saved[f"proj.{modal}.pre{i}"] = h
                    h = ops.act(h, "gelu")
                    saved[f"proj.{modal}.h{i}"] = h
                hin = h if h.shape[1] % 64 == 0 else torch.nn.functional.pad(h, (0, ops.ceil_to(h.shape[1], 64) - h.shape[1]))
                h = ops.linear(hin, ops.pack_weight(w16, b16))
            out = h.view(nI, T, -1)
            parts = []
            pre = self.params.get(f"prefix_tokens.{modal}")
            suf = self.params.get(f"suffix_tokens.{modal}")
            if pre is not None:
                parts.append(self.view(self.P16, pre.name).view(1, -1, out.shape[-1]).expand(nI, -1, -1))
            parts.append(out)
            if suf is not None:
                parts.append(self.view(self.P16, suf.name).view(1, -1, out.shape[-1]).expand(nI, -1, -1))
            from ..model.multimodal_llama import _cat_rows
            feats[modal] = _cat_rows(parts).contiguous() if len(parts) > 1 else out.contiguous()
            saved[f"proj.{modal}.nI"], saved[f"proj.{modal}.T"] = nI, T
        return feats

    def _backward_features(self, dx, plan, feats, saved):
        dev, Hd = self.dev, self.cfg.hidden_size
        i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
        sm, sr = plan.src_modal.reshape(-1), plan.src_row.reshape(-1)
        for i, modal in enumerate(plan.modal_order):
            rows = np.nonzero(sm == i)[0]
            nI, T = saved[f"proj.{modal}.nI"], saved[f"proj.{modal}.T"]
            Tb = feats[modal].shape[1]
            dblk = torch.zeros(nI * Tb, Hd, dtype=BF16, device=dev)                  # items that were not spliced keep zero gradient
            if len(rows):
                ops.copy_rows(dx, dblk, len(rows), i32(rows), i32(sr[rows]))
            pre, suf = self.params.get(f"prefix_tokens.{modal}"), self.params.get(f"suffix_tokens.{modal}")
            n_pre = pre.shape[0] if pre is not None else 0
            n_suf = suf.shape[0] if suf is not None else 0
            if n_pre or n_suf:
                tok = ops.colsum(dblk.view(nI, Tb * Hd)).view(Tb, Hd)                  # sum over items (tokens are shared: .expand)
                if n_pre:
                    self.view(self.G, pre.name).copy_(tok[:n_pre])
                if n_suf:
                    self.view(self.G, suf.name).copy_(tok[Tb - n_suf:])
            dout = dblk.view(nI, Tb, Hd)[:, n_pre:Tb - n_suf].reshape(nI * T, Hd).contiguous() if (n_pre or n_suf) else dblk
            if modal in self.qformers:
                self.qformers[modal].backward(dout)
                continue
            proj = self.model.model.modal_projectors[modal]
            keys = ["weight"] if proj.depth == 0 else [f"{2 * k}.weight" for k in range(proj.depth)]
            Mf = nI * T
            Mfp = ops.ceil_to(Mf, 64)
            d = dout
            for k_i in reversed(range(len(keys))):
                k = keys[k_i]
                wname = f"model.modal_projectors.{modal}.{k}"
                hin = saved[f"proj.{modal}.h{k_i}"] if k_i > 0 else saved[f"proj.{modal}.in"]
                # dW[o][i] = sum_m d[m][o] hin[m][i];  db = colsum(d)
                self._wgrad([d], [hin], [self.view(self.G, wname)])
                ops.colsum(d, out=self.view(self.G, wname.replace("weight", "bias")))
                if k_i > 0:
                    w16 = self.view(self.P16, wname)
                    dh = ops.linear(d, ops.pack_weight_t(w16))
                    d = ops.act(saved[f"proj.{modal}.pre{k_i}"], "gelu", dy=dh)

    # ------------------------------------------------------------------ exchange + update
    def _allreduce_async(self, lo, hi):
        import torch.distributed as dist
        return dist.all_reduce(self.G[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def optimizer_step(self):
        """AdamW on the mean gradient over ranks (DDP semantics: all-reduce SUM, then 1 / world_size)."""
        if self._accum_n:
            raise RuntimeError("optimizer_step() with accumulated micro-batches pending: the last forward_backward of a step takes accumulate=False")
        self.step_count += 1
        b1, b2 = self.betas
        lo = self.aux_lo
        gs = 1.0 / (self.world * max(getattr(self, "_micro", 1), 1))
        lr_default_adapter = self.lr_of("lora_A.default")
        if lr_default_adapter != self.lr:
            segs, n_segs = self._lora_segments()
            ops.adamw_segments(self.P, self.G, self.m1, self.m2, self.P16, segs, n_segs, self.lr, lr_default_adapter, b1, b2, self.eps, self.wd,
                               self.step_count, grad_scale=gs)
        else:
            ops.adamw(self.P[:lo], self.G[:lo], self.m1[:lo], self.m2[:lo], self.P16[:lo], self.lr, b1, b2, self.eps, self.wd, self.step_count,
                      grad_scale=gs)
        # projector / prefix / suffix tensors: only those of modalities that were in the batch (torch.optim.AdamW skips grad-None tensors and
        # keeps a step count per tensor).  Under data parallelism a modality must be present on every rank or on none (as with DDP, where a
        # parameter unused on one rank is an error without find_unused_parameters).
        for m in self._present:
            self._aux_steps[m] += 1
            for p_ in self.aux_params[m]:
                sl = slice(p_.off, p_.off + p_.n)
                # weight decay: not on biases, not on LayerNorm parameters (llava_trainer.py:208-209 `decay_parameters`)
                wd = 0.0 if (p_.name.endswith("bias") or "layernorm" in p_.name.lower()) else self.wd
                ops.adamw(self.P[sl], self.G[sl], self.m1[sl], self.m2[sl], self.P16[sl], self.lr_of(p_.name), b1, b2, self.eps, wd,
                          self._aux_steps[m], grad_scale=gs)
        self._repack()

    def step(self, input_ids, labels, modal_inputs, attention_mask=None) -> torch.Tensor:
        loss = self.forward_backward(input_ids, labels, modal_inputs, attention_mask)
        self.optimizer_step()
        return loss

    def step_accumulated(self, micro_batches) -> torch.Tensor:
        """One optimizer step over several micro-batches [(input_ids, labels, modal_inputs[, attention_mask]), ...] (HF Trainer with
        gradient_accumulation_steps = len(micro_batches)); returns the mean of the micro-batch losses."""
        micro_batches = list(micro_batches)
        losses = []
        for i, mb in enumerate(micro_batches):
            losses.append(self.forward_backward(*mb, accumulate=i < len(micro_batches) - 1))
        self.optimizer_step()
        return torch.stack([l.reshape(()) for l in losses]).mean()


def raw_or_fail(raw, key):
    if key not in raw:
        raise ValueError(f"state dict lacks {key}")
    return raw[key]
