"""MultimodalLlamaForCausalLM on the HIP path.

Host-side mirror of the reference's model API (same method names, argument meaning and error behaviour):
  modelcompose/model/language_model/multimodal_llama.py:622-767  MultimodalLlamaForCausalLM
  modelcompose/model/multimodal_arch.py:169-459                  MultimodalMetaForCausalLM (encode / splice)
All arithmetic runs in libmc_hip.so (include/mc_hip.h); PyTorch only owns device memory and the stream.
Composition: at load every LocalLoRA linear is expanded into one dense weight per routed adapter
(W + Σ scale·B·A, csrc/compose.hip), so a forward is plain GEMMs over adapter-grouped rows."""
from __future__ import annotations

import functools
import os
import threading
import time

import ctypes as C
import math
from typing import Dict, List, Optional

import numpy as np
import torch

from .. import _lib, ops
from ..constants import IGNORE_INDEX, MODAL_TOKEN_INDEXES
from .config import MultimodalConfig, adapter_plan, composition_terms, infer_modals
from .splice import SplicePlan, plan_splice, routed_layout

BF16 = _lib.storage_dtype()      # the library's 16-bit storage element: bf16, or fp16 with MC_STORAGE_DTYPE=fp16 (_lib.set_storage_dtype)


# Side streams are shared by every model instance of the process (one set per device): the library keeps a per-stream scratch slot (the
# tile GEMMs' rms_out route, a pool of 16: mc_gemm_reserve_workspace) - streams created per model would exhaust it after a few instances.
_STREAM_POOLS: Dict = {}


def _shared_streams(device, kind: str, n: int):
    idx = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    pool = _STREAM_POOLS.setdefault((idx, kind), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def _cu_count() -> int:
    n = C.c_int(0)
    _lib.check(_lib.lib().mc_device_info(C.byref(n), None, None, 0), "mc_device_info")
    return n.value


def _serialised(fn):
    """generate() / forward() drive the C handle through per-handle one-shot state (tail_adapter, key mask, sampling): calls on one model
    are serialised by the model's re-entrant lock, which a ContinuousBatcher that owns the model takes around its admissions and decode
    steps too - another thread's generate() then waits for a step instead of consuming the engine's settings."""
    @functools.wraps(fn)
    def wrapper(self, *a, **kw):
        with self._lock:
            return fn(self, *a, **kw)
    return wrapper


LINEARS = (("self_attn", ("q_proj", "k_proj", "v_proj", "o_proj")), ("mlp", ("gate_proj", "up_proj", "down_proj")))


class CausalLMOutputWithPast:
    """Field-compatible stand-in for transformers.modeling_outputs.CausalLMOutputWithPast."""

    def __init__(self, loss=None, logits=None, past_key_values=None, hidden_states=None, attentions=None):
        self.loss, self.logits, self.past_key_values = loss, logits, past_key_values
        self.hidden_states, self.attentions = hidden_states, attentions

    def __getitem__(self, i):
        return tuple(v for v in (self.loss, self.logits, self.past_key_values) if v is not None)[i]


class _PastShape:
    """What a caller reads from one cached key / value tensor of the reference's tuple cache: its shape (multimodal_arch.py:290-293 and
    HF's generation loop use past_key_values[-1][-1].shape[-2] = cached length)."""

    def __init__(self, shape):
        self.shape = tuple(shape)

    def size(self, i=None):
        return self.shape if i is None else self.shape[i]


class HipPastKeyValues:
    """The `past_key_values` of the HIP path (multimodal_llama.py:523, :676-688, :747-767): an opaque handle on the device-resident KV
    cache of one batch - [layer][B][Hkv][Smax][D] K and V buffers, per-row cached lengths - returned by forward(use_cache=True) and
    taken back by forward(past_key_values=...).  It is NOT a tuple of tensors (the cache is pre-allocated and appended in place by the
    attention kernel, never concatenated), but it answers what loops written against the tuple ask of it: len(), cache[i][j].shape
    (so `past_key_values[-1][-1].shape[-2]` is the cached length), get_seq_length().  Rows may hold different lengths (right-padded
    prompts): the shape reports the longest."""

    def __init__(self, model, st, lens):
        self.model, self.st = model, st
        self.lens = np.asarray(lens, dtype=np.int64).copy()          # tokens cached per row
        self.steps = 0

    def get_seq_length(self, layer_idx=0):
        return int(self.lens.max())

    def __len__(self):
        return self.model.config.num_hidden_layers

    def __getitem__(self, i):
        c = self.model.config
        sh = _PastShape((len(self.lens), c.num_key_value_heads, self.get_seq_length(), c.head_dim))
        return (sh, sh)

    def __iter__(self):
        return iter(self[i] for i in range(len(self)))

    def __del__(self):
        try:
            self.model._release_slot(self.st.get("slot"))
        except Exception:
            pass


_LlmConfigC = _lib.LlmConfigC


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


class ParamRef:
    """What train_multimodal.train touches of an nn.Parameter (modelcompose/train/train_multimodal.py:436-465): `.requires_grad`, read and
    written.  The tensors themselves live in the owner's reference-grammar state (`_raw`, prefix / suffix tokens) until the training step
    builds its flat fp32 master buffer from the selected set (train/step.py)."""
    __slots__ = ("_owner", "name")

    def __init__(self, owner, name):
        self._owner, self.name = owner, name

    @property
    def requires_grad(self) -> bool:
        return self._owner._requires_grad.get(self.name, True)

    @requires_grad.setter
    def requires_grad(self, flag):
        self._owner._requires_grad[self.name] = bool(flag)

    def requires_grad_(self, flag=True):
        self.requires_grad = flag
        return self

    @property
    def data(self):
        return self._owner._param_tensor(self.name)

    @property
    def shape(self):
        return self.data.shape


class ModuleDictView(dict):
    """get_modal_projectors() / get_modal_encoders(): the dict of HIP modules, answering the nn.ModuleDict calls of the reference's train()
    as well - .parameters() / .named_parameters() (projectors: the trainable tensors under model.modal_projectors.*; encoders are frozen
    and hold no trainable tensor) and .to(...) (a no-op: the modules already live on the device in the library's storage dtype)."""

    def __init__(self, modules, owner, prefix):
        super().__init__(modules)
        self._owner, self._prefix = owner, prefix

    def named_parameters(self):
        if self._owner is None:
            return
        for name, p in self._owner.named_parameters():
            if name.startswith(self._prefix + "."):
                yield name[len(self._prefix) + 1:], p

    def parameters(self):
        for _, p in self.named_parameters():
            yield p

    def to(self, *a, **k):
        return self


class MultimodalLlamaModel:
    """`.get_model()` object: embed_tokens, modal_encoders / modal_projectors dicts (multimodal_arch.py:33-63)."""

    def __init__(self, config):
        self.config = config
        self.embed_tokens = None            # [vocab, hidden] bf16 device tensor
        self.modal_encoders: Dict[str, object] = {}
        self.modal_projectors: Dict[str, object] = {}
        self._owner = None                  # weakref to the MultimodalLlamaForCausalLM that holds the tensors

    def _own(self):
        return self._owner() if self._owner is not None else None

    def get_modal_encoders(self):
        return ModuleDictView(self.modal_encoders, self._own(), "model.modal_encoders") if self.modal_encoders else None

    def get_modal_encoder(self, modal):
        return self.modal_encoders[modal]

    def get_modal_projectors(self):
        return ModuleDictView(self.modal_projectors, self._own(), "model.modal_projectors") if self.modal_projectors else None

    # ---- the train() caller's surface (modelcompose/train/train_multimodal.py:396-399, :436-465)
    def named_parameters(self):
        """parameters under `model.` (the reference's model.get_model().named_parameters(): names without that prefix)"""
        o = self._own()
        if o is None:
            return
        for name, p in o.named_parameters():
            if name.startswith("model."):
                yield name[len("model."):], p

    def parameters(self):
        for _, p in self.named_parameters():
            yield p

    def requires_grad_(self, flag=True):
        for p in self.parameters():
            p.requires_grad = flag
        return self

    def initialize_multimodal_modules(self, model_args, fsdp=None):
        """multimodal_arch.py:65-167: the config side effects (mm_*_encoder, projector types, hidden sizes, select layer / feature), the
        modal encoders and projectors built (or, when they exist, their weights loaded), and the optional projector checkpoints
        (pretrain_mm_mlp_adapter, projectors_path) read.  fsdp only changes how the reference HOLDS the encoder dict (a list wrapper that hides
        it from FSDP): nothing to mirror.  model_args: any object with the reference's ModelArguments fields."""
        import os
        from .builder import build_modal_modules
        from ..checkpoint_io import load_tensors
        o = self._own()
        if o is None:
            raise RuntimeError("initialize_multimodal_modules: this model object is not attached to a MultimodalLlamaForCausalLM")
        cfg = self.config
        g = lambda k, d=None: getattr(model_args, k, d)
        cfg.mm_vision_encoder = g("mm_vision_encoder")                              # :68-70 (mm_vision_tower is the legacy alias)
        cfg.mm_vision_tower = g("mm_vision_tower") or g("mm_vision_encoder")
        for m in ("audio", "video", "point"):
            if g(f"mm_{m}_encoder") is not None:
                setattr(cfg, f"mm_{m}_encoder", g(f"mm_{m}_encoder"))
        had = bool(self.modal_encoders)
        if g("mm_vision_encoder") is not None:                                      # :89-95
            cfg.use_mm_proj = True
            cfg.mm_projector_type = g("mm_projector_type", "linear")
            cfg.mm_vision_select_layer = g("mm_vision_select_layer", -2)
            cfg.mm_vision_select_feature = g("mm_vision_select_feature", "patch")
        for m in ("audio", "video", "point"):                                       # :97-112
            if g(f"mm_{m}_encoder") is not None:
                setattr(cfg, f"mm_{m}_projector_type", g(f"mm_{m}_projector_type", "linear"))
        if g("mm_video_encoder") is not None:
            cfg.mm_video_select_layer = g("mm_video_select_layer", -2)
            cfg.mm_video_select_feature = g("mm_video_select_feature", "patch")
        o.modal_names = infer_modals(cfg)
        if not had:                                                                  # :72-78: build; else :79-85: load_model()
            build_modal_modules(o, delay_load=True)
        for modal, enc in self.modal_encoders.items():
            enc.load_model()
            hs = getattr(enc, "hidden_size", None)
            if hs is not None:
                setattr(cfg, "mm_hidden_size" if modal == "vision" else f"mm_{modal}_hidden_size", hs)

        def get_w(weights, keyword):                                                 # :119-121
            return {k.split(keyword + ".")[1]: v for k, v in weights.items() if keyword in k}
        pre = g("pretrain_mm_mlp_adapter")
        if pre is not None:                                                          # :118-133
            w = load_tensors(pre)
            o.load_state_dict({f"model.modal_projectors.{m}.{k}": v for m in self.modal_projectors for k, v in get_w(w, f"modal_projectors.{m}").items()})
        pp = g("projectors_path")
        if pp is not None and os.path.isfile(pp):                                    # :157-167: whole non-LoRA trainable set of a saved run
            ck = load_tensors(pp)
            lora_run = any(k.startswith("base_model.model.model.") for k in ck)
            strip = "base_model.model.model." if lora_run else "model."
            o.load_state_dict({"model." + k[len(strip):]: v for k, v in ck.items() if k.startswith(strip)})
        o._dirty = True
        return self

    def get_modal_projector(self, modal):
        return self.modal_projectors[modal]


class MultimodalLlamaForCausalLM:
    config_class = MultimodalConfig

    def __init__(self, config: MultimodalConfig, device="cuda"):
        if not torch.cuda.is_available():
            raise _lib.MCError("MultimodalLlamaForCausalLM needs a HIP device: this path has no CPU fallback")
        _lib.lib()          # fail loudly when libmc_hip.so is missing
        self.config = config
        self.device = torch.device(device)
        self.dtype = BF16
        self.model = MultimodalLlamaModel(config)
        import weakref
        self.model._owner = weakref.ref(self)
        self._requires_grad: Dict[str, bool] = {}                      # train() caller's selection (ParamRef.requires_grad); default: True
        self.modal_names = infer_modals(config)                       # :631
        self.prefix_tokens: Optional[Dict[str, torch.Tensor]] = None   # :633-649
        self.suffix_tokens: Optional[Dict[str, torch.Tensor]] = None
        self._raw: Dict[str, torch.Tensor] = {}                        # host copy of llm tensors in reference key grammar
        self._handle = C.c_void_p(0)
        self._keep = []                                                # device tensors referenced by the C handle
        self._cache = {}
        self._lock = threading.RLock()                                 # see _serialised
        self._compose_events = None
        self.use_graph = True

    # ------------------------------------------------------------------ reference accessors
    def get_model(self):
        return self.model

    def get_vision_tower(self):
        return self.model.get_modal_encoders()

    def get_modal_encoders(self):
        return self.model.get_modal_encoders()

    def get_modal_encoder(self, modal):
        return self.model.get_modal_encoder(modal)

    def get_modal_projectors(self):
        return self.model.get_modal_projectors()

    def get_modal_projector(self, modal):
        return self.model.get_modal_projector(modal)

    def get_modal_processors(self):                                    # multimodal_arch.py:190-195
        return {k: getattr(v, "modal_processor", None) for k, v in self.model.modal_encoders.items()}

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def __del__(self):
        try:
            if self._handle:
                _lib.lib().mc_llm_destroy(self._handle)
        except Exception:
            pass

    # ------------------------------------------------------------------ the train() caller's surface
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, config=None, cache_dir=None, device="cuda", torch_dtype=None, **kwargs):
        """modelcompose/train/train_multimodal.py:307-325: `MultimodalLlamaForCausalLM.from_pretrained(model_base, cache_dir=...,
        lora_strategy=..., lora_r=..., lora_alpha=..., lora_dropout=..., local_prefix_tokens=..., ..., mm_vision_encoder=..., ...)`.
        As in transformers, keyword arguments that name a config attribute update the config; the base tensors are read from the
        checkpoint directory; LocalLoRA adapters the checkpoint does not hold are initialised as peft's LoraLayer.reset_lora_parameters
        does (A: kaiming_uniform(a = sqrt 5), B: zeros) for every adapter of the plan (LocalLoraLinear.__init__, multimodal_llama.py:84-107);
        prefix / suffix tokens start at zero (:633-649).  The modal encoders / projectors are built by
        get_model().initialize_multimodal_modules(model_args) afterwards, as in train().  bitsandbytes arguments (:283-305) are refused."""
        from .builder import load_base_state_dict
        if kwargs.get("load_in_4bit") or kwargs.get("load_in_8bit"):
            raise NotImplementedError("bitsandbytes quantised loading (train_multimodal.py:283-305) is out of scope of the HIP path")
        for k in ("load_in_4bit", "load_in_8bit", "quantization_config", "device_map"):
            kwargs.pop(k, None)
        if torch_dtype is not None:
            _lib.set_storage_dtype(torch_dtype)
        path = str(pretrained_model_name_or_path)
        cfg = config if config is not None else MultimodalConfig.from_pretrained(path)
        for k, v in kwargs.items():
            setattr(cfg, k, v)
        model = cls(cfg, device=device)
        model.load_state_dict(load_base_state_dict(path))
        model.reset_lora_parameters(only_missing=True)
        for which, count_key in (("prefix_tokens", "local_prefix_tokens"), ("suffix_tokens", "local_suffix_tokens")):
            if getattr(cfg, count_key, 0):
                d = getattr(model, which) or {}
                for m in infer_modals(cfg):
                    n = getattr(cfg, f"local_{m}_{which}", None)
                    n = getattr(cfg, count_key) if n is None else n
                    d.setdefault(m, torch.zeros(n, cfg.hidden_size, dtype=BF16, device=model.device))
                setattr(model, which, d)
        return model

    def reset_lora_parameters(self, only_missing: bool = True):
        """peft LoraLayer.reset_lora_parameters (0.4.0) for every adapter of the plan and every LocalLoRA linear."""
        if self.config.lora_strategy is None:
            return self
        names, _, _, _ = adapter_plan(self.config)
        r = self.config.lora_r
        for key in [k for k in self._raw if k.endswith(".weight") and ".layers." in k and (".self_attn." in k or ".mlp." in k) and ".lora_" not in k]:
            pre = key[:-len(".weight")]
            N, K = self._raw[key].shape
            for n in names:
                ka, kb = f"{pre}.lora_A.{n}.weight", f"{pre}.lora_B.{n}.weight"
                if only_missing and ka in self._raw and kb in self._raw:
                    continue
                a = torch.empty(r, K, dtype=torch.float32)
                torch.nn.init.kaiming_uniform_(a, a=math.sqrt(5))
                self._raw[ka], self._raw[kb] = a.to(self._raw[key].dtype), torch.zeros(N, r, dtype=self._raw[key].dtype)
        self._dirty = True
        return self

    def _param_tensor(self, name: str):
        if name in self._raw:
            return self._raw[name]
        which, _, m = name.partition(".")
        if which in ("prefix_tokens", "suffix_tokens") and getattr(self, which) and m in getattr(self, which):
            return getattr(self, which)[m]
        raise KeyError(name)

    def named_parameters(self):
        """(name, ParamRef) in the reference's key grammar: every tensor of the state dict that is a parameter there (base weights, LoRA
        factors, projector tensors) plus prefix_tokens.{modal} / suffix_tokens.{modal}.  The frozen encoders' tensors are not listed: no
        branch of train() makes them trainable."""
        for k in self._raw:
            yield k, ParamRef(self, k)
        for which in ("prefix_tokens", "suffix_tokens"):
            for m in (getattr(self, which) or {}):
                yield f"{which}.{m}", ParamRef(self, f"{which}.{m}")

    def parameters(self):
        for _, p in self.named_parameters():
            yield p

    def requires_grad_(self, flag=True):
        for p in self.parameters():
            p.requires_grad = flag
        return self

    def trainable_names(self):
        return [n for n, p in self.named_parameters() if p.requires_grad]

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = False):
        """Accepts the reference key grammar: base llama tensors, lora_{A,B}.{adapter}, prefix/suffix tokens,
        model.modal_projectors.*, model.modal_encoders.* .  Call finalize() afterwards (load_pretrained_model does)."""
        enc_sd: Dict[str, Dict[str, torch.Tensor]] = {}
        proj_sd: Dict[str, Dict[str, torch.Tensor]] = {}
        for k, v in sd.items():
            if k.startswith("base_model.model."):
                k = k[len("base_model.model."):]
            if k.startswith("model.modal_encoders."):
                m, rest = k[len("model.modal_encoders."):].split(".", 1)
                enc_sd.setdefault(m, {})[rest] = v
            elif k.startswith("model.modal_projectors."):
                m, rest = k[len("model.modal_projectors."):].split(".", 1)
                proj_sd.setdefault(m, {})[rest] = v
                self._raw[k] = v                                      # trainable in stage 2: the training step needs the master copy
            elif k.startswith("prefix_tokens.") or k.startswith("suffix_tokens."):
                which, m = k.split(".", 1)
                d = getattr(self, which) or {}
                d[m] = v.to(self.device, BF16).reshape(-1, v.shape[-1]).contiguous()
                setattr(self, which, d)
            else:
                self._raw[k] = v
        for m, d in proj_sd.items():
            if m in self.model.modal_projectors:
                self.model.modal_projectors[m].load_state_dict(d)
        for m, d in enc_sd.items():
            if m in self.model.modal_encoders:
                d = {k[len("vision_tower."):] if k.startswith("vision_tower.") else k: v for k, v in d.items()}
                self.model.modal_encoders[m].load_state_dict(d)
        self._dirty = True
        return self

    def _has_lora(self, prefix, key):
        return f"{prefix}.lora_A.{key}.weight" in self._raw and f"{prefix}.lora_B.{key}.weight" in self._raw

    def _compose_linear(self, prefix: str, adapter: str, out: torch.Tensor, N: int, K: int, col_scale=None, nb_stride=1, nb_offset=0, dither=False):
        """dense weight of `adapter` for one LocalLoRA linear -> packed buffer `out` (block nb at nb*nb_stride + nb_offset),
        columns multiplied by col_scale (the preceding RMSNorm's weight) before the single bf16 rounding.  dither: the unbiased rounding of
        mc_compose_weight_dither_bf16 (finalize() re-composes the adapters whose delta round-to-nearest would lose)."""
        dev = self.device
        if not dither:
            self._composed.setdefault(adapter, []).append((prefix, out, N, K, col_scale, nb_stride, nb_offset))
        w = self._raw[f"{prefix}.weight"].to(dev, BF16)
        terms = composition_terms(self.config, adapter, lambda key: self._has_lora(prefix, key))
        tl = []
        for key, scale in terms:
            if not self._has_lora(prefix, key):
                continue                                             # e.g. 'default-point' never trained: contributes B=0
            a = self._raw[f"{prefix}.lora_A.{key}.weight"].to(dev)
            b = self._raw[f"{prefix}.lora_B.{key}.weight"].to(dev)
            tl.append((a, b, scale))
        import zlib
        seed = (zlib.crc32(f"{prefix}|{adapter}".encode()) | 1) if dither else 0          # one reproducible stream of bits per (linear, adapter)
        _compose_into(w, tl, N, K, out, col_scale, nb_stride, nb_offset, retention=self._retention_parts.setdefault(adapter, []),
                      dither_seed=seed)

    def _compose_linear_multi(self, prefix: str, adapters, outs, N: int, K: int, col_scale=None, nb_stride=1, nb_offset=0):
        """The dense weights of one LocalLoRA linear for ALL routed adapters in one pass over W (round 5: W is read once, not once per
        adapter): the union of the adapters' LoRA terms, one bit mask per output."""
        dev = self.device
        w = self._raw[f"{prefix}.weight"].to(dev, BF16)
        keys, terms, masks = [], [], []
        for ad, out in zip(adapters, outs):
            self._composed.setdefault(ad, []).append((prefix, out, N, K, col_scale, nb_stride, nb_offset))
            m = 0
            for key, scale in composition_terms(self.config, ad, lambda key: self._has_lora(prefix, key)):
                if not self._has_lora(prefix, key):
                    continue                                         # e.g. 'default-point' never trained: contributes B = 0
                tk = (key, float(scale))
                if tk not in keys:
                    keys.append(tk)
                    terms.append((self._raw[f"{prefix}.lora_A.{key}.weight"].to(dev), self._raw[f"{prefix}.lora_B.{key}.weight"].to(dev), scale))
                m |= 1 << keys.index(tk)
            masks.append(m)
        if len(terms) > 8 or len(outs) > 6:                          # beyond one launch's tables: per adapter
            for ad, out in zip(adapters, outs):
                self._composed[ad].pop()
                self._compose_linear(prefix, ad, out, N, K, col_scale, nb_stride, nb_offset)
            return
        rets = [self._retention_parts.setdefault(ad, []) for ad in adapters]
        _compose_multi_into(w, terms, masks, N, K, outs, col_scale, nb_stride, nb_offset, retentions=rets, events=self._compose_events,
                            batch=getattr(self, "_compose_batch", None))
        r = max((t[0].shape[0] for t in terms), default=0)
        self.compose_bytes += 2.0 * N * K * (1 + len(outs)) + 2.0 * len(terms) * r * (N + K)

    def finalize(self):
        """Compose + pack every weight, create the C runtime handle."""
        cfg, dev = self.config, self.device
        Hd, I, Lyr = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
        H, Hkv, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        names = self.modal_names
        nA = len(names)
        raw = self._raw
        self._retention_parts = {}
        self._composed = {}                                          # adapter -> [(prefix, out, N, K, col_scale, nb_stride, nb_offset)] of this pass
        if "model.embed_tokens.weight" not in raw:
            raise ValueError("state dict lacks model.embed_tokens.weight")
        self.model.embed_tokens = raw["model.embed_tokens.weight"].to(dev, BF16).contiguous()
        qkv_n, Kp_h, Kp_i = (H + 2 * Hkv) * D, ops.ceil_to(Hd, 64), ops.ceil_to(I, 64)
        keep = []
        layer_ptrs = []
        self.compose_bytes = 0.0                                     # algorithmic HBM bytes of the composition (bench.py's compose roofline)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        self._compose_events = []
        # round 6: every linear of the model is composed by ONE launch (mc_compose_batch_bf16): the per-linear argument blocks are collected
        # here and flushed after the loop (MC_COMPOSE_BATCH=0: one launch per linear as in round 5)
        self._compose_batch = [] if os.environ.get("MC_COMPOSE_BATCH", "1") != "0" else None
        # adapters whose dense weights equal another adapter's (no LoRA terms at all) share storage with the base
        for l in range(Lyr):
            p = f"model.layers.{l}"
            # input_layernorm / post_attention_layernorm weights (fp32) folded into the q|k|v and gate|up columns
            g_in = raw[f"{p}.input_layernorm.weight"].to(dev, torch.float32).contiguous()
            g_post = raw[f"{p}.post_attention_layernorm.weight"].to(dev, torch.float32).contiguous()
            keep.extend([g_in, g_post])
            owners = []                                              # adapters that get storage of their own (the first LoRA-less one stands for all)
            base_owner = None
            owner_of = {}
            for ad in names:
                has_any = any(composition_terms(cfg, ad, lambda key, pp=f"{p}.{blk}.{lin}": self._has_lora(pp, key))
                              for blk, lins in LINEARS for lin in lins)
                if not has_any:
                    if base_owner is None:
                        base_owner = ad
                        owners.append(ad)
                    owner_of[ad] = base_owner
                else:
                    owners.append(ad)
                    owner_of[ad] = ad
            bufs = {}
            for ad in owners:
                bufs[ad] = dict(qkv=torch.empty(ops.packed_elems(qkv_n, Hd), dtype=BF16, device=dev),
                                o=torch.empty(ops.packed_elems(Hd, H * D), dtype=BF16, device=dev),
                                gu=torch.empty(ops.packed_elems(2 * I, Hd), dtype=BF16, device=dev),     # gate / up interleaved per 16-row block (gate even, up odd)
                                dn=torch.empty(ops.packed_elems(Hd, I), dtype=BF16, device=dev))
                keep.extend(bufs[ad].values())
            off = 0
            for lin, n in (("q_proj", H * D), ("k_proj", Hkv * D), ("v_proj", Hkv * D)):
                self._compose_linear_multi(f"{p}.self_attn.{lin}", owners, [bufs[ad]["qkv"][off * Kp_h:(off + n) * Kp_h] for ad in owners], n, Hd, col_scale=g_in)
                off += n
            self._compose_linear_multi(f"{p}.self_attn.o_proj", owners, [bufs[ad]["o"] for ad in owners], Hd, H * D)
            self._compose_linear_multi(f"{p}.mlp.gate_proj", owners, [bufs[ad]["gu"] for ad in owners], I, Hd, col_scale=g_post, nb_stride=2, nb_offset=0)
            self._compose_linear_multi(f"{p}.mlp.up_proj", owners, [bufs[ad]["gu"] for ad in owners], I, Hd, col_scale=g_post, nb_stride=2, nb_offset=1)
            self._compose_linear_multi(f"{p}.mlp.down_proj", owners, [bufs[ad]["dn"] for ad in owners], Hd, I)
            for ad in names:
                b = bufs[owner_of[ad]]
                layer_ptrs.extend([b["qkv"].data_ptr(), b["o"].data_ptr(), b["gu"].data_ptr(), b["dn"].data_ptr()])
        if self._compose_batch:
            _compose_flush(self._compose_batch, self._compose_events)
        self._compose_batch = None
        ev1.record()
        final_norm = raw["model.norm.weight"].to(dev, BF16).contiguous()
        self.lm_head = ops.pack_weight(raw["lm_head.weight"].to(dev))
        # rotary tables, fp32 (LlamaRotaryEmbedding 4.31: inv_freq and angles in fp32)
        n_pos = cfg.max_position_embeddings
        inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
        ang = torch.outer(torch.arange(n_pos, dtype=torch.float32), inv)
        self._cos, self._sin = ang.cos().to(dev).contiguous(), ang.sin().to(dev).contiguous()
        keep.extend([final_norm, self.lm_head.data, self._cos, self._sin, self.model.embed_tokens])
        L = _lib.lib()
        if self._handle:
            L.mc_llm_destroy(self._handle)
            self._handle = C.c_void_p(0)
        cc = _LlmConfigC(Hd, I, Lyr, H, Hkv, D, cfg.vocab_size, nA, n_pos, cfg.rms_norm_eps)
        h = C.c_void_p(0)
        _lib.check(L.mc_llm_create(C.byref(cc), C.byref(h)), "mc_llm_create")
        self._handle = h
        arr = (C.c_void_p * len(layer_ptrs))(*layer_ptrs)
        _lib.check(L.mc_llm_set_weights(h, arr, _ptr(final_norm), _ptr(self.lm_head.data), _ptr(self.model.embed_tokens),
                                        _ptr(self._cos), _ptr(self._sin)), "mc_llm_set_weights")
        _lib.check(L.mc_llm_set_option(h, b"use_graph", 1 if self.use_graph else 0), "mc_llm_set_option")
        self._keep = keep
        self._final_norm = final_norm
        self._dirty = False
        torch.cuda.synchronize()
        # device time of the composition loop (every launch of it, the small A^T transposes included; host-resident state dicts also pay their
        # H2D copies here) and its launch count: 7 per layer (one per linear, all routed adapters from one read of W)
        self.compose_ms = ev0.elapsed_time(ev1)
        self.compose_launches = len(self._compose_events)
        # the composition kernel's own time: HIP events around each of its launches (the loop above is host-bound: per linear a dozen small
        # torch ops - the A^T transposes of the LoRA factors, allocations - sit between two launches)
        self.compose_kernel_ms = sum(a.elapsed_time(b) for a, b in self._compose_events)
        self._compose_events = None
        self._summarise_delta_retention()
        return self

    # below this share of the delta surviving round-to-nearest, an adapter's weights are re-composed with the unbiased rounding
    DITHER_BELOW = 0.9

    def _summarise_delta_retention(self):
        """delta_retention[adapter] = Σ (W' - bf16(W c))·(ΔW c) / Σ (ΔW c)² over every composed linear of the adapter: the share of the
        LoRA delta that survives the single bf16 rounding of the pre-merged weight, projected on the delta itself.  1.0 for trained
        deltas (|ΔW| ~ 2^-4 |W|); it falls when |ΔW| sinks below half a bf16 step of W (0.72 at 2^-9.3 with norm weights of exactly 1: the
        base weights then sit ON the bf16 grid and small deltas round back to them; the reference's branch form keeps them).  A warning
        is raised below 0.9 - the lost part is then still under the bf16 rounding of the layer's own output, but it is a systematic
        shrink of the delta, not noise (tests/test_fullwidth_parity_gpu.py::test_small_delta_*)."""
        import warnings
        self.delta_retention = {}
        for ad, plist in self._retention_parts.items():
            if not plist:
                continue
            num = den = 0.0
            for parts in plist:
                h = parts.double().cpu().numpy()
                num += float(h[:, 0].sum())
                den += float(h[:, 1].sum())
            if den > 0:
                self.delta_retention[ad] = num / den
        self._retention_parts = {}
        low = {a: round(v, 3) for a, v in self.delta_retention.items() if v < self.DITHER_BELOW}
        self.delta_retention_rne = dict(self.delta_retention)       # what round-to-nearest would have kept
        self.delta_dithered = sorted(low)
        if low and os.environ.get("MC_COMPOSE_DITHER", "1") != "0":
            # Round 4: keep the delta instead of warning.  W + dW rounds back to W when |dW| is below half a bf16 step of W - a
            # systematic loss.  Those adapters are composed again, in place, with UNBIASED rounding (E[W'] = the fp32 composition): the delta
            # survives in expectation (retention 1.00 +- 1e-3 over the 16 M elements of a linear), the price is zero-mean rounding noise of at
            # most one bf16 step per weight - what any off-grid weight carries.
            for ad in low:
                for (prefix, out, N, K, col_scale, nb_stride, nb_offset) in self._composed.get(ad, []):
                    self._compose_linear(prefix, ad, out, N, K, col_scale, nb_stride, nb_offset, dither=True)
            torch.cuda.synchronize()
            for ad in low:
                num = den = 0.0
                for parts in self._retention_parts.get(ad, []):
                    h = parts.double().cpu().numpy()
                    num += float(h[:, 0].sum())
                    den += float(h[:, 1].sum())
                if den > 0:
                    self.delta_retention[ad] = num / den
            self._retention_parts = {}
            low = {a: round(v, 3) for a, v in self.delta_retention.items() if v < self.DITHER_BELOW}
        self._composed = {}
        if low:
            warnings.warn(f"LoRA deltas of adapters {low} are below the bf16 resolution of the base weights: only that share of them "
                          f"survives the pre-merge rounding (the reference's branch form keeps small deltas)", RuntimeWarning)

    # ------------------------------------------------------------------ encode / splice (multimodal_arch.py:197-459)
    def encode_modal_inputs(self, inputs, prefix_tokens=None, suffix_tokens=None):
        """Per modality: encoder -> projector -> cat(prefix, feat, suffix) (multimodal_arch.py:197-268).  Absent
        modalities are skipped: the reference runs their encoder on zeros only as a ZeRO-3 workaround (:203-206)
        and never uses the result."""
        feats, masks = {}, {}
        t_issue0 = time.perf_counter()
        present = [m for m in self.modal_names if m != "default" and m in inputs]
        # Round 4 (`encode_streams`, default on; MC_ENC_STREAMS=0 / model.encode_streams = False: one stream): the towers are independent, and the
        # short ones run many launches of one or two rounds of tiles (BEATs: 288-384 tiles per GEMM) - on side streams their launches fill each
        # other's partial rounds: 128.2 -> 123.5 ms for image + audio + video at B = 48 (in-process A/B, profiles/r04_probes/encode_streams_ab.json).
        # Results are the same bits: every launch computes what it computes alone.
        use_streams = bool(present) and getattr(self, "encode_streams", os.environ.get("MC_ENC_STREAMS", "1") != "0") and \
            torch.cuda.is_current_stream_capturing() is False
        # (`encode_split` / MC_ENC_SPLIT = n > 1: every tower's batch in n parts on streams of their own, so that two half-batch chains fill each
        # other's partly filled last rounds of tiles - measured and OFF: 127.9 ms one stream, 124-126 towers on side streams, **141.6** with halves,
        # 132.2 with quarters (same bits): two launches of one GEMM interleaved on an XCD lose the raster's operand sharing in L2.)
        nsplit = max(1, int(getattr(self, "encode_split", os.environ.get("MC_ENC_SPLIT", "1")))) if use_streams else 1
        work = []                                                  # (modal, chunk of the modality's input)
        for modal in present:
            x = inputs[modal]
            xb = x if torch.is_tensor(x) else (next(iter(x.values())) if isinstance(x, dict) and x else None)
            nb = int(xb.shape[0]) if torch.is_tensor(xb) else 0
            # (PointBERT keeps a per-sample FPS start index on the encoder: its batch stays whole)
            if nsplit > 1 and modal != "point" and nb >= 8 * nsplit and (torch.is_tensor(x) or (isinstance(x, dict) and all(torch.is_tensor(v) and v.shape[0] == nb for v in x.values()))):
                cuts = [nb * i // nsplit for i in range(nsplit + 1)]
                for i0, i1 in zip(cuts[:-1], cuts[1:]):
                    work.append((modal, x[i0:i1] if torch.is_tensor(x) else {k: v[i0:i1] for k, v in x.items()}))
            else:
                work.append((modal, x))
        if not use_streams or len(work) < 2:
            for modal, x in work:
                feats[modal] = self._encode_one(modal, x, prefix_tokens, suffix_tokens)
        else:
            cur = torch.cuda.current_stream()
            pool = _shared_streams(self.device, "encode", len(work))
            parts = {}
            for (modal, x), st in zip(work, pool):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    f = self._encode_one(modal, x, prefix_tokens, suffix_tokens)
                f.record_stream(cur)
                parts.setdefault(modal, []).append(f)
            for st in pool[:len(work)]:
                cur.wait_stream(st)
            for modal, fl in parts.items():
                feats[modal] = fl[0] if len(fl) == 1 else torch.cat(fl, dim=0)
        for modal, f in feats.items():
            masks[modal] = torch.ones(f.shape[0], f.shape[1], device=f.device)
        # host time spent ISSUING the towers' launches (several hundred per batch, from Python): bench.py reports it per batch so that a
        # slowdown with 8 ranks on one host can be attributed
        self.last_encode_issue_ms = (time.perf_counter() - t_issue0) * 1e3
        # deferred input checks of the encoders (BEATs: a device padding mask that is not a suffix) are raised HERE, for every caller of this
        # public method - not only for the ones that go on to _plan().  The read synchronises with the towers; generate() and
        # forward() copy the ids to the host right after this call anyway, so the wait only moves.
        if not torch.cuda.is_current_stream_capturing():
            for modal in present:
                chk = getattr(self.model.modal_encoders.get(modal), "check_pending", None)
                if chk is not None:
                    chk()
        return feats, masks

    def _encode_one(self, modal, x, prefix_tokens, suffix_tokens):
        """One modality: encoder -> projector -> cat(prefix, feat, suffix) on the current stream.  Every linear runs the tile GEMM family
        whatever its row count (B x tokens; the Q-Former's 32 queries of ONE sample are fewer than 64 rows): a sample's features are the
        same bits in a batch of 1 and in a batch of 48 (ops.gemm_family)."""
        with ops.gemm_family("tile"):
            return self._encode_one_impl(modal, x, prefix_tokens, suffix_tokens)

    def _encode_one_impl(self, modal, x, prefix_tokens, suffix_tokens):
        encoder, projector = self.model.get_modal_encoder(modal), self.model.get_modal_projector(modal)
        if type(x) is list:
            if modal == "audio":
                x = torch.stack(x, dim=0)                          # :215
            else:
                raise ValueError("list-of-tensors inputs are only defined for audio in the reference (:224-230 raises)")
        if modal == "audio" and isinstance(x, dict):
            f = encoder(**x)
            f = f[0] if isinstance(f, tuple) else f                # :233-235
            f = projector(f)
        elif modal == "video":
            f = encoder(x)                                         # b t n d
            b, t, n, d = f.shape
            f = projector(f.reshape(b, t * n, d))                  # :236-240
        else:
            f = projector(encoder(x))
        f = f.to(BF16)
        b = f.shape[0]
        parts = []
        if prefix_tokens is not None and modal in prefix_tokens:
            parts.append(prefix_tokens[modal].view(1, -1, f.shape[-1]).expand(b, -1, -1))
        parts.append(f)
        if suffix_tokens is not None and modal in suffix_tokens:
            parts.append(suffix_tokens[modal].view(1, -1, f.shape[-1]).expand(b, -1, -1))
        if len(parts) > 1:
            f = _cat_rows(parts)
        return f.contiguous()

    def _plan(self, input_ids, attention_mask, labels, modal_inputs, feats) -> SplicePlan:
        ids = input_ids.detach().cpu().numpy()
        # the copy above synchronised with the encoders: deferred input checks of theirs are read here at no extra cost
        for enc in self.model.modal_encoders.values():
            chk = getattr(enc, "check_pending", None)
            if chk is not None:
                chk()
        am = None if attention_mask is None else attention_mask.detach().cpu().numpy().astype(bool)
        lb = None if labels is None else labels.detach().cpu().numpy()
        keys = list(modal_inputs.keys())
        return plan_splice(ids, am, lb, keys, {m: feats[m].shape[1] for m in feats}, {m: feats[m].shape[0] for m in feats})

    def _gather_rows(self, plan: SplicePlan, feats, order_b, order_t, out: torch.Tensor):
        """Fill out[r] for r in range(len(order_b)) with the embedding / feature row of token (order_b[r], order_t[r])."""
        dev = self.device
        tok = plan.tok_id[order_b, order_t]
        sm = plan.src_modal[order_b, order_t]
        sr = plan.src_row[order_b, order_t]
        pad = np.nonzero((sm < 0) & (tok < 0))[0]                  # slots behind a sample's spliced length (padded layouts): zero rows
        if len(pad):
            ops.copy_rows(out, out, len(pad), torch.full((len(pad),), -1, dtype=torch.int32, device=dev),
                          torch.from_numpy(pad.astype(np.int32)).to(dev))
        text = np.nonzero((sm < 0) & (tok >= 0))[0]
        if len(text):
            # nn.Embedding raises on an id outside the table (embed_tokens, multimodal_llama.py:505-520); the gather kernel would read past it
            hi = int(tok[text].max())
            if hi >= self.model.embed_tokens.shape[0]:
                raise IndexError(f"token id {hi} is out of range for the embedding table of {self.model.embed_tokens.shape[0]} rows")
            ids = torch.from_numpy(tok[text]).to(dev)
            ops.embed_rows(self.model.embed_tokens, ids, out, torch.from_numpy(text.astype(np.int32)).to(dev))
        for i, m in enumerate(plan.modal_order):
            sel = np.nonzero(sm == i)[0]
            if len(sel) == 0:
                continue
            f = feats[m].view(-1, feats[m].shape[-1])
            ops.copy_rows(f, out, len(sel), torch.from_numpy(sr[sel].astype(np.int32)).to(dev),
                          torch.from_numpy(sel.astype(np.int32)).to(dev))
        return out

    def prepare_inputs_labels_for_multimodal(self, input_ids, attention_mask, past_key_values, labels, modal_inputs,
                                             prefix_tokens, suffix_tokens):
        """Reference-shaped splice (multimodal_arch.py:287-459): returns
        (None, attention_mask, past_key_values, inputs_embeds (B, L, hidden), labels, modal_attention_mask)."""
        if modal_inputs is None or input_ids.shape[1] == 1:                                 # :290-293
            if past_key_values is not None and modal_inputs is not None and input_ids.shape[1] == 1:
                attention_mask = torch.ones((attention_mask.shape[0], _past_len(past_key_values) + 1), dtype=attention_mask.dtype,
                                            device=attention_mask.device)
            return input_ids, attention_mask, past_key_values, None, labels, None
        feats, _ = self.encode_modal_inputs(modal_inputs, prefix_tokens, suffix_tokens)
        plan = self._plan(input_ids, attention_mask, labels, modal_inputs, feats)
        B, Lmax = plan.tok_id.shape
        emb = torch.zeros(B * Lmax, self.config.hidden_size, dtype=BF16, device=self.device)
        bb, tt = np.nonzero(np.arange(Lmax)[None, :] < plan.lens[:, None])
        rows = torch.empty(len(bb), self.config.hidden_size, dtype=BF16, device=self.device)
        self._gather_rows(plan, feats, bb, tt, rows)
        ops.copy_rows(rows, emb, len(bb), None, torch.from_numpy((bb * Lmax + tt).astype(np.int32)).to(self.device))
        dev = input_ids.device
        am = torch.from_numpy(plan.attention_mask).to(dev) if attention_mask is not None else None
        lab = torch.from_numpy(plan.labels).to(dev) if labels is not None else None
        mam = {k: torch.from_numpy(v).to(dev) for k, v in plan.modal_masks.items()} or None
        return None, am, past_key_values, emb.view(B, Lmax, -1), lab, mam

    # ------------------------------------------------------------------ device forward
    def _buffers(self, B, Smax, M, Lq, slot=0):
        """KV cache + workspace of generation pipeline `slot` (generate_pipelined alternates two slots; everything a captured decode
        graph points at lives here, so every slot keeps replaying its own graph)."""
        cfg, dev = self.config, self.device
        key = ("kv", slot, B, Smax)
        if key not in self._cache:
            for k in [k for k in self._cache if k[0] == "kv" and k[1] == slot]:
                del self._cache[k]
            shape = (cfg.num_hidden_layers, B, cfg.num_key_value_heads, Smax, cfg.head_dim)
            with torch.inference_mode(False):                      # cached buffers outlive the call: never inference tensors
                self._cache[key] = (torch.zeros(shape, dtype=BF16, device=dev), torch.zeros(shape, dtype=BF16, device=dev))
        nbytes = C.c_int64(0)
        _lib.check(_lib.lib().mc_llm_workspace_bytes(self._handle, M, B, Lq, C.byref(nbytes)), "mc_llm_workspace_bytes")
        ws = self._cache.get(("ws", slot))
        if ws is None or ws.numel() < nbytes.value:
            with torch.inference_mode(False):
                ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
            self._cache[("ws", slot)] = ws
        return self._cache[key], ws

    def _prefill(self, plan: SplicePlan, feats, max_new_tokens: int, want_hidden=False, want_logits=True, slot=0, capture=(False, False)):
        if getattr(self, "_dirty", True):
            self.finalize()
        cfg, dev = self.config, self.device
        key_valid = None
        if not plan.mask_is_suffix:
            # zeros that are not a suffix (left padding, holes): the reference's own semantics (multimodal_llama.py:526-531, :543-545) -
            # every slot keeps its row and its position id (positions ignore padding), masked slots are hidden as KEYS from every query
            # by the additive padding mask.  Here: all slots are rows, the mask goes to the attention kernels as per-key validity bytes
            # (round 3; right-padded batches keep the cheaper one-length-per-row description, which also drops the padded rows).
            plan.valid_lens = plan.lens.astype(np.int32).copy()
            key_valid = plan.attention_mask
        if int(plan.valid_lens.min()) < 1:
            raise ValueError("a sample of the batch has no attended token")
        routed = cfg.lora_strategy in ("modal", "modal+language") and bool(plan.modal_masks)     # :703-704
        lay = routed_layout(plan, {m: self.modal_names.index(m) for m in plan.modal_order}, routed)
        B, Lmax, M = plan.B, plan.Lmax, lay.M
        if Lmax + max_new_tokens > cfg.max_position_embeddings:
            raise ValueError(f"sequence length {Lmax}+{max_new_tokens} exceeds max_position_embeddings {cfg.max_position_embeddings}")
        Smax = ops.ceil_to(Lmax + max_new_tokens, 64)
        (kc, vc), ws = self._buffers(B, Smax, M, Lmax, slot)
        kmask = None
        if key_valid is not None:
            kv_np = np.ones((B, Smax), dtype=np.uint8)                     # generated positions are always attended
            kv_np[:, :Lmax] = key_valid.astype(np.uint8)
            for b in range(B):
                # a row shorter than Lmax appends its generated tokens at lens[b] + s (valid_lens): the splice's right-pad zeros of the
                # mask over [lens[b], Lmax) must not hide them (mixed spliced lengths in a left-padded batch)
                kv_np[b, int(plan.lens[b]):] = 1
            kmask = torch.from_numpy(kv_np).to(dev)
        x = torch.empty(M, cfg.hidden_size, dtype=BF16, device=dev)
        self._gather_rows(plan, feats, lay.order_b, lay.order_t, x)
        i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
        row_b, row_t = i32(lay.order_b), i32(lay.order_t)
        out_map, kv_lens, last_rows = i32(lay.out_map), i32(plan.valid_lens), i32(lay.last_rows)
        hidden = torch.empty(M, cfg.hidden_size, dtype=BF16, device=dev) if want_hidden else None
        logits = torch.empty(B, cfg.vocab_size, dtype=torch.float32, device=dev) if want_logits else None
        next_ids = self._cache.get(("next_ids", slot, B))          # persistent: the decode graph of this slot points at it
        if next_ids is None:
            with torch.inference_mode(False):
                next_ids = torch.empty(B, dtype=torch.int64, device=dev)
            self._cache[("next_ids", slot, B)] = next_ids
        gs = np.ascontiguousarray(lay.group_start, dtype=np.int32)
        ga = np.ascontiguousarray(lay.group_adapter, dtype=np.int32)
        # generate() reads the last layer's output for the last token of every sequence only: when those tokens share one adapter (they are
        # text tokens of the prompt's tail) the runtime runs that layer's attention + MLP for them alone ("tail_adapter"; forward() needs
        # every row)
        tail = -1
        if kmask is not None:
            _lib.check(_lib.lib().mc_llm_set_key_mask(self._handle, _ptr(kmask), kmask.stride(0)), "mc_llm_set_key_mask")
        if kmask is None and not want_hidden and getattr(self, "last_layer_tail", os.environ.get("MC_LAST_LAYER_TAIL", "1") != "0"):
            g_of_last = np.searchsorted(gs, np.asarray(lay.last_rows), side="right") - 1
            ads = np.unique(ga[np.clip(g_of_last, 0, len(ga) - 1)])
            if len(ads) == 1:
                tail = int(ads[0])
        _lib.check(_lib.lib().mc_llm_set_option(self._handle, b"tail_adapter", tail), "mc_llm_set_option")
        cap_h = cap_q = None
        if capture[0]:                                            # forward(output_hidden_states): n_layers + 1 snapshots of the routed rows
            cap_h = torch.empty(cfg.num_hidden_layers + 1, M, cfg.hidden_size, dtype=BF16, device=dev)
        if capture[1]:                                            # forward(output_attentions): every layer's rotated queries, sequence order
            cap_q = torch.empty(cfg.num_hidden_layers, B * Lmax, cfg.num_attention_heads * cfg.head_dim, dtype=BF16, device=dev)
        if cap_h is not None or cap_q is not None:
            _lib.check(_lib.lib().mc_llm_set_capture(self._handle, _ptr(cap_h), _ptr(cap_q)), "mc_llm_set_capture")
        _lib.check(_lib.lib().mc_llm_prefill(self._handle, _ptr(x), M, len(ga), gs.ctypes.data_as(C.c_void_p), ga.ctypes.data_as(C.c_void_p),
                                             _ptr(row_b), _ptr(row_t), _ptr(row_t), _ptr(out_map), _ptr(kv_lens), _ptr(last_rows), B, Lmax,
                                             _ptr(kc), _ptr(vc), Smax, _ptr(ws), _ptr(hidden), _ptr(logits), _ptr(next_ids), _stream()),
                   "mc_llm_prefill")
        return dict(plan=plan, layout=lay, kc=kc, vc=vc, ws=ws, Smax=Smax, logits=logits, next_ids=next_ids, hidden=hidden,
                    kv_lens=kv_lens, out_map=out_map, slot=slot, key_valid=kmask, cap_hidden=cap_h, cap_q=cap_q)

    def _decode(self, st, n_steps: int, out_ids: torch.Tensor, step0: int, want_logits=False):
        B = st["plan"].B
        dev = self.device
        slot = st.get("slot", 0)
        state = self._cache.get(("state", slot, B))
        if state is None:
            with torch.inference_mode(False):
                state = torch.zeros(4 * B + 4, dtype=torch.int32, device=dev)
            self._cache[("state", slot, B)] = state
        L = _lib.lib()
        _lib.check(L.mc_decode_state_init(_ptr(state), _ptr(st["kv_lens"]), B, step0, _stream()), "mc_decode_state_init")
        smp = st.get("sampling")
        if smp is not None:                                       # seed: 2 x u32 behind the step counter of the device state
            lo, hi = smp[3] & 0xFFFFFFFF, (smp[3] >> 32) & 0xFFFFFFFF
            state[4 * B + 1:4 * B + 3] = torch.tensor([lo - (1 << 32) if lo >= 1 << 31 else lo, hi - (1 << 32) if hi >= 1 << 31 else hi],
                                                      dtype=torch.int32)
        _lib.check(L.mc_llm_set_sampling(self._handle, int(smp is not None), *(smp[:3] if smp else (1.0, 0, 1.0))), "mc_llm_set_sampling")
        logits = torch.empty(n_steps, B, self.config.vocab_size, dtype=torch.float32, device=dev) if want_logits else None
        # the workspace of the prefill is at least as large as the decode one (M >= B)
        kv_len_max = int(st["plan"].valid_lens.max()) + step0    # keys already cached by the longest sequence
        if st.get("key_valid") is not None and st.get("mask_in_decode"):
            # the batch's padding mask stays in force for its decode steps (HF extends it by ones) - one-shot per call, the call then runs
            # without graph replay
            _lib.check(L.mc_llm_set_key_mask(self._handle, _ptr(st["key_valid"]), st["key_valid"].stride(0)), "mc_llm_set_key_mask")
        _lib.check(L.mc_llm_decode(self._handle, B, n_steps, _ptr(st["next_ids"]), _ptr(out_ids), out_ids.stride(0), _ptr(state),
                                   _ptr(st["kc"]), _ptr(st["vc"]), st["Smax"], kv_len_max, _ptr(st["ws"]), _ptr(logits), _stream()),
                   "mc_llm_decode")
        return logits

    def runtime_option(self, name: str) -> int:
        """mc_llm_get_option: 'graph_active' (the last decode call replayed a hipGraph), 'graph_captures', 'graph_failures', 'use_graph'."""
        v = C.c_int(0)
        _lib.check(_lib.lib().mc_llm_get_option(self._handle, name.encode(), C.byref(v)), "mc_llm_get_option")
        return v.value

    # ------------------------------------------------------------------ public API
    @_serialised
    def forward(self, input_ids=None, attention_mask=None, past_key_values=None, inputs_embeds=None, labels=None, use_cache=None,
                output_attentions=None, output_hidden_states=None, modal_inputs=None, return_dict=None, *, modal_attention_mask=None,
                cache_reserve=None):
        """MultimodalLlamaForCausalLM.forward (multimodal_llama.py:676-745) incl. its cached-decoding contract (:676-688, :747-767):

          * no past_key_values: prefill over input_ids (+ modal_inputs spliced in at the sentinels) or over inputs_embeds; logits for
            every position; with use_cache=True the result carries a HipPastKeyValues handle on the batch's KV cache;
          * past_key_values given: ONE new token per row (input_ids (B, 1), or (B, n) whose last column is taken, as
            prepare_inputs_for_generation hands them over) is appended to the cache and its logits (B, 1, V) returned - the step an
            external loop (HF generate, a server's own scheduler) drives; modal_inputs are ignored then, as in the reference (:290-293).

        inputs_embeds (B, L, hidden): the reference routes by modal_attention_mask, which its ForCausalLM.forward leaves unbound on this
        entry; here `modal_attention_mask` ({modal: bool (B, L)}, optional) routes the rows, default = every row on
        the `default` adapter.  cache_reserve: tokens of KV capacity beyond the prompt (default 256; the cache grows when it fills)."""
        cfg_ = self.config
        output_attentions = bool(getattr(cfg_, "output_attentions", False) if output_attentions is None else output_attentions)          # :687-690
        output_hidden_states = bool(getattr(cfg_, "output_hidden_states", False) if output_hidden_states is None else output_hidden_states)
        V = self.config.vocab_size
        if past_key_values is not None:
            if output_attentions or output_hidden_states:
                raise NotImplementedError("a cached decode step returns logits only: per-layer hidden states / attention maps are captured "
                                          "by the prefill call (forward without past_key_values)")
            if not isinstance(past_key_values, HipPastKeyValues) or past_key_values.model is not self:
                raise TypeError("past_key_values must be the HipPastKeyValues returned by this model's forward(use_cache=True)")
            if labels is not None or inputs_embeds is not None:
                raise NotImplementedError("a cached step takes input_ids only")
            if input_ids is None or input_ids.dim() != 2 or input_ids.shape[0] != len(past_key_values.lens):
                raise ValueError("a cached step needs input_ids (B, 1) for the cache's B rows")
            logits = self._cached_step(past_key_values, input_ids[:, -1])
            return CausalLMOutputWithPast(loss=None, logits=logits.view(-1, 1, V), past_key_values=past_key_values)
        if (input_ids is None) == (inputs_embeds is None):
            raise ValueError("You have to specify either decoder_input_ids or decoder_inputs_embeds")
        reserve = 0
        slot = 0
        mask_in_decode = True                                     # see generate()
        if use_cache:
            reserve = 256 if cache_reserve is None else int(cache_reserve)
            slot = self._new_slot()
        if inputs_embeds is not None:
            if labels is not None:
                raise NotImplementedError("labels with inputs_embeds")
            plan, feats = self._plan_from_embeds(inputs_embeds, attention_mask, modal_attention_mask)
        else:
            modal_inputs = modal_inputs or {}
            feats, _ = self.encode_modal_inputs(modal_inputs, self.prefix_tokens, self.suffix_tokens)
            plan = self._plan(input_ids, attention_mask, labels, modal_inputs, feats)
        reserve = max(0, min(reserve, self.config.max_position_embeddings - plan.Lmax))       # the rotary table bounds the cache
        st = self._prefill(plan, feats, reserve, want_hidden=True, want_logits=False, slot=slot, capture=(output_hidden_states, output_attentions))
        st["mask_in_decode"] = mask_in_decode
        B, Lmax = plan.B, plan.Lmax
        hidden_states = attentions = raw_last = None
        if output_hidden_states:
            # the reference's tuple (multimodal_llama.py:561-604): the input of every decoder layer, then the final norm of the last layer's
            # output - n_layers + 1 tensors (B, L, hidden); padded slots are zero rows
            seq = lambda rows: self._rows_to_sequence(rows, st["out_map"], B, Lmax)
            hs = [seq(st["cap_hidden"][l]) for l in range(self.config.num_hidden_layers)]
            raw_last = seq(st["cap_hidden"][self.config.num_hidden_layers])
            hidden_states = tuple(hs) + (seq(st["hidden"]),)
        if output_attentions:
            # `attn_weights` of every layer (:295-312): softmax over the scaled, masked scores, (B, H, L, L) in the model dtype.  The flash
            # kernels never form them; they are recomputed from the layer's rotated queries (captured) and its keys (in the KV cache)
            c = self.config
            H, Hkv, D = c.num_attention_heads, c.num_key_value_heads, c.head_dim
            S_ = st["Smax"]
            atts = []
            for l in range(c.num_hidden_layers):
                atts.append(ops.attn_probs(st["cap_q"][l], st["kc"][l], B, H, Hkv, Lmax, Lmax, D, (Lmax * H * D, H * D, D), (Hkv * S_ * D, D, S_ * D),
                                           causal=True, kv_lens=st["kv_lens"], key_valid=st["key_valid"]))
            attentions = tuple(atts)
        lg_r = ops.linear(st["hidden"], self.lm_head, out_f32=True)                          # lm_head (:720), routed order
        logits = torch.zeros(B * Lmax, V, dtype=torch.float32, device=self.device)
        valid = st["out_map"] >= 0
        logits[valid] = lg_r[st["out_map"][valid].long()]
        logits = logits.view(B, Lmax, V)
        loss = None
        if labels is not None:                                                                # :722-733: shifted CE, mean over kept targets
            lab = torch.from_numpy(plan.labels).to(self.device)
            tgt = torch.full((B, Lmax), IGNORE_INDEX, dtype=torch.int64, device=self.device)
            tgt[:, :-1] = lab[:, 1:]
            rows, _ = ops.ce_loss(logits.view(B * Lmax, V), tgt.view(-1), 1.0, want_grad=False)
            loss = rows.sum() / (tgt != IGNORE_INDEX).sum()                                   # nan when every target is ignored, as torch's CE
        pkv = HipPastKeyValues(self, st, plan.valid_lens) if use_cache else None
        out = CausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=pkv, hidden_states=hidden_states, attentions=attentions)
        out.raw_last_hidden_state = raw_last                      # (extra, not in the reference's output): the last layer's output before the final norm
        return out

    def _rows_to_sequence(self, rows: torch.Tensor, out_map: torch.Tensor, B: int, Lmax: int) -> torch.Tensor:
        """routed rows [M, hidden] -> (B, Lmax, hidden) in sequence order (slot without a row: zeros)."""
        out = torch.empty(B * Lmax, rows.shape[-1], dtype=rows.dtype, device=rows.device)
        ops.copy_rows(rows, out, B * Lmax, src_idx=out_map)
        return out.view(B, Lmax, -1)

    # ---- cached decoding for external loops ----------------------------------------------------------------------------------
    def _new_slot(self):
        """A private buffer slot (KV cache, workspace, decode state) for one HipPastKeyValues; generate() keeps slots 0 / 1."""
        n = self._cache.get(("ext_slots",), 0) + 1
        self._cache[("ext_slots",)] = n
        return ("ext", n)

    def _release_slot(self, slot):
        if isinstance(slot, tuple) and slot and slot[0] == "ext":
            # the slot's own buffers and those of its admission prefills (serve.ContinuousBatcher: slot ("adm", slot))
            for k in [k for k in self._cache if isinstance(k, tuple) and len(k) > 1 and k[1] in (slot, ("adm", slot))]:
                del self._cache[k]

    def _plan_from_embeds(self, inputs_embeds, attention_mask, modal_attention_mask):
        """A splice plan whose rows are the rows of inputs_embeds: every token is a "feature row" b * L + t of its routing group."""
        E = inputs_embeds.to(self.device, BF16).contiguous()
        B, L, Hd = E.shape
        if Hd != self.config.hidden_size:
            raise ValueError(f"inputs_embeds has width {Hd}, the model {self.config.hidden_size}")
        am = np.ones((B, L), dtype=bool) if attention_mask is None else attention_mask.detach().cpu().numpy().astype(bool)
        mam = {k: v.detach().cpu().numpy().astype(bool) for k, v in (modal_attention_mask or {}).items() if k != "default"}
        for k in mam:
            if k not in self.modal_names:
                raise ValueError(f"modal_attention_mask names '{k}', the model's adapters are {self.modal_names}")
        order = [m for m in self.modal_names if m in mam] + ["default"]
        src_modal = np.full((B, L), len(order) - 1, dtype=np.int8)
        for i, m in enumerate(order[:-1]):
            src_modal[mam[m]] = i
        n_att = am.sum(1)
        valid_lens = n_att.astype(np.int32)
        suffix_only = bool(all(am[b, :n_att[b]].all() for b in range(B)))
        masks = {m: src_modal == i for i, m in enumerate(order)}
        src_row = (np.arange(B)[:, None] * L + np.arange(L)[None, :]).astype(np.int32)
        plan = SplicePlan(B, L, np.full(B, L, dtype=np.int32), L, np.full((B, L), -1, dtype=np.int64), src_modal, src_row, None, am, masks,
                          order, {m: 0 for m in order}, valid_lens, suffix_only)
        return plan, {m: E for m in order}

    def _cached_step(self, pkv: "HipPastKeyValues", token_ids: torch.Tensor) -> torch.Tensor:
        """Append one token per row to the cache of `pkv` and return its logits (B, V) fp32: one decode step of the runtime (the launch
        sequence generate() replays from a graph), driven from outside."""
        st = pkv.st
        B = len(pkv.lens)
        need = int(pkv.lens.max()) + 1
        if need > st["Smax"]:
            self._grow_cache(st, need)
        if need > self.config.max_position_embeddings:
            raise ValueError(f"sequence length {need} exceeds max_position_embeddings {self.config.max_position_embeddings}")
        tid = token_ids.to(self.device, torch.int64).reshape(B)
        lo_hi = torch.stack([tid.min(), tid.max()]).tolist()           # (the caller's loop synchronises per token anyway: it reads the logits)
        if lo_hi[0] < 0 or lo_hi[1] >= self.model.embed_tokens.shape[0]:
            raise IndexError(f"token id {lo_hi[1] if lo_hi[1] >= self.model.embed_tokens.shape[0] else lo_hi[0]} is out of range for the embedding table of "
                             f"{self.model.embed_tokens.shape[0]} rows")
        st["next_ids"].copy_(tid)
        scratch = self._cache.get(("step_out", st["slot"], B))
        if scratch is None:
            with torch.inference_mode(False):
                scratch = torch.zeros(B, 1, dtype=torch.int64, device=self.device)
            self._cache[("step_out", st["slot"], B)] = scratch
        lg = self._decode(st, 1, scratch, pkv.steps, want_logits=True)
        pkv.steps += 1
        pkv.lens += 1
        return lg[0]

    def _grow_cache(self, st, need):
        """KV capacity exhausted: a larger cache, the cached keys copied over (device-to-device), the old one dropped."""
        cfg = self.config
        old_k, old_v, old_S = st["kc"], st["vc"], st["Smax"]
        new_S = ops.ceil_to(max(need + 256, old_S + old_S // 2), 64)
        new_S = min(new_S, ops.ceil_to(cfg.max_position_embeddings, 64))
        shape = (cfg.num_hidden_layers, old_k.shape[1], cfg.num_key_value_heads, new_S, cfg.head_dim)
        with torch.inference_mode(False):
            kc, vc = torch.zeros(shape, dtype=BF16, device=self.device), torch.zeros(shape, dtype=BF16, device=self.device)
        kc[:, :, :, :old_S].copy_(old_k)
        vc[:, :, :, :old_S].copy_(old_v)
        for k in [k for k in self._cache if isinstance(k, tuple) and k[0] == "kv" and k[1] == st["slot"]]:
            del self._cache[k]
        self._cache[("kv", st["slot"], old_k.shape[1], new_S)] = (kc, vc)
        st["kc"], st["vc"], st["Smax"] = kc, vc, new_S
        if st.get("key_valid") is not None:                           # the key mask covers cache positions: generated ones are attended
            km = torch.ones(old_k.shape[1], new_S, dtype=torch.uint8, device=self.device)
            km[:, :old_S].copy_(st["key_valid"])
            st["key_valid"] = km

    __call__ = forward

    @_serialised
    @torch.no_grad()
    def generate(self, input_ids=None, modal_inputs=None, do_sample=False, temperature=None, top_p=None, num_beams=1,
                 max_new_tokens=128, use_cache=True, attention_mask=None, ignore_eos=False, return_step_logits=False, slot=0,
                 prefill_done=None, **kw):
        """Greedy or sampled generation (model_multimodal_qa_loader.py:94-102).  Returns LongTensor (B, L_text + n_new): the text-length
        prompt followed by the new ids, rows that hit EOS are padded with pad_token_id (transformers greedy_search / sample).
        do_sample=True applies transformers 4.31's warpers in their order (temperature, top_k - GenerationConfig default 50, pass
        top_k=0 to disable - then top_p) and draws with a Philox stream keyed by `seed` (kwarg; default: drawn from torch's global
        generator, so torch.manual_seed makes runs repeatable).  num_beams > 1: beam search (_beam_search; kwargs length_penalty, early_stopping)."""
        if num_beams not in (None, 1):
            if do_sample or kw.get("streamer") is not None or kw.get("stopping_criteria") or kw.get("forced_ids") is not None or return_step_logits:
                raise NotImplementedError("beam search runs greedy-scored, without streamer / stopping criteria / step logits")
            return self._beam_search(input_ids, modal_inputs or {}, attention_mask, int(num_beams), max_new_tokens,
                                     float(kw.pop("length_penalty", 1.0)), bool(kw.pop("early_stopping", False)), ignore_eos)
        sampling = None
        if do_sample:
            T = 1.0 if temperature is None else float(temperature)
            P = 1.0 if top_p is None else float(top_p)
            Kk = kw.pop("top_k", 50)
            Kk = 0 if Kk is None else int(Kk)
            if not T > 0:                                                                     # logits_process.py TemperatureLogitsWarper
                raise ValueError(f"`temperature` (={temperature}) has to be a strictly positive float, otherwise your next token scores will be invalid.")
            if P < 0 or P > 1.0:
                raise ValueError(f"`top_p` has to be a float > 0 and < 1, but is {top_p}")
            if Kk < 0:
                raise ValueError(f"`top_k` has to be a strictly positive integer, but is {Kk}")
            seed = kw.pop("seed", None)
            if seed is None:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            sampling = (T, Kk, P, int(seed))
        streamer = kw.pop("streamer", None)
        criteria = kw.pop("stopping_criteria", None)
        # forced_ids (B, >= max_new_tokens - 1), optional: TEACHER FORCING - decode step s is fed forced_ids[:, s] instead of the token the
        # model chose, so every step's logits are conditioned on a given history (parity tests compare all steps with an oracle even
        # after a near-tie departure); the returned ids are still the model's own choices, one launch sequence per token
        forced = kw.pop("forced_ids", None)
        # stage_events (dict, optional): receives torch.cuda.Event pairs recorded on the current stream around the three stages, as
        # {"encode": (e0, e1), "prefill": (e1, e2), "decode": (e2, e3)} (bench.py: stage times and the decode roofline)
        stage_events = kw.pop("stage_events", None)

        def mark():
            if stage_events is None:
                return None
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        if input_ids is None:
            raise ValueError("generate() needs input_ids")
        # A prompt mask with zeros that are not a suffix stays in force over the decode steps (HF's loop: the prompt's mask extended by
        # ones).  DELIBERATE DEVIATION: when modal_inputs is passed, the reference replaces the mask by all ones on every decode step
        # (multimodal_arch.py:290-293), which makes the left pads visible again - and what sits at those positions is the output of rows
        # that could see no key at all (uniform attention over causally-masked FUTURE tokens in the reference, zeros here): garbage by
        # construction.  The mask is kept instead; tests pin this against the oracle run with keep_mask=True.
        mask_in_decode = True
        modal_inputs = modal_inputs or {}
        e0 = mark()
        feats, _ = self.encode_modal_inputs(modal_inputs, self.prefix_tokens, self.suffix_tokens)
        e1 = mark()
        plan = self._plan(input_ids, attention_mask, None, modal_inputs, feats)
        st = self._prefill(plan, feats, max_new_tokens, want_logits=return_step_logits or sampling is not None, slot=slot)
        st["mask_in_decode"] = mask_in_decode
        e2 = mark()
        if prefill_done is not None:
            prefill_done.record()                                  # generate_pipelined: the other pipeline's prefill may start now
        if True:
            B = plan.B
            out = self._cache.get(("out_ids", slot, B, max_new_tokens))   # persistent for the same reason as next_ids (the result is a copy)
            if out is None:
                with torch.inference_mode(False):
                    out = torch.zeros(B, max_new_tokens, dtype=torch.int64, device=self.device)
                self._cache[("out_ids", slot, B, max_new_tokens)] = out
            else:
                out.zero_()
            if sampling is not None:                                  # first token: same rule, RNG counter -1 (decode steps count from 0)
                st["next_ids"].copy_(ops.sample_step(st["logits"], sampling[0], sampling[1], sampling[2], seed=sampling[3], step=-1))
                st["sampling"] = sampling
            out[:, 0] = st["next_ids"]
            step_logits = [st["logits"][None]] if return_step_logits else None
            eos, pad = self.config.eos_token_id, self.config.pad_token_id
            pad = eos if pad is None else pad
            done_at = max_new_tokens
            # transformers' per-token hooks (serve/model_worker.py:160-185, eval/model_vqa*.py): streamer.put(prompt) then put(new ids) per
            # step and end(); stopping_criteria(ids so far, scores) -> True stops every row.  Either one switches to one launch per token.
            ids_dev = input_ids.to(self.device)
            stopped_at = None

            def hooks(n_done):                                       # n_done new tokens are in out[:, :n_done]
                nonlocal stopped_at
                if streamer is not None:
                    streamer.put(out[:, n_done - 1].cpu())
                if criteria and stopped_at is None:
                    seq = torch.cat([ids_dev, out[:, :n_done]], dim=1)
                    if any(bool(c(seq, None)) for c in criteria):
                        stopped_at = n_done
                return stopped_at is not None

            per_token = streamer is not None or bool(criteria)
            if streamer is not None:
                streamer.put(input_ids.cpu())
            halted = hooks(1) if per_token else False
            if forced is not None:
                if sampling is not None or per_token or not ignore_eos:
                    raise ValueError("forced_ids is a greedy, ignore_eos=True, hook-free mode")
                forced = forced.to(self.device, torch.int64)
                if forced.numel() and (int(forced.min()) < 0 or int(forced.max()) >= self.model.embed_tokens.shape[0]):
                    raise IndexError(f"forced_ids holds an id outside the embedding table of {self.model.embed_tokens.shape[0]} rows")
                for s_ in range(max_new_tokens - 1):
                    st["next_ids"].copy_(forced[:, s_])
                    lg = self._decode(st, 1, out[:, 1:], s_, want_logits=return_step_logits)
                    if return_step_logits:
                        step_logits.append(lg)
            elif max_new_tokens > 1 and not halted:
                if ignore_eos and not per_token:
                    lg = self._decode(st, max_new_tokens - 1, out[:, 1:], 0, want_logits=return_step_logits)
                    if return_step_logits:
                        step_logits.append(lg)
                else:
                    chunk, s = (1 if per_token else 16), 0
                    while s < max_new_tokens - 1:
                        n = min(chunk, max_new_tokens - 1 - s)
                        lg = self._decode(st, n, out[:, 1:], s, want_logits=return_step_logits)
                        if return_step_logits:
                            step_logits.append(lg)
                        s += n
                        if per_token and hooks(1 + s):
                            break
                        if not ignore_eos and bool(((out[:, :1 + s] == eos).any(dim=1)).all()):   # host sync once per chunk
                            break
            if streamer is not None:
                streamer.end()
            if stage_events is not None:
                stage_events.update(encode=(e0, e1), prefill=(e1, e2), decode=(e2, mark()), spliced_lens=plan.valid_lens.copy())
            if stopped_at is not None:
                out = out[:, :stopped_at]
                done_at = stopped_at
            new = out
            if not ignore_eos:
                is_eos = new == eos
                after = (is_eos.cumsum(1) - is_eos.long()) > 0                           # strictly after the first EOS
                new = torch.where(after, torch.full_like(new, pad), new)
                fin = is_eos.any(1)
                if bool(fin.all()):
                    done_at = int((is_eos.float().argmax(1) + 1).max().item())
                new = new[:, :done_at]
            res = torch.cat([input_ids.to(self.device), new], dim=1)
            if return_step_logits:
                return res, torch.cat(step_logits, 0).transpose(0, 1)[:, :new.shape[1]]
            return res

    def _beam_search(self, input_ids, modal_inputs, attention_mask, k, max_new_tokens, length_penalty, early_stopping, ignore_eos):
        """generate(num_beams = k > 1): transformers 4.31's beam_search + BeamSearchScorer (the loop eval/model_multimodal_qa_loader.py:94-102
        reaches through --num_beams; restated and pinned in oracle/beam.py).  The prompt is prefilled ONCE per sample and its KV cache rows
        are replicated k times (the reference expands input_ids k-fold and prefills k copies - and, on samples with modal tokens, fails in
        its splice loop because modal_inputs is not expanded with them, multimodal_arch.py:343-346; here those samples work).  Every step is
        one cached decode step of the runtime over the B k beam rows; the scoring (log-softmax) is a kernel, the candidate bookkeeping - 2k
        candidates per sample and step - is the scorer's host logic, the cache rows follow their beams by a device gather per step."""
        import types
        dev = self.device
        eos, pad = (None if ignore_eos else self.config.eos_token_id), self.config.pad_token_id
        pad = self.config.eos_token_id if pad is None else pad
        feats, _ = self.encode_modal_inputs(modal_inputs, self.prefix_tokens, self.suffix_tokens)
        plan = self._plan(input_ids, attention_mask, None, modal_inputs, feats)
        if not plan.mask_is_suffix:
            raise NotImplementedError("beam search with a non-suffix attention mask")
        st0 = self._prefill(plan, feats, max_new_tokens, want_logits=True, slot=("beam", 0))
        B, Smax = plan.B, st0["Smax"]
        Bk = B * k
        (kc, vc), ws = self._buffers(Bk, Smax, Bk, 1, slot=("beam", 1))
        (kc2, vc2), _ = self._buffers(Bk, Smax, Bk, 1, slot=("beam", 2))
        rep = torch.arange(B, device=dev).repeat_interleave(k)
        torch.index_select(st0["kc"], 1, rep, out=kc)
        torch.index_select(st0["vc"], 1, rep, out=vc)
        lens = np.repeat(plan.valid_lens, k).astype(np.int32)
        next_ids = torch.zeros(Bk, dtype=torch.int64, device=dev)
        st = dict(plan=types.SimpleNamespace(B=Bk, valid_lens=lens), kc=kc, vc=vc, ws=ws, Smax=Smax, next_ids=next_ids,
                  kv_lens=torch.from_numpy(lens).to(dev), slot=("beam", 1), key_valid=None)
        scratch = torch.zeros(Bk, 1, dtype=torch.int64, device=dev)
        ids = input_ids.cpu().repeat_interleave(k, dim=0)
        L0 = ids.shape[1]
        max_len = L0 + max_new_tokens
        scores = torch.zeros(B, k, dtype=torch.float32)
        scores[:, 1:] = -1e9
        scores = scores.view(-1)
        from ..beam import BeamHypotheses
        hyps = [BeamHypotheses(k, length_penalty, early_stopping) for _ in range(B)]
        done = [False] * B
        logits = st0["logits"].index_select(0, rep)                   # every beam of a sample starts from the prompt's last-position logits
        step = 0
        V = self.config.vocab_size
        while True:
            cur_len = ids.shape[1]
            if getattr(self, "_beam_trace", None) is not None:       # tests / tools: (beam rows so far, the logits they are scored with) per step
                self._beam_trace.append((ids.clone(), logits.float().cpu()))
            logp = ops.log_softmax(logits)
            cand = (logp + scores.to(dev)[:, None]).view(B, k * V)
            top_s, top_i = torch.topk(cand, 2 * k, dim=1, largest=True, sorted=True)
            top_s, top_i = top_s.cpu(), top_i.cpu()
            top_beam, top_tok = top_i // V, top_i % V
            nxt_scores = torch.zeros(B, k)
            nxt_tok = torch.zeros(B, k, dtype=torch.long)
            nxt_idx = torch.zeros(B, k, dtype=torch.long)
            for b in range(B):
                if done[b]:
                    nxt_scores[b], nxt_tok[b], nxt_idx[b] = 0.0, pad, b * k
                    continue
                n = 0
                for rank in range(2 * k):
                    tok, sc, src = int(top_tok[b, rank]), float(top_s[b, rank]), b * k + int(top_beam[b, rank])
                    if eos is not None and tok == eos:
                        if rank >= k:
                            continue
                        hyps[b].add(ids[src].clone(), sc)
                    else:
                        nxt_scores[b, n], nxt_tok[b, n], nxt_idx[b, n] = sc, tok, src
                        n += 1
                    if n == k:
                        break
                done[b] = done[b] or hyps[b].is_done(float(top_s[b].max()), cur_len + 1)      # 4.31: the length next_scores is calculated on
            scores = nxt_scores.view(-1)
            sel = nxt_idx.view(-1)
            ids = torch.cat([ids[sel], nxt_tok.view(-1, 1)], dim=1)
            if all(done) or ids.shape[1] >= max_len:
                break
            if not torch.equal(sel, torch.arange(Bk)):                # the cache rows follow their beams
                sel_d = sel.to(dev)
                torch.index_select(st["kc"], 1, sel_d, out=kc2)
                torch.index_select(st["vc"], 1, sel_d, out=vc2)
                (st["kc"], kc2), (st["vc"], vc2) = (kc2, st["kc"]), (vc2, st["vc"])
            next_ids.copy_(nxt_tok.view(-1))
            logits = self._decode(st, 1, scratch, step, want_logits=True)[0]
            step += 1
        for b in range(B):
            if not done[b]:
                for j in range(k):
                    hyps[b].add(ids[b * k + j], float(scores[b * k + j]))
        best = [h.best() for h in hyps]
        out_len = min(max(int(x.shape[0]) for x in best) + 1, max_len)
        out = torch.full((B, out_len), pad, dtype=torch.long)
        for b, x in enumerate(best):
            out[b, :x.shape[0]] = x
            if eos is not None and x.shape[0] < out_len:
                out[b, x.shape[0]] = eos
        return out.to(dev)

    @torch.no_grad()
    def generate_pipelined(self, batches, **kw):
        """Throughput mode for a stream of batches (the eval loop): yields generate()'s result for every (input_ids, modal_inputs[,
        attention_mask]) tuple of `batches`, in order.  Two generation pipelines (own KV cache, workspace, decode state and graph each) alternate on two HIP
        streams: the decode of batch i - HBM-bound, the MFMA pipes idle - runs beside the encoders + prefill of batch i+1 - MFMA-bound,
        HBM idle.  The prefill of batch i+1 is ordered after the prefill of batch i (event), so the two never compete for the matrix
        units.  Same tokens as sequential generate() calls; host syncs inside generate() (EOS checks without ignore_eos) shorten the
        overlap but do not break it."""
        cur = torch.cuda.current_stream()
        # (A CU partition - decode steps on a stream masked to a few CUs, the prefill on the others - was measured in rounds 4-5 and lost at
        # every size: 9.98-17.7 against 25.5 samples/s, profiles/r05_probes/cu_partition_ab.json.  It is not in the tree.)
        streams = _shared_streams(self.device, "pipeline", 2)
        try:
            pending = None
            last_prefill = None
            for i, item in enumerate(batches):
                input_ids, modal_inputs = item[0], item[1]
                am = item[2] if len(item) > 2 else None                # optional right-padding mask of the batch
                slot = i & 1
                s = streams[slot]
                s.wait_stream(cur)                                     # inputs produced on the caller's stream
                if last_prefill is not None:
                    s.wait_event(last_prefill)
                ev = torch.cuda.Event()
                with torch.cuda.stream(s):
                    out = self.generate(input_ids, modal_inputs=modal_inputs, attention_mask=am, slot=slot, prefill_done=ev, **kw)
                last_prefill = ev
                if pending is not None:
                    pending[1].synchronize()
                    yield pending[0]
                pending = (out, s)
            if pending is not None:
                pending[1].synchronize()
                yield pending[0]
        finally:
            pass

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None, **kwargs):
        """multimodal_llama.py:747-767 (kept for API parity; generate() does not call it)."""
        if past_key_values:
            input_ids = input_ids[:, -1:]
        model_inputs = {"inputs_embeds": inputs_embeds} if inputs_embeds is not None and past_key_values is None else {"input_ids": input_ids}
        model_inputs.update({"past_key_values": past_key_values, "use_cache": kwargs.get("use_cache"), "attention_mask": attention_mask,
                             "modal_inputs": kwargs.get("modal_inputs", None)})
        return model_inputs


# alias kept by the reference's package __init__ (model/__init__.py:1-3); the legacy single-image class is out of scope
LlavaLlamaForCausalLM = MultimodalLlamaForCausalLM


def _past_len(pkv):
    return pkv[-1][-1].shape[-2]


_CAT_IDX = {}


def _cat_rows(parts: List[torch.Tensor]) -> torch.Tensor:
    """cat along dim 1 of (b, T_i, H) bf16 tensors with the device copy kernel: ONE launch per part (row r = i T_p + j of the part goes to
    row i T + off + j of the result; the index vectors are cached per shape), not one per (part, sample)."""
    b, Hd = parts[0].shape[0], parts[0].shape[-1]
    T = sum(p.shape[1] for p in parts)
    dev = parts[0].device
    out = torch.empty(b, T, Hd, dtype=BF16, device=dev)
    flat = out.view(b * T, Hd)
    off = 0
    for p in parts:
        t = p.shape[1]
        if t == 0:
            continue
        key = (b, T, off, t, str(dev))
        idx = _CAT_IDX.get(key)
        if idx is None:
            idx = (torch.arange(b, dtype=torch.int32)[:, None] * T + off + torch.arange(t, dtype=torch.int32)[None, :]).reshape(-1).to(dev)
            if len(_CAT_IDX) > 256:
                _CAT_IDX.clear()
            _CAT_IDX[key] = idx
        if p.stride(0) == 0 and p[0].is_contiguous():             # a block broadcast over the batch (prefix / suffix tokens): gather it
            skey = ("src", b, t, str(dev))
            sidx = _CAT_IDX.get(skey)
            if sidx is None:
                sidx = torch.arange(t, dtype=torch.int32).repeat(b).to(dev)
                _CAT_IDX[skey] = sidx
            ops.copy_rows(p[0], flat, b * t, sidx, idx)
        else:
            src = (p if p.is_contiguous() else p.contiguous()).view(b * t, Hd)
            ops.copy_rows(src, flat, b * t, None, idx)
        off += t
    return out


def _retention_floats(N: int, K: int) -> int:
    n = C.c_int64(0)
    _lib.check(_lib.lib().mc_compose_retention_floats(N, K, C.byref(n)), "mc_compose_retention_floats")
    return int(n.value)


def _prep_terms(terms):
    """[(A [r, K], B [N, r], scale)] -> (A^T list [K, rp], B list [N, rp], rp): bf16, the rank padded to a multiple of 32."""
    ats, bs, r = [], [], 0
    for (a, b, s) in terms:
        r0 = a.shape[0]
        rp = ops.ceil_to(r0, 32)
        at = a.to(BF16).t().contiguous()
        bb = b.to(BF16).contiguous()
        if rp != r0:
            at = torch.nn.functional.pad(at, (0, rp - r0))
            bb = torch.nn.functional.pad(bb, (0, rp - r0))
        if r and rp != r:
            raise ValueError("the LoRA terms of one linear must share one (padded) rank")
        r = rp
        ats.append(at)
        bs.append(bb)
    return ats, bs, r


def _compose_flush(batch, events=None):
    """The collected argument blocks of many linears -> ONE mc_compose_batch_bf16 call (the structs and every tensor they point to are kept
    alive by the batch entries until the call has returned; the library copies the descriptors and synchronises the stream once)."""
    if not batch:
        return
    arr = (_lib.ComposeMultiArgsC * len(batch))()
    for i, (a, _keep) in enumerate(batch):
        C.memmove(C.addressof(arr[i]), C.addressof(a), C.sizeof(_lib.ComposeMultiArgsC))
    if events is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.lib().mc_compose_batch_bf16(arr, len(batch), _stream()), "mc_compose_batch_bf16")
    if events is not None:
        e1.record()
        events.append((e0, e1))
    batch.clear()


def _compose_multi_into(w: torch.Tensor, terms, masks, N: int, K: int, outs, col_scale=None, nb_stride=1, nb_offset=0, retentions=None, events=None,
                        batch=None):
    """ONE pass over W for every routed adapter of a linear (mc_compose_multi_bf16): output o = (W + sum of the terms named by masks[o])
    diag(col_scale), packed into outs[o].  retentions: per output a list that receives this call's partial sums, or None.
    batch: a list - the call is not issued but appended (argument block + everything it points to) for _compose_flush."""
    n, n_out = len(terms), len(outs)
    ats, bs, r = _prep_terms(terms)
    if w.stride(1) != 1:
        w = w.contiguous()
    if N % 16 and nb_stride != 1:
        raise ValueError("interleaved packing needs N to be a multiple of 16")
    for o in outs:
        if o.numel() != ops.packed_elems(N, K) * nb_stride:
            raise ValueError("packed buffer has the wrong size")
    parts = []
    for oi in range(n_out):
        pt = None
        if retentions is not None and retentions[oi] is not None and masks[oi]:
            pt = torch.zeros(_retention_floats(N, K) // 2, 2, dtype=torch.float32, device=outs[oi].device)     # (entries a kernel's grid does not own stay 0)
            retentions[oi].append(pt)
        parts.append(pt)
    a = _lib.ComposeMultiArgsC()
    a.w, a.ldw = w.data_ptr(), w.stride(0)
    a.at_list = (C.c_void_p * max(n, 1))(*[t.data_ptr() for t in ats])
    a.b_list = (C.c_void_p * max(n, 1))(*[t.data_ptr() for t in bs])
    a.scales = (C.c_float * max(n, 1))(*[float(t[2]) for t in terms])
    a.n_terms, a.r, a.n_out = n, r, n_out
    a.out_packed = (C.c_void_p * n_out)(*[o.data_ptr() for o in outs])
    a.out_rowmajor = None
    a.term_mask = (C.c_uint32 * n_out)(*[int(m) for m in masks])
    a.dither_seeds = None
    a.retention_parts = (C.c_void_p * n_out)(*[0 if pt is None else pt.data_ptr() for pt in parts])
    a.ldo, a.N, a.K = 0, N, K
    a.col_scale = 0 if col_scale is None else col_scale.data_ptr()
    a.nb_stride, a.nb_offset = nb_stride, nb_offset
    if batch is not None:
        batch.append((a, (w, ats, bs, parts, col_scale, outs)))
        return outs
    if events is not None:                                        # (start, end) pair per launch: the kernel's own device time (bench.py's compose roofline)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.lib().mc_compose_multi_bf16(C.byref(a), _stream()), "mc_compose_multi_bf16")
    if events is not None:
        e1.record()
        events.append((e0, e1))
    return outs


def _compose_into(w: torch.Tensor, terms, N: int, K: int, out: torch.Tensor, col_scale=None, nb_stride=1, nb_offset=0, retention=None,
                  dither_seed: int = 0):
    """W' = (W + Σ scale·B·A)·diag(col_scale) packed into the preallocated buffer `out` (16-row block nb at nb*nb_stride + nb_offset).
    retention: a list that receives this call's [workgroups, 2] fp32 partial sums (Σ (W' - bf16(W c))·(ΔW c), Σ (ΔW c)²), see
    delta_retention()."""
    n = len(terms)
    ats, bs, r = [], [], 0
    for (a, b, s) in terms:
        r0 = a.shape[0]
        rp = ops.ceil_to(r0, 32)
        at = a.to(BF16).t().contiguous()
        bb = b.to(BF16).contiguous()
        if rp != r0:
            at = torch.nn.functional.pad(at, (0, rp - r0))
            bb = torch.nn.functional.pad(bb, (0, rp - r0))
        r = rp
        ats.append(at)
        bs.append(bb)
    if w.stride(1) != 1:
        w = w.contiguous()
    if N % 16 and nb_stride != 1:
        raise ValueError("interleaved packing needs N to be a multiple of 16")
    at_arr = (C.c_void_p * max(n, 1))(*[t.data_ptr() for t in ats])
    b_arr = (C.c_void_p * max(n, 1))(*[t.data_ptr() for t in bs])
    sc = (C.c_float * max(n, 1))(*[float(t[2]) for t in terms])
    if out.numel() != ops.packed_elems(N, K) * nb_stride:
        raise ValueError("packed buffer has the wrong size")
    parts = None
    if retention is not None and n > 0:
        parts = torch.zeros(_retention_floats(N, K) // 2, 2, dtype=torch.float32, device=out.device)
        retention.append(parts)
    _lib.check(_lib.lib().mc_compose_weight_dither_bf16(_ptr(w), w.stride(0), at_arr, b_arr, sc, n, r, _ptr(out), None, 0, N, K,
                                                        _ptr(col_scale), nb_stride, nb_offset, _ptr(parts), int(dither_seed) & 0xFFFFFFFF,
                                                        _stream()), "mc_compose_weight_dither_bf16")
    # keep operands alive until the kernel has been enqueued on the stream (stream-ordered allocator semantics)
    return out
