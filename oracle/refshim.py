"""TEST INFRASTRUCTURE — build-container only.

Import shim that makes the *unmodified* reference sources under /root/reference
importable with the library versions of this image (transformers 5.x, no peft /
timm / torchaudio ...).  It is used ONLY by ``oracle/gen_golden.py`` to produce
the small fp32 fixtures in ``tests/golden/``; nothing here (and nothing from
/root/reference) travels to the GPU box or is imported by the product package.

The third-party pieces the reference pins but this image lacks are restated here
from their published behaviour (peft==0.4.0 ``tuners.lora.Linear``,
transformers==4.31 llama helpers).  They are therefore NOT reference-pinned; the
in-reference files are the authority.
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import math
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = "/root/reference"
_installed = False


# ----------------------------------------------------------------------------
# peft 0.4.0 restatement (only what multimodal_llama.py / builder.py touch)
# ----------------------------------------------------------------------------
class _LoraLayer:
    def __init__(self, in_features: int, out_features: int, **kwargs):
        self.r = {}
        self.lora_alpha = {}
        self.scaling = {}
        self.lora_dropout = nn.ModuleDict({})
        self.lora_A = nn.ModuleDict({})
        self.lora_B = nn.ModuleDict({})
        self.lora_embedding_A = nn.ParameterDict({})
        self.lora_embedding_B = nn.ParameterDict({})
        self.merged = False
        self.disable_adapters = False
        self.in_features = in_features
        self.out_features = out_features
        self.kwargs = kwargs

    def update_layer(self, adapter_name, r, lora_alpha, lora_dropout, init_lora_weights):
        self.r[adapter_name] = r
        self.lora_alpha[adapter_name] = lora_alpha
        if lora_dropout > 0.0:
            lora_dropout_layer = nn.Dropout(p=lora_dropout)
        else:
            lora_dropout_layer = nn.Identity()
        self.lora_dropout.update(nn.ModuleDict({adapter_name: lora_dropout_layer}))
        if r > 0:
            self.lora_A.update(nn.ModuleDict({adapter_name: nn.Linear(self.in_features, r, bias=False)}))
            self.lora_B.update(nn.ModuleDict({adapter_name: nn.Linear(r, self.out_features, bias=False)}))
            self.scaling[adapter_name] = lora_alpha / r
        if init_lora_weights:
            self.reset_lora_parameters(adapter_name)
        self.to(self.weight.device)

    def reset_lora_parameters(self, adapter_name):
        if adapter_name in self.lora_A.keys():
            nn.init.kaiming_uniform_(self.lora_A[adapter_name].weight, a=math.sqrt(5))
            nn.init.zeros_(self.lora_B[adapter_name].weight)


class _LoraLinear(nn.Linear, _LoraLayer):
    def __init__(self, adapter_name, in_features, out_features, r=0, lora_alpha=1, lora_dropout=0.0,
                 fan_in_fan_out=False, is_target_conv_1d_layer=False, **kwargs):
        init_lora_weights = kwargs.pop("init_lora_weights", True)
        nn.Linear.__init__(self, in_features, out_features, **kwargs)
        _LoraLayer.__init__(self, in_features=in_features, out_features=out_features)
        self.weight.requires_grad = False
        self.fan_in_fan_out = fan_in_fan_out
        if fan_in_fan_out:
            self.weight.data = self.weight.data.T
        nn.Linear.reset_parameters(self)
        self.update_layer(adapter_name, r, lora_alpha, lora_dropout, init_lora_weights)
        self.active_adapter = adapter_name
        self.is_target_conv_1d_layer = is_target_conv_1d_layer


def _transpose(weight, fan_in_fan_out):
    return weight.T if fan_in_fan_out else weight


# ----------------------------------------------------------------------------
# transformers 4.31 llama helpers (restated)
# ----------------------------------------------------------------------------
class _LlamaRotaryEmbedding431(nn.Module):
    def __init__(self, dim=None, max_position_embeddings=2048, base=10000, device=None, config=None, **kw):
        super().__init__()
        if dim is None and config is not None:
            dim = config.hidden_size // config.num_attention_heads
            max_position_embeddings = config.max_position_embeddings
        self.dim = dim
        self.max_position_embeddings = max_position_embeddings
        self.base = base
        inv_freq = 1.0 / (self.base ** (torch.arange(0, self.dim, 2).float().to(device) / self.dim))
        self.register_buffer("inv_freq", inv_freq, persistent=False)
        self._set_cos_sin_cache(max_position_embeddings, device=self.inv_freq.device, dtype=torch.get_default_dtype())

    def _set_cos_sin_cache(self, seq_len, device, dtype):
        self.max_seq_len_cached = seq_len
        t = torch.arange(self.max_seq_len_cached, device=device, dtype=self.inv_freq.dtype)
        freqs = torch.einsum("i,j->ij", t, self.inv_freq)
        emb = torch.cat((freqs, freqs), dim=-1)
        self.register_buffer("cos_cached", emb.cos()[None, None, :, :].to(dtype), persistent=False)
        self.register_buffer("sin_cached", emb.sin()[None, None, :, :].to(dtype), persistent=False)

    def forward(self, x, seq_len=None):
        if seq_len > self.max_seq_len_cached:
            self._set_cos_sin_cache(seq_len=seq_len, device=x.device, dtype=x.dtype)
        return (
            self.cos_cached[:, :, :seq_len, ...].to(dtype=x.dtype),
            self.sin_cached[:, :, :seq_len, ...].to(dtype=x.dtype),
        )


def _rotate_half(x):
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def _apply_rotary_pos_emb431(q, k, cos, sin, position_ids):
    cos = cos.squeeze(1).squeeze(0)
    sin = sin.squeeze(1).squeeze(0)
    cos = cos[position_ids].unsqueeze(1)
    sin = sin[position_ids].unsqueeze(1)
    q_embed = (q * cos) + (_rotate_half(q) * sin)
    k_embed = (k * cos) + (_rotate_half(k) * sin)
    return q_embed, k_embed


def _make_causal_mask(input_ids_shape, dtype, device, past_key_values_length=0):
    bsz, tgt_len = input_ids_shape
    mask = torch.full((tgt_len, tgt_len), torch.finfo(dtype).min, device=device)
    mask_cond = torch.arange(mask.size(-1), device=device)
    mask.masked_fill_(mask_cond < (mask_cond + 1).view(mask.size(-1), 1), 0)
    mask = mask.to(dtype)
    if past_key_values_length > 0:
        mask = torch.cat([torch.zeros(tgt_len, past_key_values_length, dtype=dtype, device=device), mask], dim=-1)
    return mask[None, None, :, :].expand(bsz, 1, tgt_len, tgt_len + past_key_values_length)


def _expand_mask(mask, dtype, tgt_len=None):
    bsz, src_len = mask.size()
    tgt_len = tgt_len if tgt_len is not None else src_len
    expanded_mask = mask[:, None, None, :].expand(bsz, 1, tgt_len, src_len).to(dtype)
    inverted_mask = 1.0 - expanded_mask
    return inverted_mask.masked_fill(inverted_mask.to(torch.bool), torch.finfo(dtype).min)


def _prepare_decoder_attention_mask(self, attention_mask, input_shape, inputs_embeds, past_key_values_length):
    combined = None
    if input_shape[-1] > 1:
        combined = _make_causal_mask(input_shape, inputs_embeds.dtype, device=inputs_embeds.device,
                                     past_key_values_length=past_key_values_length)
    if attention_mask is not None:
        expanded = _expand_mask(attention_mask, inputs_embeds.dtype, tgt_len=input_shape[-1]).to(inputs_embeds.device)
        combined = expanded if combined is None else expanded + combined
    return combined


# ----------------------------------------------------------------------------
# stub finder for missing leaf dependencies
# ----------------------------------------------------------------------------
_STUB_ROOTS = {
    "timm", "torchaudio", "ftfy", "cv2", "decord", "torchvision", "pytorchvideo", "easydict", "moviepy",
    "omegaconf", "librosa", "iopath", "fvcore", "deepspeed", "termcolor", "open3d", "bitsandbytes",
    "flash_attn", "shortuuid", "mayavi", "vtk", "cartopy", "xformers", "apex", "gradio", "webdataset",
}


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name not in ("__path__",):
            raise AttributeError(name)
        full = f"{self.__name__}.{name}"
        if full in sys.modules:
            return sys.modules[full]
        # attribute access yields a callable/class-like stub
        return _StubObj(full)


class _StubObj:
    def __init__(self, name="stub"):
        self._name = name

    def __call__(self, *a, **k):
        # used as decorator -> return the function unchanged
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return _StubObj(self._name + "()")

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _StubObj(self._name + "." + name)

    def __mro_entries__(self, bases):
        return (object,)


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


class _EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _EasyDict):
            v = _EasyDict(v)
        elif isinstance(v, (list, tuple)):
            v = type(v)(_EasyDict(x) if isinstance(x, dict) else x for x in v)
        super().__setattr__(k, v)
        super().__setitem__(k, v)

    __setitem__ = __setattr__


class _DropPath(nn.Module):
    def __init__(self, drop_prob=0.0, *a, **k):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        return x


def install():
    """Install every shim; idempotent."""
    global _installed
    if _installed:
        return
    _installed = True
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    sys.path.insert(0, REF_ROOT + "/scripts/model_composition")

    # 1. namespace packages that skip the eager __init__ files
    for name, rel in [("modelcompose", "modelcompose"), ("modelcompose.model", "modelcompose/model"),
                      ("modelcompose.model.language_model", "modelcompose/model/language_model")]:
        m = types.ModuleType(name)
        m.__path__ = [f"{REF_ROOT}/{rel}"]
        sys.modules[name] = m

    # 2. fake peft
    peft = types.ModuleType("peft"); peft.__path__ = []
    tuners = types.ModuleType("peft.tuners"); tuners.__path__ = []
    lora = types.ModuleType("peft.tuners.lora")
    utils = types.ModuleType("peft.utils")
    lora.Linear = _LoraLinear
    lora.LoraLayer = _LoraLayer
    utils.transpose = _transpose
    peft.tuners = tuners; tuners.lora = lora; peft.utils = utils
    peft.LoraConfig = _StubObj("peft.LoraConfig"); peft.get_peft_model = _StubObj("peft.get_peft_model")
    peft.PeftModel = _StubObj("peft.PeftModel")
    for m_ in (peft, tuners, lora, utils):
        m_.__spec__ = importlib.machinery.ModuleSpec(m_.__name__, None)
    sys.modules.update({"peft": peft, "peft.tuners": tuners, "peft.tuners.lora": lora, "peft.utils": utils})
    import transformers.utils.import_utils as _iu
    import transformers.integrations.peft as _ip
    _iu.is_peft_available = lambda *a, **k: False
    _ip.is_peft_available = lambda *a, **k: False

    # 3. transformers.modeling_utils names removed since 4.31
    import transformers
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu

    def get_parameter_device(p):
        return next(p.parameters()).device

    def get_parameter_dtype(p):
        return next(p.parameters()).dtype

    for nm, fn in [("get_parameter_device", get_parameter_device), ("get_parameter_dtype", get_parameter_dtype)]:
        if not hasattr(mu, nm):
            setattr(mu, nm, fn)
    for nm in ("apply_chunking_to_forward", "prune_linear_layer", "find_pruneable_heads_and_indices"):
        if not hasattr(mu, nm):
            setattr(mu, nm, getattr(pu, nm, _StubObj(nm)))

    # 4. clip helpers
    import transformers.models.clip.modeling_clip as mc
    if not hasattr(mc, "_expand_mask"):
        mc._expand_mask = _expand_mask
    if not hasattr(mc, "clip_loss"):
        mc.clip_loss = _StubObj("clip_loss")

    # 5. llama 4.31 names
    import transformers.models.llama.modeling_llama as ml
    from transformers.activations import ACT2FN
    from transformers.modeling_outputs import BaseModelOutputWithPast
    inject = dict(BaseModelOutputWithPast=BaseModelOutputWithPast, ACT2FN=ACT2FN, math=math, torch=torch, nn=nn, F=F,
                  logger=ml.logger, LlamaRMSNorm=ml.LlamaRMSNorm, repeat_kv=ml.repeat_kv,
                  LlamaRotaryEmbedding=_LlamaRotaryEmbedding431,
                  LlamaLinearScalingRotaryEmbedding=_LlamaRotaryEmbedding431,
                  LlamaDynamicNTKScalingRotaryEmbedding=_LlamaRotaryEmbedding431,
                  apply_rotary_pos_emb=_apply_rotary_pos_emb431)
    for k, v in inject.items():
        setattr(ml, k, v)
    allnames = list(getattr(ml, "__all__", []))
    for k in inject:
        if k not in allnames:
            allnames.append(k)
    ml.__all__ = allnames
    ml.LlamaModel._prepare_decoder_attention_mask = _prepare_decoder_attention_mask

    # 6. stub finder
    sys.meta_path.append(_StubFinder())
    import timm.models.layers as tml  # noqa: stubbed
    tml.DropPath = _DropPath
    tml.trunc_normal_ = nn.init.trunc_normal_
    tml.to_2tuple = lambda x: (x, x) if not isinstance(x, tuple) else x
    import timm.models.hub as tmh  # noqa
    import easydict
    easydict.EasyDict = _EasyDict


def import_ref(name: str):
    install()
    return importlib.import_module(name)
