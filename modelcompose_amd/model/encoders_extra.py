"""Audio / video / point encoders and the Q-Former projector on the HIP path.

Host-side mirrors of (paths under /root/reference/modelcompose/model):
  BeatsAudioEncoder        multimodal_encoder/audio_encoder.py:8-77  (BEATs.extract_features_new, beats/BEATs.py:149-189)
  VideoLlamaAudioQformer   multimodal_projector/builder.py:111-173   (BLIP-2 Q-Former, multimodal_projector/Qformer.py)
  LanguageBindVideoTower   multimodal_encoder/languagebind/__init__.py:180-260 (video/modeling_video.py:65-161,599-678)
  PointEncoder             multimodal_encoder/point_encoder.py:12-117 (pointbert/point_encoder.py:169-189, pointbert/dvae.py)
Every tensor op is a kernel of libmc_hip.so; torch is used for parameter folding at load time (weight-norm, BatchNorm
running statistics, bucket tables) and for memory."""
from __future__ import annotations

import math
import os
from typing import Dict, Optional

import numpy as np
import torch

from ..checkpoint_io import load_nested, load_tensors

from .. import _lib, ops
from .clip import _strip

BF16 = _lib.storage_dtype()      # the library's 16-bit storage element: bf16, or fp16 with MC_STORAGE_DTYPE=fp16 (_lib.set_storage_dtype)


def _pw(sd, w, b=None, dev="cuda"):
    return ops.pack_weight(sd[w].to(dev), None if b is None or b not in sd else sd[b].to(dev))


def _padk(x: torch.Tensor, Kp: int) -> torch.Tensor:
    return x if x.shape[1] == Kp else torch.nn.functional.pad(x, (0, Kp - x.shape[1]))


# =========================================================================================================
# BEATs
# =========================================================================================================
class BeatsConfig:
    """beats/BEATs.py:25-65 defaults, overridden by the checkpoint's cfg dict."""

    def __init__(self, cfg: Optional[dict] = None):
        self.input_patch_size, self.embed_dim, self.conv_bias = -1, 512, False
        self.encoder_layers, self.encoder_embed_dim, self.encoder_ffn_embed_dim, self.encoder_attention_heads = 12, 768, 3072, 12
        self.activation_fn, self.layer_norm_first, self.deep_norm = "gelu", False, False
        self.conv_pos, self.conv_pos_groups = 128, 16
        self.relative_position_embedding, self.num_buckets, self.max_distance, self.gru_rel_pos = False, 320, 1280, False
        if cfg:
            self.__dict__.update(cfg)


def _rel_table(emb: torch.Tensor, T: int, num_buckets: int, max_distance: int) -> torch.Tensor:
    """Emb[bucket(j - i)] as a table over the offset j - i in [-(T-1), T-1]  (beats/backbone.py:431-468): [H, 2T-1] fp32."""
    rel = torch.arange(-(T - 1), T, dtype=torch.long)
    nb = num_buckets // 2
    buckets = (rel > 0).to(torch.long) * nb
    r = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(r.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, nb - 1))
    buckets = buckets + torch.where(r < max_exact, r, large)
    return emb.float().cpu()[buckets].t().contiguous()          # [H, 2T-1]


class HipBeatsAudioEncoder:
    def __init__(self, audio_encoder: Optional[str], args=None, delay_load=False, config: Optional[BeatsConfig] = None, device="cuda"):
        self.audio_encoder_name, self.device, self.dtype = audio_encoder, torch.device(device), BF16
        self.is_loaded, self.cfg, self.audio_processor = False, config, None
        self._bad_mask, self._pending_mask_check = None, False               # device flag of the padding analysis (forward / check_pending)
        try:
            from .audio_processor import HipBeatsAudioProcessor
            self.audio_processor = HipBeatsAudioProcessor(device=device)          # audio_encoder.py:33 (BeatsAudioProcessor())
        except Exception:
            self.audio_processor = None
        if audio_encoder is not None and os.path.isfile(str(audio_encoder)):
            ck = self._read_checkpoint(audio_encoder)
            self.cfg = BeatsConfig(ck["cfg"])
            if not delay_load:
                self.load_state_dict(ck["model"])

    config = property(lambda self: self.cfg)
    hidden_size = property(lambda self: self.cfg.encoder_embed_dim)      # audio_encoder.py hidden_size
    modal_processor = property(lambda self: self.audio_processor)

    @property
    def dummy_inputs(self):
        return {"audio_inputs": torch.zeros(1, 1024, 128, device=self.device, dtype=self.dtype),
                "audio_padding_mask": torch.zeros(1, 1024, device=self.device, dtype=torch.bool)}

    @staticmethod
    def _read_checkpoint(path):
        """BEATs files are {'cfg': {...scalars...}, 'model': state_dict} (beats/BEATs.py:120-148 reads checkpoint['cfg'] and
        checkpoint['model']): the tree form of the native reader keeps the config's scalar leaves."""
        ck = load_nested(path)
        if not isinstance(ck, dict) or "cfg" not in ck or "model" not in ck:
            raise KeyError(f"{path}: a BEATs checkpoint holds 'cfg' and 'model'; found {sorted(ck) if isinstance(ck, dict) else type(ck).__name__}")
        return ck

    def load_model(self):
        if not self.is_loaded:
            ck = self._read_checkpoint(self.audio_encoder_name)
            self.cfg = BeatsConfig(ck["cfg"])
            self.load_state_dict(ck["model"])

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        if any(k.startswith("audio_encoder.") for k in sd):
            sd = _strip(sd, "audio_encoder.")
        c, dev = self.cfg, self.device
        if c.activation_fn != "gelu":
            raise NotImplementedError(f"BEATs activation '{c.activation_fn}' (only gelu is used by released checkpoints)")
        t = lambda k: sd[k].to(dev, BF16).contiguous()
        E = c.embed_dim
        self.patch_w = ops.pack_weight(sd["patch_embedding.weight"].to(dev).reshape(E, -1),
                                       sd["patch_embedding.bias"].to(dev) if "patch_embedding.bias" in sd else None)
        self.ln0 = (t("layer_norm.weight"), t("layer_norm.bias"))
        self.proj = _pw(sd, "post_extract_proj.weight", "post_extract_proj.bias", dev) if "post_extract_proj.weight" in sd else None
        # weight-norm (dim=2) folded once: w = v * g / ||v||_{(0,1)}   (beats/backbone.py:28-47)
        p = "encoder.pos_conv.0"
        if f"{p}.parametrizations.weight.original0" in sd:
            g, v = sd[f"{p}.parametrizations.weight.original0"].float(), sd[f"{p}.parametrizations.weight.original1"].float()
            w = v * (g / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())
        elif f"{p}.weight_g" in sd:
            g, v = sd[f"{p}.weight_g"].float(), sd[f"{p}.weight_v"].float()
            w = v * (g / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())
        else:
            w = sd[f"{p}.weight"].float()
        C = c.encoder_embed_dim
        G = c.conv_pos_groups
        Cg = C // G
        pb = sd[f"{p}.bias"]
        # per group: [Cg out, Cg in, k] -> [Cg out, k, Cg in]: the reduction index runs (tap, channel), so the operand row of output
        # token t is the CONTIGUOUS window of k x Cg values starting at padded token t of the group's channel slice (see forward)
        self.pos_conv = [ops.pack_weight(w[g_ * Cg:(g_ + 1) * Cg].permute(0, 2, 1).reshape(Cg, -1).contiguous().to(dev), pb[g_ * Cg:(g_ + 1) * Cg].to(dev))
                         for g_ in range(G)]
        self.enc_ln = (t("encoder.layer_norm.weight"), t("encoder.layer_norm.bias"))
        self.layers = []
        for i in range(c.encoder_layers):
            q = f"encoder.layers.{i}."
            wqkv = torch.cat([sd[q + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0).to(dev)
            bqkv = torch.cat([sd[q + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0).to(dev)
            L = dict(qkv=ops.pack_weight(wqkv, bqkv), out=_pw(sd, q + "self_attn.out_proj.weight", q + "self_attn.out_proj.bias", dev),
                     ln1=(t(q + "self_attn_layer_norm.weight"), t(q + "self_attn_layer_norm.bias")),
                     fc1=_pw(sd, q + "fc1.weight", q + "fc1.bias", dev), fc2=_pw(sd, q + "fc2.weight", q + "fc2.bias", dev),
                     ln2=(t(q + "final_layer_norm.weight"), t(q + "final_layer_norm.bias")))
            if c.gru_rel_pos:
                L["grep"] = _pw(sd, q + "self_attn.grep_linear.weight", q + "self_attn.grep_linear.bias", dev)
                L["grep_a"] = sd[q + "self_attn.grep_a"].to(dev, torch.float32).reshape(-1).contiguous()
            self.layers.append(L)
        self._rel_emb = sd["encoder.layers.0.self_attn.relative_attention_bias.weight"] if c.relative_position_embedding else None
        self._rel_cache = {}
        self.is_loaded = True

    def __call__(self, audio_inputs, audio_padding_mask=None):
        return self.forward(audio_inputs, audio_padding_mask)

    def forward(self, audio_inputs, audio_padding_mask=None):
        """(features (B, T, C), valid-token mask (B, T)) like BeatsAudioEncoder.forward (audio_encoder.py:37-40)."""
        c, dev = self.cfg, self.device
        # a deferred refusal the caller of the PREVIOUS forward never collected (standalone use of the encoder: nothing calls check_pending
        # for it) is raised now rather than never
        self.check_pending()
        x = audio_inputs.to(dev, BF16)
        B, Fr, Mel = x.shape
        ps = c.input_patch_size
        cols, oh, ow = ops.im2col(x.view(B, 1, Fr, Mel), ps, ps, ps, ps)
        T = oh * ow
        h = ops.linear(cols, self.patch_w)                                     # conv patch embedding, tokens time-major
        h = ops.layernorm(h, self.ln0[0], self.ln0[1], 1e-5)
        kv_lens, pooled, pm_dev = None, None, None
        if audio_padding_mask is not None:                                     # forward_padding_mask (BEATs.py:120-132)
            if audio_padding_mask.device.type == "cpu":
                # a host mask: analysed on the host for free, the unsupported case refused at once
                pm = audio_padding_mask.bool()
                extra = pm.shape[1] % T
                pm = pm[:, :-extra] if extra > 0 else pm
                pooled = pm.view(B, T, -1).all(-1)
                valid = (~pooled).long().sum(1)
                if not bool((pooled == (torch.arange(T)[None] >= valid[:, None])).all()):
                    raise NotImplementedError("only trailing audio padding is supported (what BeatsAudioProcessor produces)")
                kv_lens = valid.to(torch.int32).to(dev)
            else:
                # a device mask (the eval loop moves the collator's batch to the GPU): NO host round trip (round 4 - the .to("cpu") here cost a
                # device sync per audio encode, after which the host issued the encoder's ~200 launches with the GPU waiting on it).  One launch
                # pools the frame mask, writes the clip lengths, zeroes the padded rows and flags non-suffix padding; the flag is read at
                # the caller's next natural sync (check_pending: MultimodalLlamaForCausalLM._plan copies the ids to the host anyway).
                pm_dev = audio_padding_mask.to(torch.uint8)
                if not pm_dev.is_contiguous():
                    pm_dev = pm_dev.contiguous()
                kv_lens = torch.empty(B, dtype=torch.int32, device=dev)
                if self._bad_mask is None:
                    self._bad_mask = torch.zeros(1, dtype=torch.int32, device=dev)
        if self.proj is not None:
            h = ops.linear(h, self.proj)
        C = c.encoder_embed_dim
        if pooled is not None and bool(pooled.any()):                          # x[padding_mask] = 0 (backbone.py:150-151)
            rows = torch.nonzero(pooled.reshape(-1)).to(torch.int32).reshape(-1).to(dev)
            ops.zero_rows(h, rows)
        if pm_dev is not None:
            span = pm_dev.shape[1] // T
            if span < 1:
                raise ValueError(f"audio_padding_mask has {pm_dev.shape[1]} frames for {T} tokens")
            ops.beats_padding(pm_dev, B, T, span, h, kv_lens, self._bad_mask)
            self._pending_mask_check = True
        # convolutional position embedding: grouped conv1d(k, pad k//2) -> drop last -> GELU, added to x (backbone.py:71-85,153-155)
        # No im2col: the tokens of every clip are laid out with k/2 zero rows before and k/2 after (Tp = T + k rows per clip); with the
        # weights' reduction index ordered (tap, channel) the operand row of output token (b, t) is the contiguous run of k x Cg values
        # that starts at row b Tp + t of the group's channel slice, i.e. the GEMM's activation matrix is that slice viewed with a row
        # stride of Cg elements (overlapping rows: 2.9 MB per group, L2 resident, instead of a 292 MB im2col matrix written and read per
        # group at B = 48).  Rows t >= T of a clip (windows that run into the next clip) are computed and dropped.
        G, Cg, kp = c.conv_pos_groups, C // c.conv_pos_groups, c.conv_pos
        if Cg % 8 or (kp * Cg) % 64:
            raise NotImplementedError("conv_pos groups must give a channel slice that is a multiple of 8 and a window that is a multiple of 64")
        Tp = T + kp
        key = ("posconv", B, T)
        if key not in self._rel_cache:
            tok = torch.arange(B * T, dtype=torch.int64)
            rows_in = ((tok // T) * Tp + kp // 2 + tok % T).to(torch.int32).to(dev)          # padded row of token (b, t)
            rows_out = ((tok // T) * Tp + tok % T).to(torch.int32).to(dev)                   # GEMM row that holds its output
            self._rel_cache[key] = (rows_in, rows_out)
        rows_in, rows_out = self._rel_cache[key]
        hp = torch.zeros(B * Tp + kp, C, dtype=BF16, device=dev)
        ops.copy_rows(h, hp, B * T, dst_idx=rows_in)
        xop = torch.empty(B * Tp, C, dtype=BF16, device=dev)
        res = hp[kp // 2:]                                                     # output row r is token r + k/2 of the padded layout
        xg = torch.empty(B * Tp + kp, Cg, dtype=BF16, device=dev)
        for g in range(G):
            ops.copy_rows(hp[:, g * Cg:(g + 1) * Cg], xg, B * Tp + kp)
            win = torch.as_strided(xg, (B * Tp, kp * Cg), (Cg, 1))
            ops.linear(win, self.pos_conv[g], act="gelu", residual=res[:, g * Cg:(g + 1) * Cg], out=xop[:, g * Cg:(g + 1) * Cg])
        xo = torch.empty_like(h)
        ops.copy_rows(xop, xo, B * T, src_idx=rows_out)
        h = xo
        if not c.layer_norm_first:
            h = ops.layernorm(h, self.enc_ln[0], self.enc_ln[1], 1e-5)
        else:
            raise NotImplementedError("layer_norm_first BEATs variants are not released with ModelCompose")
        H = c.encoder_attention_heads
        d = C // H
        alpha = math.pow(2 * c.encoder_layers, 0.25) if c.deep_norm else 1.0
        rel = None
        if self._rel_emb is not None:
            if T not in self._rel_cache:
                self._rel_cache[T] = _rel_table(self._rel_emb, T, c.num_buckets, c.max_distance).to(dev)
            rel = self._rel_cache[T]
        st = (T * 3 * C, 3 * C, d)
        for L in self.layers:
            qkv = ops.linear(h, L["qkv"])
            gate = None
            if rel is not None and c.gru_rel_pos:
                qh = qkv[:, :C].contiguous().view(B * T * H, d)                # per-head q rows (b, t, h)
                g8 = ops.linear(_padk(qh, L["grep"].Kp), L["grep"], out_f32=True)
                gate = ops.beats_gate(g8, L["grep_a"], B, T, H)
            a = torch.empty(B * T, C, dtype=BF16, device=dev)
            ops.attn_prefill(qkv, qkv[:, C:], qkv[:, 2 * C:], a, B, H, H, T, T, d, st, st, st, C, False, 0, scale=d ** -0.5,
                             kv_lens=kv_lens, rel_table=rel, rel_off=T - 1, q_gate=gate)
            h = ops.layernorm(ops.linear(a, L["out"], residual=h, beta=alpha), L["ln1"][0], L["ln1"][1], 1e-5)
            f = ops.linear(h, L["fc1"], act="gelu")
            h = ops.layernorm(ops.linear(f, L["fc2"], residual=h, beta=alpha), L["ln2"][0], L["ln2"][1], 1e-5)
        feats = h.view(B, T, C)
        if pooled is not None:
            mask = (~pooled).to(dev)
        elif kv_lens is not None:
            mask = torch.arange(T, device=dev)[None] < kv_lens[:, None]        # valid-token mask (the caller ignores it, multimodal_arch.py:233-235)
        else:
            mask = None
        return feats, mask

    def check_pending(self):
        """Deferred refusal of a device padding mask that was not a suffix (read at a point where the caller synchronises anyway)."""
        if getattr(self, "_pending_mask_check", False):
            self._pending_mask_check = False
            if int(self._bad_mask.item()) != 0:
                self._bad_mask.zero_()
                raise NotImplementedError("only trailing audio padding is supported (what BeatsAudioProcessor produces)")


# =========================================================================================================
# Q-Former projector
# =========================================================================================================
class HipQformerProjector:
    """VideoLlamaAudioQformer (multimodal_projector/builder.py:111-173): learned queries, self- + cross-attention per layer,
    query FFN, Linear(768 -> hidden)."""

    def __init__(self, num_query_token=8, vision_width=1024, num_hidden_layers=2, num_positions=1024, hidden_size=768,
                 num_attention_heads=12, intermediate_size=3072, layer_norm_eps=1e-12, device="cuda"):
        self.nq, self.width, self.nl, self.npos = num_query_token, vision_width, num_hidden_layers, num_positions
        self.hidden, self.heads, self.inter, self.eps = hidden_size, num_attention_heads, intermediate_size, layer_norm_eps
        self.device = torch.device(device)

    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str = ""):
        sd = _strip(sd, prefix) if prefix else sd
        dev = self.device
        t = lambda k: sd[k].to(dev, BF16).contiguous()
        self.pos = t("audio_position_embedding.weight")
        self.query = sd["audio_query_tokens"].to(dev, BF16).reshape(-1, self.hidden).contiguous()
        e = "audio_Qformer.bert.embeddings.LayerNorm"
        self.emb_ln = (t(e + ".weight"), t(e + ".bias"))
        self.layers = []
        for i in range(self.nl):
            p = f"audio_Qformer.bert.encoder.layer.{i}."
            L = {}
            for nm, pre in (("sa", p + "attention."), ("ca", p + "crossattention.")):
                if nm == "sa":
                    w = torch.cat([sd[pre + f"self.{n}.weight"] for n in ("query", "key", "value")], 0).to(dev)
                    b = torch.cat([sd[pre + f"self.{n}.bias"] for n in ("query", "key", "value")], 0).to(dev)
                    L["sa_qkv"] = ops.pack_weight(w, b)
                else:
                    L["ca_q"] = _pw(sd, pre + "self.query.weight", pre + "self.query.bias", dev)
                    w = torch.cat([sd[pre + f"self.{n}.weight"] for n in ("key", "value")], 0).to(dev)
                    b = torch.cat([sd[pre + f"self.{n}.bias"] for n in ("key", "value")], 0).to(dev)
                    L["ca_kv"] = ops.pack_weight(w, b)
                L[nm + "_o"] = _pw(sd, pre + "output.dense.weight", pre + "output.dense.bias", dev)
                L[nm + "_ln"] = (t(pre + "output.LayerNorm.weight"), t(pre + "output.LayerNorm.bias"))
            L["fc1"] = _pw(sd, p + "intermediate_query.dense.weight", p + "intermediate_query.dense.bias", dev)
            L["fc2"] = _pw(sd, p + "output_query.dense.weight", p + "output_query.dense.bias", dev)
            L["ff_ln"] = (t(p + "output_query.LayerNorm.weight"), t(p + "output_query.LayerNorm.bias"))
            self.layers.append(L)
        self.out = _pw(sd, "audio_llama_proj.weight", "audio_llama_proj.bias", dev)

    def __call__(self, x: torch.Tensor, *a, **k) -> torch.Tensor:
        dev, Dm, H = self.device, self.hidden, self.heads
        d = Dm // H
        B, T, W = x.shape
        xe = x.to(dev, BF16).reshape(B * T, W).contiguous()
        idx = torch.arange(T, dtype=torch.int32, device=dev).repeat(B)
        xe = ops.add_rows(xe, self.pos, idx)                                     # + audio_position_embedding (builder.py:136-140)
        xe = _padk(xe, ops.ceil_to(W, 64))
        N = self.nq
        q0 = self.query.repeat(B, 1)
        h = ops.layernorm(q0, self.emb_ln[0], self.emb_ln[1], self.eps)           # BertEmbeddings: LayerNorm(query_embeds)
        for L in self.layers:
            qkv = ops.linear(h, L["sa_qkv"])
            a = torch.empty(B * N, Dm, dtype=BF16, device=dev)
            st = (N * 3 * Dm, 3 * Dm, d)
            ops.attn_prefill(qkv, qkv[:, Dm:], qkv[:, 2 * Dm:], a, B, H, H, N, N, d, st, st, st, Dm, False, 0)
            h = ops.layernorm(ops.linear(a, L["sa_o"], residual=h), L["sa_ln"][0], L["sa_ln"][1], self.eps)
            q = ops.linear(h, L["ca_q"])
            kv = ops.linear(xe, L["ca_kv"])
            ops.attn_prefill(q, kv, kv[:, Dm:], a, B, H, H, N, T, d, (N * Dm, Dm, d), (T * 2 * Dm, 2 * Dm, d), (T * 2 * Dm, 2 * Dm, d), Dm,
                             False, 0)
            h = ops.layernorm(ops.linear(a, L["ca_o"], residual=h), L["ca_ln"][0], L["ca_ln"][1], self.eps)
            f = ops.linear(h, L["fc1"], act="gelu")
            h = ops.layernorm(ops.linear(f, L["fc2"], residual=h), L["ff_ln"][0], L["ff_ln"][1], self.eps)
        return ops.linear(h, self.out).view(B, N, -1)


# =========================================================================================================
# LanguageBind video tower
# =========================================================================================================
class VideoConfig:
    def __init__(self, hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=224,
                 patch_size=14, num_channels=3, num_frames=8, add_time_attn=True, layer_norm_eps=1e-5, hidden_act="quick_gelu", **kw):
        self.hidden_size, self.intermediate_size, self.num_hidden_layers = hidden_size, intermediate_size, num_hidden_layers
        self.num_attention_heads, self.image_size, self.patch_size, self.num_channels = num_attention_heads, image_size, patch_size, num_channels
        self.num_frames, self.add_time_attn, self.layer_norm_eps, self.hidden_act = num_frames, add_time_attn, layer_norm_eps, hidden_act

    @classmethod
    def from_pretrained(cls, path):
        import json
        d = json.load(open(os.path.join(path, "config.json")))
        return cls(**d.get("vision_config", d))


class HipLanguageBindVideoTower:
    def __init__(self, video_tower: Optional[str], args=None, delay_load=False, config: Optional[VideoConfig] = None, device="cuda"):
        self.video_tower_name, self.device, self.dtype = video_tower, torch.device(device), BF16
        self.select_layer = getattr(args, "mm_video_select_layer", -2) if args is not None else -2
        self.config, self.is_loaded, self.video_processor = config, False, None
        if config is None and video_tower is not None and os.path.isdir(str(video_tower)):
            self.config = VideoConfig.from_pretrained(video_tower)
        if self.config is not None:                      # languagebind/__init__.py:205 (LanguageBindVideoProcessor(model.config))
            from .video_processor import HipLanguageBindVideoProcessor
            self.video_processor = HipLanguageBindVideoProcessor(self.config.num_frames, self.config.image_size, device=device)
        if not delay_load and video_tower is not None:
            self.load_model()

    hidden_size = property(lambda self: self.config.hidden_size)
    modal_processor = property(lambda self: self.video_processor)

    @property
    def dummy_inputs(self):
        c = self.config
        return torch.zeros(1, c.num_channels, c.num_frames, c.image_size, c.image_size, device=self.device, dtype=self.dtype)

    def load_model(self):
        if self.is_loaded:
            return
        from .builder import load_base_state_dict
        sd = load_base_state_dict(self.video_tower_name)
        self.load_state_dict({k[len("vision_model."):]: v for k, v in sd.items() if k.startswith("vision_model.")} or sd)

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        for pre in ("video_tower.", "vision_model."):
            if any(k.startswith(pre) for k in sd):
                sd = _strip(sd, pre)
        c, dev = self.config, self.device
        D = c.hidden_size
        t = lambda k: sd[k].to(dev, BF16).contiguous()
        self.patch_w = ops.pack_weight(sd["embeddings.patch_embedding.weight"].to(dev).reshape(D, -1))
        self.cls, self.pos = t("embeddings.class_embedding"), t("embeddings.position_embedding.weight")
        self.pre_ln = (t("pre_layrnorm.weight"), t("pre_layrnorm.bias"))

        def attn(p):
            w = torch.cat([sd[p + f"{n}_proj.weight"] for n in "qkv"], 0).to(dev)
            b = torch.cat([sd[p + f"{n}_proj.bias"] for n in "qkv"], 0).to(dev)
            return ops.pack_weight(w, b), _pw(sd, p + "out_proj.weight", p + "out_proj.bias", dev)
        self.layers = []
        for i in range(c.num_hidden_layers):
            p = f"encoder.layers.{i}."
            if p + "self_attn.q_proj.weight" not in sd:
                break
            L = dict(ln1=(t(p + "layer_norm1.weight"), t(p + "layer_norm1.bias")), ln2=(t(p + "layer_norm2.weight"), t(p + "layer_norm2.bias")),
                     fc1=_pw(sd, p + "mlp.fc1.weight", p + "mlp.fc1.bias", dev), fc2=_pw(sd, p + "mlp.fc2.weight", p + "mlp.fc2.bias", dev))
            L["qkv"], L["out"] = attn(p + "self_attn.")
            if c.add_time_attn:
                L["t_qkv"], L["t_out"] = attn(p + "temporal_attn.")
                L["t_ln"] = (t(p + "temporal_layer_norm1.weight"), t(p + "temporal_layer_norm1.bias"))
                L["t_emb"] = sd[p + "temporal_embedding"].to(dev, BF16).reshape(-1, D).contiguous()
            self.layers.append(L)
        self.is_loaded = True

    def hidden_state(self, videos: torch.Tensor, index: int) -> torch.Tensor:
        c, dev = self.config, self.device
        D, H = c.hidden_size, c.num_attention_heads
        d = D // H
        v = videos.to(dev, BF16)
        B, Cc, T, Hh, Ww = v.shape
        frames = v.permute(0, 2, 1, 3, 4).reshape(B * T, Cc, Hh, Ww)            # 'b c t h w -> (b t) c h w' (:641-643)
        cols, oh, ow = ops.im2col(frames, c.patch_size, c.patch_size, c.patch_size, c.patch_size)
        n = oh * ow + 1
        h = ops.vit_assemble(ops.linear(cols, self.patch_w), self.cls, self.pos, B * T, oh * ow, D).view(B * T * n, D)
        h = ops.layernorm(h, self.pre_ln[0], self.pre_ln[1], c.layer_norm_eps)
        t = c.num_frames if (c.add_time_attn and B * T >= c.num_frames) else 1
        if c.add_time_attn and t != 1 and t != T:
            raise ValueError(f"video has {T} frames but the tower was built for num_frames={t}")
        if c.add_time_attn and t != 1:
            # index tensors of the temporal branch, built once per shape: a host -> device copy from pageable memory is ordered behind
            # everything already queued on the stream and blocks the host until then (round 4: the per-call copies here made the host wait
            # for the previous encoder to finish before it could issue this tower's launches)
            key = ("tidx", B, T, n)
            cache = self.__dict__.setdefault("_idx_cache", {})
            if key not in cache:
                r = np.arange(B * T * n)
                tt = (r // n) % T
                t_idx = torch.from_numpy(tt.astype(np.int32)).to(dev)            # temporal embedding row per (b t n) row
                # (b n t) order for the temporal attention; out_map scatters its output back to (b t n)
                bb, nn_, t2 = np.meshgrid(np.arange(B), np.arange(n), np.arange(T), indexing="ij")
                perm = ((bb * T + t2) * n + nn_).reshape(-1).astype(np.int32)
                cache[key] = (t_idx, torch.from_numpy(perm).to(dev))
            t_idx, perm_d = cache[key]
        st_s = (n * 3 * D, 3 * D, d)
        st_t = (T * 3 * D, 3 * D, d)
        st_tn = (T * n * 3 * D, n * 3 * D, d)                                    # clip stride, frame stride (tokens of one (b, n) sequence), head
        for i in range(index):
            L = self.layers[i]
            if c.add_time_attn:                                                  # modeling_video.py:105-130
                if t != 1:
                    # round 4: `h + temporal_embedding` and temporal_layer_norm1 in one pass, and the temporal attention IN PLACE over the
                    # (b t) n d layout - sequence (b, n) starts at row b T n + n', its tokens lie n rows apart - instead of through a
                    # permuted copy of the hidden state (rows are independent in the LayerNorm and the GEMM: same values, no permutation)
                    h, nrm = ops.add_layernorm(h, L["t_emb"], t_idx, L["t_ln"][0], L["t_ln"][1], c.layer_norm_eps)
                    qkv = ops.linear(nrm, L["t_qkv"])
                    a = torch.empty(B * T * n, D, dtype=BF16, device=dev)
                    ops.attn_prefill(qkv, qkv[:, D:], qkv[:, 2 * D:], a, B * n, H, H, T, T, d, st_tn, st_tn, st_tn, D, False, 0,
                                     scale=d ** -0.5, out_map=perm_d, batch_split=(n, 3 * D))
                else:                                                            # bt < t: every token attends to itself only
                    qkv = ops.linear(ops.layernorm(h, L["t_ln"][0], L["t_ln"][1], c.layer_norm_eps), L["t_qkv"])
                    a = torch.empty(B * T * n, D, dtype=BF16, device=dev)
                    s1 = (3 * D, 3 * D, d)
                    ops.attn_prefill(qkv, qkv[:, D:], qkv[:, 2 * D:], a, B * T * n, H, H, 1, 1, d, s1, s1, s1, D, False, 0, scale=d ** -0.5)
                h = ops.linear(a, L["t_out"], residual=h)
            nrm = ops.layernorm(h, L["ln1"][0], L["ln1"][1], c.layer_norm_eps)
            qkv = ops.linear(nrm, L["qkv"])
            a = torch.empty(B * T * n, D, dtype=BF16, device=dev)
            ops.attn_prefill(qkv, qkv[:, D:], qkv[:, 2 * D:], a, B * T, H, H, n, n, d, st_s, st_s, st_s, D, False, 0, scale=d ** -0.5)
            h = ops.linear(a, L["out"], residual=h)
            f = ops.linear(ops.layernorm(h, L["ln2"][0], L["ln2"][1], c.layer_norm_eps), L["fc1"], act=c.hidden_act)
            h = ops.linear(f, L["fc2"], residual=h)
        return h.view(B, T, n, D)

    def __call__(self, videos):
        return self.forward(videos)

    def forward(self, videos):
        """hidden_states[select_layer] shaped (b, t, n, c), class token included (languagebind/__init__.py:209-233)."""
        if type(videos) is list:
            return [self.forward(v.unsqueeze(0)) for v in videos]
        n_hs = self.config.num_hidden_layers + 1
        idx = self.select_layer if self.select_layer >= 0 else n_hs + self.select_layer
        return self.hidden_state(videos, idx)


# =========================================================================================================
# PointBERT
# =========================================================================================================
class PointConfig:
    """pointbert/PointTransformer_8192point_2layer.yaml with point_dims = 6 (point_encoder.py:24-29)."""

    def __init__(self, trans_dim=384, depth=12, num_heads=6, group_size=32, num_group=512, encoder_dims=256, point_dims=6,
                 use_max_pool=False, **kw):
        self.trans_dim, self.depth, self.num_heads, self.group_size = trans_dim, depth, num_heads, group_size
        self.num_group, self.encoder_dims, self.point_dims, self.use_max_pool = num_group, encoder_dims, point_dims, use_max_pool


class PointCloudProcessor:
    """point_encoder.py:87-113: file loading is the whole processor (pc_norm is defined but never applied by __call__)."""

    def __call__(self, pc_files):
        if isinstance(pc_files, str):
            pc_files = [pc_files]
        return torch.from_numpy(np.stack([np.load(f) for f in pc_files], axis=0).astype(np.float32))

    def pc_norm(self, pc):
        xyz, other = pc[:, :3], pc[:, 3:]
        xyz = xyz - np.mean(xyz, axis=0)
        return np.concatenate((xyz / np.max(np.sqrt(np.sum(xyz ** 2, axis=1))), other), axis=1)


class HipPointEncoder:
    def __init__(self, point_encoder: Optional[str], args=None, delay_load=False, config: Optional[PointConfig] = None, device="cuda"):
        self.point_encoder_name, self.device, self.dtype = point_encoder, torch.device(device), BF16
        self.cfg, self.is_loaded, self.point_processor = config or PointConfig(), False, PointCloudProcessor()
        self.fps_start = None            # optional LongTensor (B,): first FPS index; default = torch.randint like misc.py:52
        if self.cfg.use_max_pool:
            raise NotImplementedError("use_max_pool=True is not the released configuration")
        if not delay_load and point_encoder is not None:
            self.load_model()

    config = property(lambda self: self.cfg)
    hidden_size = property(lambda self: self.cfg.trans_dim)
    modal_processor = property(lambda self: self.point_processor)

    @property
    def dummy_inputs(self):
        return torch.zeros(1, 8192, 6, device=self.device, dtype=self.dtype)

    def load_model(self):
        if not self.is_loaded:
            self.load_state_dict(load_tensors(self.point_encoder_name))

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        if any(k.startswith("point_encoder.") for k in sd):
            sd = _strip(sd, "point_encoder.")
        c, dev = self.cfg, self.device
        t = lambda k: sd[k].to(dev, BF16).contiguous()

        def conv_bn(conv, bn):          # 1x1 conv followed by eval-mode (Sync)BatchNorm folded into one affine (dvae.py:193-207)
            w, b = sd[conv + ".weight"].float().squeeze(-1), sd[conv + ".bias"].float()
            s = sd[bn + ".weight"].float() / torch.sqrt(sd[bn + ".running_var"].float() + 1e-5)
            return ops.pack_weight((w * s[:, None]).to(dev), ((b - sd[bn + ".running_mean"].float()) * s + sd[bn + ".bias"].float()).to(dev))

        def conv(name):
            return ops.pack_weight(sd[name + ".weight"].squeeze(-1).to(dev), sd[name + ".bias"].to(dev))
        self.c1 = conv_bn("encoder.first_conv.0", "encoder.first_conv.1")
        self.c2 = conv("encoder.first_conv.3")
        self.c3 = conv_bn("encoder.second_conv.0", "encoder.second_conv.1")
        self.c4 = conv("encoder.second_conv.3")
        self.reduce = _pw(sd, "reduce_dim.weight", "reduce_dim.bias", dev)
        self.pos0, self.pos2 = _pw(sd, "pos_embed.0.weight", "pos_embed.0.bias", dev), _pw(sd, "pos_embed.2.weight", "pos_embed.2.bias", dev)
        self.cls_token = sd["cls_token"].to(dev, BF16).reshape(-1).contiguous()
        self.cls_pos = sd["cls_pos"].to(dev, BF16).reshape(1, -1).contiguous()
        self.blocks = []
        for i in range(c.depth):
            p = f"blocks.blocks.{i}."
            self.blocks.append(dict(n1=(t(p + "norm1.weight"), t(p + "norm1.bias")), n2=(t(p + "norm2.weight"), t(p + "norm2.bias")),
                                    qkv=_pw(sd, p + "attn.qkv.weight", p + "attn.qkv.bias", dev),
                                    proj=_pw(sd, p + "attn.proj.weight", p + "attn.proj.bias", dev),
                                    fc1=_pw(sd, p + "mlp.fc1.weight", p + "mlp.fc1.bias", dev),
                                    fc2=_pw(sd, p + "mlp.fc2.weight", p + "mlp.fc2.bias", dev)))
        self.norm = (t("norm.weight"), t("norm.bias"))
        self.is_loaded = True

    def __call__(self, point_clouds):
        return self.forward(point_clouds)

    def forward(self, point_clouds, return_aux=False):
        """(B, G+1, trans_dim) like PointEncoder.forward (point_encoder.py:48-50)."""
        c, dev = self.cfg, self.device
        pts = point_clouds.to(dev, BF16).contiguous()
        B, N, Cc = pts.shape
        G, M, Dm, H = c.num_group, c.group_size, c.trans_dim, c.num_heads
        if self.fps_start is not None:
            start = self.fps_start.to(torch.int32).to(dev)
        else:
            start = torch.randint(0, N, (B,), dtype=torch.long).to(torch.int32).to(dev)    # misc.py:52
        cidx, centers = ops.fps(pts, G, start)
        nb, nidx = ops.knn_group(pts, centers, M, Kp=64)                         # [(b g m), 64] centred xyz | rgb | 0
        f = ops.linear(nb, self.c1, act="relu")
        cat = torch.empty(B * G * M, 512, dtype=BF16, device=dev)
        ops.linear(_padk(f, self.c2.Kp), self.c2, out=cat[:, 256:])
        ops.group_max(cat[:, 256:], B * G, M, bcast=cat[:, :256])                # cat([global.expand, feature]) (dvae.py:217-218)
        f = ops.linear(cat, self.c3, act="relu")
        f = ops.linear(f, self.c4)
        tok = ops.group_max(f, B * G, M)                                         # [(b g), encoder_dims]
        tok = ops.linear(_padk(tok, self.reduce.Kp), self.reduce)
        cen = ops.f32_rows_to_bf16(centers.view(B * G, 3), 64)
        pos = ops.linear(ops.linear(cen, self.pos0, act="gelu"), self.pos2)      # pos_embed (:140-144)
        x = ops.vit_assemble(tok, self.cls_token, None, B, G, Dm).view(B * (G + 1), Dm)
        posf = ops.vit_assemble(pos, self.cls_pos.view(-1), None, B, G, Dm).view(B * (G + 1), Dm)
        T = G + 1
        d = Dm // H
        st = (T * 3 * Dm, 3 * Dm, d)
        for L in self.blocks:
            x = ops.add(x, posf)                                                 # block(x + pos) (:95-98)
            qkv = ops.linear(ops.layernorm(x, L["n1"][0], L["n1"][1], 1e-5), L["qkv"])
            a = torch.empty(B * T, Dm, dtype=BF16, device=dev)
            ops.attn_prefill(qkv, qkv[:, Dm:], qkv[:, 2 * Dm:], a, B, H, H, T, T, d, st, st, st, Dm, False, 0, scale=d ** -0.5)
            x = ops.linear(a, L["proj"], residual=x)
            f = ops.linear(ops.layernorm(x, L["n2"][0], L["n2"][1], 1e-5), L["fc1"], act="gelu")
            x = ops.linear(f, L["fc2"], residual=x)
        x = ops.layernorm(x, self.norm[0], self.norm[1], 1e-5).view(B, T, Dm)
        if return_aux:
            return x, cidx, nidx, centers
        return x


# =========================================================================================================
# builders used by model/builder.py
# =========================================================================================================
def build(modal: str, cfg, dev, delay_load=True, config=None):
    """multimodal_encoder/builder.py:86-117 for audio / video / point.  `config` (a dict) replaces the configuration the
    reference reads from the encoder checkpoint / directory (in-memory construction: tests, synthetic benchmarks)."""
    if modal == "audio":
        path = cfg.mm_audio_encoder
        if "VideoLLaMA" in str(path):
            from .imagebind_audio import HipImageBindAudioEncoder
            enc = HipImageBindAudioEncoder(path, cfg, delay_load=delay_load, config=config, device=dev)
            return enc, enc.hidden_size
        enc = HipBeatsAudioEncoder(path, cfg, delay_load=delay_load, config=BeatsConfig(config) if config is not None else None, device=dev)
        return enc, (enc.cfg.encoder_embed_dim if enc.cfg is not None else None)
    if modal == "video":
        path = cfg.mm_video_encoder
        if config is None and os.path.isdir(str(path)) is False and not str(path).endswith("LanguageBind_Video_merge"):
            raise ValueError(f"Unknown video encoder: {path}")                  # builder.py:97-101
        enc = HipLanguageBindVideoTower(path if os.path.isdir(str(path)) else None, cfg, delay_load=delay_load,
                                        config=VideoConfig(**config) if config is not None else None, device=dev)
        return enc, (enc.config.hidden_size if enc.config is not None else None)
    if modal == "point":
        enc = HipPointEncoder(cfg.mm_point_encoder if os.path.isfile(str(cfg.mm_point_encoder)) else None, cfg, delay_load=delay_load,
                              config=PointConfig(**config) if config is not None else None, device=dev)
        return enc, enc.cfg.trans_dim
    raise ValueError(f"unknown modality {modal}")


def build_qformer_projector(cfg, ptype: str, hidden, dev, config=None):
    """'qformer_{N}N_{L}L' (multimodal_projector/builder.py:217-220).  `config` overrides the BERT-base widths."""
    import re
    m = re.match(r"^qformer_(\d+)N_(\d+)L$", ptype)
    if not m:
        raise ValueError(f"Unknown projector type: {ptype}")
    kw = {}
    if config:
        kw = {k: config[k] for k in ("hidden_size", "num_attention_heads", "intermediate_size", "layer_norm_eps", "num_positions")
              if k in config}
    return HipQformerProjector(int(m.group(1)), hidden, int(m.group(2)), device=dev, **kw)


def build_audio_qformer(cfg, dev, num_positions=8, config=None):
    kw = dict(config or {})
    kw.pop("out_features", None)
    kw.setdefault("num_positions", num_positions)
    if "encoder_width" in kw:
        kw["vision_width"] = kw.pop("encoder_width")
    return HipQformerProjector(device=dev, **kw)
