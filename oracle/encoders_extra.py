"""Oracle (TEST INFRASTRUCTURE): audio / video / point encoders and the Q-Former projector, torch CPU fp32.

Restates (file:line under /root/reference/modelcompose/model):
  BEATs            multimodal_encoder/beats/BEATs.py:120-189, beats/backbone.py:64-189 (encoder), :192-315 (layer),
                   :431-468 (T5-style buckets), :471-717 (attention with gated relative position bias)
  Q-Former         multimodal_projector/builder.py:111-173 (VideoLlamaAudioQformer), multimodal_projector/Qformer.py:51-110,
                   :112-277, :379-486 (BERT blocks with cross-attention, query FFN)
  LanguageBind-V   multimodal_encoder/languagebind/video/modeling_video.py:65-161 (temporal + spatial block), :599-678
  PointBERT        multimodal_encoder/pointbert/point_encoder.py:11-189, pointbert/dvae.py:107-223, pointbert/misc.py:40-60
The CLIP attention/MLP blocks LanguageBind imports from transformers==4.31 are third-party (restated in oracle/encoders.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .encoders import _act, mha


def _ln(x, sd, p, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[f"{p}.weight"], sd[f"{p}.bias"], eps)


def _lin(x, sd, p):
    return F.linear(x, sd[f"{p}.weight"], sd.get(f"{p}.bias"))


# =========================================================================================================
# BEATs
# =========================================================================================================
def beats_rel_buckets(q_len, k_len, num_buckets, max_distance):
    """_relative_positions_bucket, bidirectional (beats/backbone.py:431-456)."""
    ctx = torch.arange(q_len, dtype=torch.long)[:, None]
    mem = torch.arange(k_len, dtype=torch.long)[None, :]
    rel = mem - ctx
    nb = num_buckets // 2
    buckets = (rel > 0).to(torch.long) * nb
    rel = torch.abs(rel)
    max_exact = nb // 2
    is_small = rel < max_exact
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, nb - 1))
    return buckets + torch.where(is_small, rel, large)


def beats_pos_conv_weight(sd, p="encoder.pos_conv.0"):
    """weight-norm parametrisation with dim=2 (beats/backbone.py:28-47): w = v * g / ||v||_{dims 0,1}."""
    if f"{p}.parametrizations.weight.original0" in sd:
        g, v = sd[f"{p}.parametrizations.weight.original0"], sd[f"{p}.parametrizations.weight.original1"]
    elif f"{p}.weight_g" in sd:
        g, v = sd[f"{p}.weight_g"], sd[f"{p}.weight_v"]
    else:
        return sd[f"{p}.weight"]
    norm = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    return v * (g / norm)


def beats_attention(x, sd, p, cfg, key_padding_mask, pos_bias):
    """MultiheadAttention.forward for self-attention (beats/backbone.py:471-717).  x: (B, T, C).
    The (q/32, (s - max)*32) rescaling of :552-554,663 is an exact no-op for the softmax and is not repeated."""
    B, T, Cdim = x.shape
    H = cfg["encoder_attention_heads"]
    d = Cdim // H
    q_raw = _lin(x, sd, f"{p}.q_proj")
    k = _lin(x, sd, f"{p}.k_proj")
    v = _lin(x, sd, f"{p}.v_proj")
    q = (q_raw * d ** -0.5).view(B, T, H, d).transpose(1, 2)
    k = k.view(B, T, H, d).transpose(1, 2)
    v = v.view(B, T, H, d).transpose(1, 2)
    w = q @ k.transpose(-1, -2)                                              # (B,H,T,T)
    if key_padding_mask is not None:
        w = w.masked_fill(key_padding_mask[:, None, None, :].to(torch.bool), float("-inf"))
    if pos_bias is not None:
        bias = pos_bias[None]                                                # (1,H,T,T)
        if cfg.get("gru_rel_pos", False):
            ql = q_raw.view(B, T, H, d).transpose(1, 2)                       # == q * alpha / scaling  (:690)
            g = torch.sigmoid(_lin(ql, sd, f"{p}.grep_linear").view(B, H, T, 2, 4).sum(-1))
            ga, gb = g.chunk(2, dim=-1)
            gate = ga * (gb * sd[f"{p}.grep_a"] - 1.0) + 2.0                   # (B,H,T,1)
            bias = gate * bias
        w = w + bias
    w = F.softmax(w, dim=-1)
    o = (w @ v).transpose(1, 2).reshape(B, T, Cdim)
    return _lin(o, sd, f"{p}.out_proj")


def beats_encode(fbank, padding_mask, sd, cfg):
    """BEATs.extract_features_new (BEATs.py:149-189) + TransformerEncoder.extract_features (backbone.py:148-189).
    fbank (B, frames, 128); padding_mask (B, frames) bool or None.  Returns (features (B,T,C), pooled padding mask)."""
    ps = cfg["input_patch_size"]
    x = F.conv2d(fbank.unsqueeze(1), sd["patch_embedding.weight"], sd.get("patch_embedding.bias"), stride=ps)
    x = x.reshape(x.shape[0], x.shape[1], -1).transpose(1, 2)                 # B x T x 512
    x = _ln(x, sd, "layer_norm")
    if padding_mask is not None:                                               # forward_padding_mask (:120-132)
        extra = padding_mask.size(1) % x.size(1)
        pm = padding_mask[:, :-extra] if extra > 0 else padding_mask
        padding_mask = pm.view(pm.size(0), x.size(1), -1).all(-1)
    if "post_extract_proj.weight" in sd:
        x = _lin(x, sd, "post_extract_proj")
    if padding_mask is not None:
        x = x.clone()
        x[padding_mask] = 0                                                    # backbone.py:150-151
    # convolutional position embedding (:71-85,153-155): conv1d(k, pad k//2, groups) -> drop last (SamePad) -> GELU
    w = beats_pos_conv_weight(sd)
    kpos = cfg["conv_pos"]
    xc = F.conv1d(x.transpose(1, 2), w, sd["encoder.pos_conv.0.bias"], padding=kpos // 2, groups=cfg["conv_pos_groups"])
    if kpos % 2 == 0:
        xc = xc[:, :, :-1]
    x = x + F.gelu(xc).transpose(1, 2)
    lnf = cfg.get("layer_norm_first", False)
    if not lnf:
        x = _ln(x, sd, "encoder.layer_norm")
    L = cfg["encoder_layers"]
    alpha = math.pow(2 * L, 0.25) if cfg.get("deep_norm", False) else 1.0      # :249
    T = x.shape[1]
    pos_bias = None
    if cfg.get("relative_position_embedding", False):                         # computed in layer 0, shared (:117-120,170-176)
        buckets = beats_rel_buckets(T, T, cfg["num_buckets"], cfg["max_distance"])
        pos_bias = sd["encoder.layers.0.self_attn.relative_attention_bias.weight"][buckets].permute(2, 0, 1)
    for i in range(L):
        p = f"encoder.layers.{i}"
        if lnf:
            r = x
            h = beats_attention(_ln(x, sd, f"{p}.self_attn_layer_norm"), sd, f"{p}.self_attn", cfg, padding_mask, pos_bias)
            x = r + h
            r = x
            h = _lin(_act(_lin(_ln(x, sd, f"{p}.final_layer_norm"), sd, f"{p}.fc1"), "gelu"), sd, f"{p}.fc2")
            x = r + h
        else:                                                                  # post-LN / deep-norm (:292-313)
            h = beats_attention(x, sd, f"{p}.self_attn", cfg, padding_mask, pos_bias)
            x = _ln(x * alpha + h, sd, f"{p}.self_attn_layer_norm")
            h = _lin(_act(_lin(x, sd, f"{p}.fc1"), "gelu"), sd, f"{p}.fc2")
            x = _ln(x * alpha + h, sd, f"{p}.final_layer_norm")
    if lnf:
        x = _ln(x, sd, "encoder.layer_norm")
    return x, padding_mask


# =========================================================================================================
# Q-Former projector
# =========================================================================================================
# training only: the nn.Dropout modules of the Q-Former (BertConfig defaults 0.1: Qformer.py:67/108 embeddings, :136/259 attention
# probabilities, :284/288 BertSelfOutput, :370/374 BertOutput).  QFORMER_DROPOUT(tag, tensor) -> dropped tensor; None in eval mode.  Tags:
# "emb", (layer, "self.probs" | "self.out" | "cross.probs" | "cross.out" | "ffn.out").  oracle tests pass the HIP step's Philox masks.
QFORMER_DROPOUT = None


def _qdrop(tag, x):
    return x if QFORMER_DROPOUT is None else QFORMER_DROPOUT(tag, x)


def _bert_attn(x, kv, sd, p, H, eps, tag=None):
    """BertAttention: BertSelfAttention (Qformer.py:176-277; scores / sqrt(d), no mask terms for all-ones masks; dropout on the
    probabilities :259) + BertSelfOutput (dense, dropout, LayerNorm(h + input), :280-291)."""
    B, L, Dm = x.shape
    d = Dm // H
    q = _lin(x, sd, f"{p}.self.query").view(B, L, H, d).transpose(1, 2)
    k = _lin(kv, sd, f"{p}.self.key").view(B, kv.shape[1], H, d).transpose(1, 2)
    v = _lin(kv, sd, f"{p}.self.value").view(B, kv.shape[1], H, d).transpose(1, 2)
    w = F.softmax((q @ k.transpose(-1, -2)) / math.sqrt(d), dim=-1)
    if tag is not None:
        w = _qdrop((tag[0], tag[1] + ".probs"), w)
    o = (w @ v).transpose(1, 2).reshape(B, L, Dm)
    y = _lin(o, sd, f"{p}.output.dense")
    if tag is not None:
        y = _qdrop((tag[0], tag[1] + ".out"), y)
    return _ln(y + x, sd, f"{p}.output.LayerNorm", eps)


def qformer_project(x, sd, cfg, prefix=""):
    """VideoLlamaAudioQformer.forward (multimodal_projector/builder.py:130-155).  x: (B, T, encoder_width)."""
    pf = prefix
    sub = {k[len(pf):]: v for k, v in sd.items() if k.startswith(pf)} if pf else sd
    B, T, _ = x.shape
    eps = cfg.get("layer_norm_eps", 1e-12)
    H = cfg["num_attention_heads"]
    x = x + sub["audio_position_embedding.weight"][:T][None]                  # :136-140
    q = sub["audio_query_tokens"].expand(B, -1, -1)
    h = _ln(q, sub, "audio_Qformer.bert.embeddings.LayerNorm", eps)            # BertEmbeddings with query_embeds only (:79-110)
    h = _qdrop("emb", h)                                                       # :108
    for i in range(cfg["num_hidden_layers"]):
        p = f"audio_Qformer.bert.encoder.layer.{i}"
        h = _bert_attn(h, h, sub, f"{p}.attention", H, eps, tag=(i, "self"))
        h = _bert_attn(h, x, sub, f"{p}.crossattention", H, eps, tag=(i, "cross"))   # cross_attention_freq = 1
        f = F.gelu(_lin(h, sub, f"{p}.intermediate_query.dense"))              # query FFN (:482-485)
        h = _ln(_qdrop((i, "ffn.out"), _lin(f, sub, f"{p}.output_query.dense")) + h, sub, f"{p}.output_query.LayerNorm", eps)
    return _lin(h, sub, "audio_llama_proj")


# =========================================================================================================
# LanguageBind video tower
# =========================================================================================================
def languagebind_video_hidden_states(video, sd, cfg, n_layers=None, prefix=""):
    """CLIPVisionTransformer.forward of languagebind/video/modeling_video.py:613-678; video (B, C, T, H, W).
    Returns the hidden_states list, each (B, T, n, c)."""
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)} if prefix else sd
    B, Cc, T, Hh, Ww = video.shape
    Dm, Hn, eps = cfg["hidden_size"], cfg["num_attention_heads"], cfg.get("layer_norm_eps", 1e-5)
    px = video.permute(0, 2, 1, 3, 4).reshape(B * T, Cc, Hh, Ww)                # 'b c t h w -> (b t) c h w'
    x = F.conv2d(px, sub["embeddings.patch_embedding.weight"], None, stride=cfg["patch_size"]).flatten(2).transpose(1, 2)
    x = torch.cat([sub["embeddings.class_embedding"].expand(B * T, 1, -1), x], 1) + sub["embeddings.position_embedding.weight"][None]
    x = _ln(x, sub, "pre_layrnorm", eps)
    n = x.shape[1]
    hs = [x]
    L = cfg["num_hidden_layers"] if n_layers is None else n_layers
    act = cfg.get("hidden_act", "quick_gelu")
    for i in range(L):
        p = f"encoder.layers.{i}"
        if cfg.get("add_time_attn", False):                                    # :105-130
            t = cfg["num_frames"]
            if B * T < t:
                t = 1
            xt = x.view(B, t, n, Dm).transpose(1, 2).reshape(B * n, t, Dm) if t == T else x.view(-1, 1, Dm)
            if t != 1:
                xt = xt + sub[f"{p}.temporal_embedding"][:, :t, :]
            res = xt
            h = mha(_ln(xt, sub, f"{p}.temporal_layer_norm1", eps), sub, f"{p}.temporal_attn", Hn)
            xt = res + h
            x = xt.view(B, n, t, Dm).transpose(1, 2).reshape(B * t, n, Dm) if t == T else xt.view(B * T, n, Dm)
        r = x
        x = r + mha(_ln(x, sub, f"{p}.layer_norm1", eps), sub, f"{p}.self_attn", Hn)
        r = x
        h = _lin(_act(_lin(_ln(x, sub, f"{p}.layer_norm2", eps), sub, f"{p}.mlp.fc1"), act), sub, f"{p}.mlp.fc2")
        x = r + h
        hs.append(x)
    return [h.view(B, T, n, Dm) for h in hs]


def languagebind_video_tower(video, sd, cfg, select_layer=-2, prefix=""):
    """LanguageBindVideoTower.forward + feature_select (languagebind/__init__.py:209-233): hidden_states[select_layer], class
    token included, (b, t, n, c)."""
    n_hs = cfg["num_hidden_layers"] + 1
    idx = select_layer if select_layer >= 0 else n_hs + select_layer
    return languagebind_video_hidden_states(video, sd, cfg, n_layers=idx, prefix=prefix)[idx]


# =========================================================================================================
# PointBERT
# =========================================================================================================
def fps_indices(xyz, npoint, start):
    """misc.fps (pointbert/misc.py:40-60) with an explicit first index instead of torch.randint."""
    B, N, _ = xyz.shape
    cent = torch.zeros(B, npoint, dtype=torch.long)
    distance = torch.ones(B, N) * 1e10
    far = start.clone().long()
    bi = torch.arange(B)
    for i in range(npoint):
        cent[:, i] = far
        c = xyz[bi, far, :].view(B, 1, 3)
        dist = torch.sum((xyz - c) ** 2, -1)
        distance = torch.min(distance, dist)
        far = torch.max(distance, -1)[1]
    return cent


def _bn_eval(x, sd, p, eps=1e-5):
    """SyncBatchNorm in eval mode = per-channel affine with running statistics (dvae.py:197,204). x: (.., C, n)."""
    w, b, m, v = sd[f"{p}.weight"], sd[f"{p}.bias"], sd[f"{p}.running_mean"], sd[f"{p}.running_var"]
    return (x - m[:, None]) / torch.sqrt(v[:, None] + eps) * w[:, None] + b[:, None]


def pointbert_encode(pts, sd, cfg, fps_start, return_aux=False):
    """PointTransformer.forward (pointbert/point_encoder.py:169-189), use_max_pool=False path.  pts (B, N, 3+c)."""
    B, N, Cc = pts.shape
    G, M = cfg["num_group"], cfg["group_size"]
    xyz = pts[:, :, :3]
    cidx = fps_indices(xyz, G, fps_start)
    center = torch.gather(xyz, 1, cidx[:, :, None].expand(B, G, 3))            # index_points
    # knn_point (dvae.py:107-141): squared distances, k smallest (order irrelevant: max-pooled later)
    d = -2 * torch.matmul(center, xyz.permute(0, 2, 1)) + torch.sum(center ** 2, -1)[:, :, None] + torch.sum(xyz ** 2, -1)[:, None, :]
    idx = torch.topk(d, M, dim=-1, largest=False, sorted=False)[1]              # (B,G,M)
    nb = torch.gather(pts[:, None].expand(B, G, N, Cc), 2, idx[..., None].expand(B, G, M, Cc))
    nb = torch.cat([nb[..., :3] - center[:, :, None, :], nb[..., 3:]], dim=-1)  # centre-subtract xyz, keep the rest
    # mini-PointNet (dvae.py:189-223)
    f = nb.reshape(B * G, M, Cc).transpose(2, 1)
    f = F.conv1d(f, sd["encoder.first_conv.0.weight"], sd["encoder.first_conv.0.bias"])
    f = F.relu(_bn_eval(f, sd, "encoder.first_conv.1"))
    f = F.conv1d(f, sd["encoder.first_conv.3.weight"], sd["encoder.first_conv.3.bias"])
    fg = f.max(dim=2, keepdim=True)[0]
    f = torch.cat([fg.expand(-1, -1, M), f], dim=1)
    f = F.conv1d(f, sd["encoder.second_conv.0.weight"], sd["encoder.second_conv.0.bias"])
    f = F.relu(_bn_eval(f, sd, "encoder.second_conv.1"))
    f = F.conv1d(f, sd["encoder.second_conv.3.weight"], sd["encoder.second_conv.3.bias"])
    tok = f.max(dim=2)[0].reshape(B, G, -1)
    tok = _lin(tok, sd, "reduce_dim")
    Dm = tok.shape[-1]
    pos = _lin(F.gelu(_lin(center, sd, "pos_embed.0")), sd, "pos_embed.2")
    x = torch.cat([sd["cls_token"].expand(B, -1, -1), tok], 1)
    pos = torch.cat([sd["cls_pos"].expand(B, -1, -1), pos], 1)
    H = cfg["num_heads"]
    dh = Dm // H
    for i in range(cfg["depth"]):                                              # x = block(x + pos) (:95-98)
        p = f"blocks.blocks.{i}"
        x = x + pos
        h = _ln(x, sd, f"{p}.norm1")
        qkv = F.linear(h, sd[f"{p}.attn.qkv.weight"], sd.get(f"{p}.attn.qkv.bias")).reshape(B, -1, 3, H, dh).permute(2, 0, 3, 1, 4)
        a = F.softmax((qkv[0] @ qkv[1].transpose(-2, -1)) * dh ** -0.5, dim=-1) @ qkv[2]
        x = x + _lin(a.transpose(1, 2).reshape(B, -1, Dm), sd, f"{p}.attn.proj")
        x = x + _lin(F.gelu(_lin(_ln(x, sd, f"{p}.norm2"), sd, f"{p}.mlp.fc1")), sd, f"{p}.mlp.fc2")
    x = _ln(x, sd, "norm")
    if return_aux:
        return x, cidx, idx, center
    return x


# =========================================================================================================
# ImageBind audio branch  (imagebind/imagebind_model.py:186-203, 342-349, 402-406, 436-439, 493-527; imagebind/transformer.py:94-97,
# 105-173; imagebind/multimodal_preprocessors.py:120-160, 205-316)
# =========================================================================================================
def imagebind_audio_encode(x, sd, cfg, return_cls=False):
    """x (B, S, 1, mel, frames) -> (B, S, out_embed_dim): per clip conv(k, stride, no bias) -> LayerNorm -> [cls | patches] + pos ->
    pre-LN blocks with nn.MultiheadAttention(add_bias_kv) -> LayerNorm(1e-6) -> token 0 -> Linear(no bias) -> L2 normalise x exp(log_scale).
    Clips are folded into the batch (:496-503) and unfolded at the end (:522-525)."""
    B, S = x.shape[:2]
    v = x.reshape(B * S, *x.shape[2:]).float()
    pp, tr = "modality_preprocessors.audio.", "modality_trunks.audio."
    t = F.conv2d(v, sd[pp + "rgbt_stem.proj.weight"], None, stride=cfg["audio_stride"])
    t = t.flatten(2).transpose(1, 2)                                                    # B (T)HW C   (PatchEmbedGeneric.forward)
    t = _ln(t, sd, pp + "rgbt_stem.norm_layer", 1e-5)
    t = torch.cat([sd[pp + "cls_token"].expand(t.shape[0], -1, -1), t], dim=1)
    pos = sd[pp + "pos_embedding_helper.pos_embed"]
    assert pos.shape[1] == t.shape[1], "input size differs from the trained grid: position interpolation is not restated"
    t = t + pos
    E, H = cfg["audio_embed_dim"], cfg["audio_num_heads"]
    d = E // H
    n = t.shape[0]
    for i in range(cfg["audio_num_blocks"]):
        p = f"{tr}blocks.{i}."
        h = _ln(t, sd, p + "norm_1", 1e-6)
        qkv = F.linear(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"])
        q, k, vv = qkv.split(E, dim=-1)
        # add_bias_kv: one learned key / value appended AFTER the input projection (torch MultiheadAttention)
        k = torch.cat([k, sd[p + "attn.bias_k"].reshape(1, 1, E).expand(n, -1, -1)], dim=1)
        vv = torch.cat([vv, sd[p + "attn.bias_v"].reshape(1, 1, E).expand(n, -1, -1)], dim=1)
        sh = lambda z: z.reshape(n, -1, H, d).transpose(1, 2)
        a = torch.softmax((sh(q) * d ** -0.5) @ sh(k).transpose(-1, -2), dim=-1) @ sh(vv)
        a = a.transpose(1, 2).reshape(n, -1, E)
        t = t + F.linear(a, sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"])
        h = _ln(t, sd, p + "norm_2", 1e-6)
        h = F.linear(F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        t = t + h
    cls = _ln(t, sd, "modality_heads.audio.0", 1e-6)[:, 0]
    y = F.linear(cls, sd["modality_heads.audio.2.weight"])
    y = F.normalize(y, dim=-1, p=2) * torch.clip(sd["modality_postprocessors.audio.1.log_logit_scale"].exp(), max=100.0)
    y = y.reshape(B, S, -1)
    return (cls.reshape(B, S, -1), y) if return_cls else y


# =========================================================================================================
# dispatch used by oracle.pipeline.OracleModel.encode_modal
# =========================================================================================================
def encode(model, modal: str, x):
    sd, meta = model.sd, model.meta
    if modal == "audio" and "imagebind" in meta:
        sub = model._sub("model.modal_encoders.audio.")
        feats = imagebind_audio_encode(x, sub, meta["imagebind"])
        return qformer_project(feats, sd, meta["qformer"], prefix="model.modal_projectors.audio.")
    if modal == "audio":
        sub = model._sub("model.modal_encoders.audio.audio_encoder.")
        feats, _ = beats_encode(x["audio_inputs"], x.get("audio_padding_mask"), sub, meta["beats"])
        return qformer_project(feats, sd, meta["qformer"], prefix="model.modal_projectors.audio.")
    if modal == "video":
        f = languagebind_video_tower(x, sd, meta["video"], meta.get("mm_video_select_layer", -2),
                                     prefix="model.modal_encoders.video.video_tower.")
        b, t, n, d = f.shape
        from .encoders import projector
        return projector(f.reshape(b, t * n, d), sd, "model.modal_projectors.video", meta.get("mm_video_projector_type", "linear"))
    if modal == "point":
        sub = model._sub("model.modal_encoders.point.point_encoder.")
        f = pointbert_encode(x, sub, meta["point"], torch.as_tensor(meta.get("fps_start", [0] * x.shape[0])))
        from .encoders import projector
        return projector(f, sd, "model.modal_projectors.point", meta.get("mm_point_projector_type", "linear"))
    raise ValueError(f"unknown modality {modal}")
