"""Synthetic (seeded, random-init) checkpoints in the reference's state_dict key grammar, generated directly in
HBM — there is no network for real weights: weights N(0, 0.02²), LoRA A kaiming-uniform(a=√5),
LoRA B N(0, 0.01²) (non-zero so composition matters), prefix/suffix N(0, 0.02²), norm gains 1."""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch


def vicuna7b_meta(modals: Sequence[str] = ("vision",), reset: Optional[str] = None, layers: int = 32, prefix: int = 5) -> dict:
    meta = dict(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=layers, num_attention_heads=32,
                num_key_value_heads=32, max_position_embeddings=4096, rms_norm_eps=1e-5, lora_r=128, lora_alpha=256,
                lora_strategy="modal+language", reset_scaling_weights=reset, local_prefix_tokens=prefix, local_suffix_tokens=prefix,
                pad_token_id=0, eos_token_id=2, mm_projector_type="mlp2x_gelu", mm_vision_select_layer=-2,
                mm_vision_select_feature="patch")
    for m in modals:
        meta[f"mm_{m}_encoder"] = f"synthetic/{m}"
    if "vision" in modals:
        meta["clip"] = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336,
                            patch_size=14, layer_norm_eps=1e-5, hidden_act="quick_gelu")
    if "audio" in modals:      # BEATs_iter3+ (beats/BEATs.py:25-65 + checkpoint cfg) and the 32-query, 2-layer Q-Former (run_finetune_audio_damc.sh:37-38)
        meta["beats"] = dict(input_patch_size=16, embed_dim=512, conv_bias=False, encoder_layers=12, encoder_embed_dim=768,
                             encoder_ffn_embed_dim=3072, encoder_attention_heads=12, activation_fn="gelu", layer_norm_first=False,
                             deep_norm=True, conv_pos=128, conv_pos_groups=16, relative_position_embedding=True, num_buckets=320,
                             max_distance=800, gru_rel_pos=True)
        meta["qformer"] = dict(hidden_size=768, num_attention_heads=12, intermediate_size=3072, num_hidden_layers=2, layer_norm_eps=1e-12,
                               num_query_token=32, encoder_width=768, num_positions=1024)
        meta["mm_audio_projector_type"], meta["mm_audio_hidden_size"] = "qformer_32N_2L", 768
    if "video" in modals:      # LanguageBind_Video_merge: ViT-L/14 at 224 px, 8 frames, temporal attention in every layer
        meta["video"] = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=224,
                             patch_size=14, num_frames=8, add_time_attn=True, layer_norm_eps=1e-5, hidden_act="quick_gelu")
        meta["mm_video_projector_type"], meta["mm_video_hidden_size"], meta["mm_video_select_layer"] = "mlp2x_gelu", 1024, -2
    if "point" in modals:      # pointbert/PointTransformer_8192point_2layer.yaml
        meta["point"] = dict(trans_dim=384, depth=12, num_heads=6, group_size=32, num_group=512, encoder_dims=256, point_dims=6,
                             use_max_pool=False)
        meta["mm_point_projector_type"], meta["mm_point_hidden_size"] = "mlp2x_gelu", 384
    order = ["default"] + [m for m in ("audio", "vision", "video", "point") if m in modals]
    meta["modal_names"] = order
    return meta


def synthetic_state_dict(meta: dict, device="cuda", seed: int = 1234, dtype=None, lora_adapters: Optional[Sequence[str]] = None
                         ) -> Dict[str, torch.Tensor]:
    if dtype is None:                                # the library's storage element at CALL time (bf16, or fp16 under MC_STORAGE_DTYPE=fp16)
        from . import _lib
        dtype = _lib.storage_dtype()
    g = torch.Generator(device=device).manual_seed(seed)
    Hd, I, V, Lyr = meta["hidden_size"], meta["intermediate_size"], meta["vocab_size"], meta["num_hidden_layers"]
    H, Hkv = meta["num_attention_heads"], meta["num_key_value_heads"]
    D = Hd // H
    r = meta["lora_r"]
    names = meta["modal_names"]
    if lora_adapters is None:
        lora_adapters = list(names)
        if meta.get("reset_scaling_weights") and "default-" in meta["reset_scaling_weights"]:
            lora_adapters += [f"default-{m}" for m in names[1:]]

    # MC_SYNTH_THREADS=n (CPU only; the GPU test session sets it): the SAME tensors, drawn in parallel.  torch's CPU generator is one mt19937
    # stream, and randn / rand on a float32 tensor of n >= 16 elements consume exactly one 32-bit draw per element (+ 16 when n % 16 != 0: the
    # vectorised Box-Muller recomputes the last 16), so the main thread only ADVANCES the stream past a tensor (int32 random_(): one draw per
    # element, ~3 ns) after saving the generator state in front of it, and worker threads draw the tensors from their saved states.  Bit-identical
    # to the sequential path (tests/test_host_logic_cpu.py::test_parallel_synthetic_weights_equal_the_sequential_draw); fixtures are unaffected.
    par = 0
    if torch.device(device).type == "cpu":
        import os
        par = int(os.environ.get("MC_SYNTH_THREADS", "0") or 0)
    pending = []                                  # (key placeholder, generator state, kind, shape, scale)
    skip_buf = torch.empty(1 << 22, dtype=torch.int32) if par > 1 else None

    class _Later:                                 # stands in for a tensor until the pool has drawn it
        __slots__ = ("idx",)

        def __init__(self, idx):
            self.idx = idx

    def _defer(kind, shape, scale):
        n = 1
        for d_ in shape:
            n *= d_
        if par <= 1 or n < 4096:
            return None
        st = g.get_state()
        left = n + (16 if (kind == "randn" and n % 16) else 0)
        while left > 0:
            m = min(left, skip_buf.numel())
            skip_buf[:m].random_(generator=g)
            left -= m
        pending.append((st, kind, shape, scale))
        return _Later(len(pending) - 1)

    def nrm(*shape, std=0.02):
        later = _defer("randn", shape, std)
        if later is not None:
            return later
        return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * std).to(dtype)

    def kaiming(out_f, in_f):
        bound = 1.0 / math.sqrt(in_f)           # kaiming_uniform_(a=sqrt(5)) on [out, in]
        later = _defer("rand", (out_f, in_f), bound)
        if later is not None:
            return later
        return ((torch.rand(out_f, in_f, generator=g, device=device, dtype=torch.float32) * 2 - 1) * bound).to(dtype)

    sd = {"model.embed_tokens.weight": nrm(V, Hd), "lm_head.weight": nrm(V, Hd),
          "model.norm.weight": torch.ones(Hd, device=device, dtype=dtype)}
    shapes = {"self_attn.q_proj": (H * D, Hd), "self_attn.k_proj": (Hkv * D, Hd), "self_attn.v_proj": (Hkv * D, Hd),
              "self_attn.o_proj": (Hd, H * D), "mlp.gate_proj": (I, Hd), "mlp.up_proj": (I, Hd), "mlp.down_proj": (Hd, I)}
    for l in range(Lyr):
        p = f"model.layers.{l}"
        sd[f"{p}.input_layernorm.weight"] = torch.ones(Hd, device=device, dtype=dtype)
        sd[f"{p}.post_attention_layernorm.weight"] = torch.ones(Hd, device=device, dtype=dtype)
        for lin, (n, k) in shapes.items():
            sd[f"{p}.{lin}.weight"] = nrm(n, k)
            for ad in lora_adapters:
                sd[f"{p}.{lin}.lora_A.{ad}.weight"] = kaiming(r, k)
                sd[f"{p}.{lin}.lora_B.{ad}.weight"] = nrm(n, r, std=0.01)
    npre = meta.get("local_prefix_tokens", 0)
    if npre:
        for m in names:
            sd[f"prefix_tokens.{m}"] = nrm(1, npre, Hd)
            sd[f"suffix_tokens.{m}"] = nrm(1, meta.get("local_suffix_tokens", npre), Hd)
    if "clip" in meta:
        c = meta["clip"]
        Dm, Im, T = c["hidden_size"], c["intermediate_size"], (c["image_size"] // c["patch_size"]) ** 2 + 1
        pre = "model.modal_encoders.vision.vision_tower.vision_model."
        sd[pre + "embeddings.class_embedding"] = nrm(Dm)
        sd[pre + "embeddings.patch_embedding.weight"] = nrm(Dm, 3, c["patch_size"], c["patch_size"])
        sd[pre + "embeddings.position_embedding.weight"] = nrm(T, Dm)
        for nm in ("pre_layrnorm", "post_layernorm"):
            sd[pre + nm + ".weight"] = torch.ones(Dm, device=device, dtype=dtype)
            sd[pre + nm + ".bias"] = torch.zeros(Dm, device=device, dtype=dtype)
        for i in range(c["num_hidden_layers"]):
            q = f"{pre}encoder.layers.{i}."
            for nm in ("layer_norm1", "layer_norm2"):
                sd[q + nm + ".weight"] = torch.ones(Dm, device=device, dtype=dtype)
                sd[q + nm + ".bias"] = torch.zeros(Dm, device=device, dtype=dtype)
            for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
                sd[q + f"self_attn.{nm}.weight"] = nrm(Dm, Dm)
                sd[q + f"self_attn.{nm}.bias"] = nrm(Dm)
            sd[q + "mlp.fc1.weight"] = nrm(Im, Dm); sd[q + "mlp.fc1.bias"] = nrm(Im)
            sd[q + "mlp.fc2.weight"] = nrm(Dm, Im); sd[q + "mlp.fc2.bias"] = nrm(Dm)
        sd["model.modal_projectors.vision.0.weight"] = nrm(Hd, Dm); sd["model.modal_projectors.vision.0.bias"] = nrm(Hd)
        sd["model.modal_projectors.vision.2.weight"] = nrm(Hd, Hd); sd["model.modal_projectors.vision.2.bias"] = nrm(Hd)
    ones = lambda n: torch.ones(n, device=device, dtype=dtype)
    zeros = lambda n: torch.zeros(n, device=device, dtype=dtype)

    def ln(prefix, n):
        sd[prefix + ".weight"], sd[prefix + ".bias"] = ones(n), zeros(n)

    def lin(prefix, out_f, in_f, bias=True):
        sd[prefix + ".weight"] = nrm(out_f, in_f)
        if bias:
            sd[prefix + ".bias"] = nrm(out_f)

    def mlp2x(modal, in_f):
        lin(f"model.modal_projectors.{modal}.0", Hd, in_f)
        lin(f"model.modal_projectors.{modal}.2", Hd, Hd)

    if "beats" in meta:
        c, pre = meta["beats"], "model.modal_encoders.audio.audio_encoder."
        E, C, Fd, Hh = c["embed_dim"], c["encoder_embed_dim"], c["encoder_ffn_embed_dim"], c["encoder_attention_heads"]
        lin(pre + "post_extract_proj", C, E)
        sd[pre + "patch_embedding.weight"] = nrm(E, 1, c["input_patch_size"], c["input_patch_size"])
        sd[pre + "encoder.pos_conv.0.bias"] = nrm(C)
        sd[pre + "encoder.pos_conv.0.parametrizations.weight.original0"] = torch.ones(1, 1, c["conv_pos"], device=device, dtype=dtype)
        sd[pre + "encoder.pos_conv.0.parametrizations.weight.original1"] = nrm(C, C // c["conv_pos_groups"], c["conv_pos"])
        for i in range(c["encoder_layers"]):
            q = f"{pre}encoder.layers.{i}."
            sd[q + "self_attn.grep_a"] = torch.ones(1, Hh, 1, 1, device=device, dtype=dtype)
            sd[q + "self_attn.relative_attention_bias.weight"] = nrm(c["num_buckets"], Hh, std=0.2)
            for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
                lin(q + "self_attn." + nm, C, C)
            lin(q + "self_attn.grep_linear", 8, C // Hh)
            ln(q + "self_attn_layer_norm", C); ln(q + "final_layer_norm", C)
            lin(q + "fc1", Fd, C); lin(q + "fc2", C, Fd)
        ln(pre + "encoder.layer_norm", C); ln(pre + "layer_norm", E)
        qc, pp = meta["qformer"], "model.modal_projectors.audio."
        Q, QI = qc["hidden_size"], qc["intermediate_size"]
        sd[pp + "audio_query_tokens"] = nrm(1, qc["num_query_token"], Q)
        ln(pp + "audio_Qformer.bert.embeddings.LayerNorm", Q)
        for i in range(qc["num_hidden_layers"]):
            q = f"{pp}audio_Qformer.bert.encoder.layer.{i}."
            for att, kvw in (("attention", Q), ("crossattention", qc["encoder_width"])):
                lin(q + att + ".self.query", Q, Q); lin(q + att + ".self.key", Q, kvw); lin(q + att + ".self.value", Q, kvw)
                lin(q + att + ".output.dense", Q, Q); ln(q + att + ".output.LayerNorm", Q)
            lin(q + "intermediate_query.dense", QI, Q); lin(q + "output_query.dense", Q, QI); ln(q + "output_query.LayerNorm", Q)
        lin(pp + "audio_llama_proj", Hd, Q)
        sd[pp + "audio_position_embedding.weight"] = nrm(qc["num_positions"], qc["encoder_width"])
    if "video" in meta:
        c, pre = meta["video"], "model.modal_encoders.video.video_tower."
        Dm, Im, T = c["hidden_size"], c["intermediate_size"], (c["image_size"] // c["patch_size"]) ** 2 + 1
        sd[pre + "embeddings.class_embedding"] = nrm(Dm)
        sd[pre + "embeddings.patch_embedding.weight"] = nrm(Dm, 3, c["patch_size"], c["patch_size"])
        sd[pre + "embeddings.position_embedding.weight"] = nrm(T, Dm)
        ln(pre + "pre_layrnorm", Dm); ln(pre + "post_layernorm", Dm)
        for i in range(c["num_hidden_layers"]):
            q = f"{pre}encoder.layers.{i}."
            sd[q + "temporal_embedding"] = nrm(1, c["num_frames"], Dm)
            for att in ("self_attn", "temporal_attn"):
                for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
                    lin(q + att + "." + nm, Dm, Dm)
            ln(q + "layer_norm1", Dm); ln(q + "layer_norm2", Dm); ln(q + "temporal_layer_norm1", Dm)
            lin(q + "mlp.fc1", Im, Dm); lin(q + "mlp.fc2", Dm, Im)
        mlp2x("video", Dm)
    if "point" in meta:
        c, pre = meta["point"], "model.modal_encoders.point.point_encoder."
        Tm, Ed = c["trans_dim"], c["encoder_dims"]
        sd[pre + "cls_token"], sd[pre + "cls_pos"] = nrm(1, 1, Tm), nrm(1, 1, Tm)
        for nm, o, i_ in (("first_conv.0", 128, c["point_dims"]), ("first_conv.3", 256, 128), ("second_conv.0", 512, 512), ("second_conv.3", Ed, 512)):
            sd[pre + f"encoder.{nm}.weight"] = nrm(o, i_, 1, std=0.1); sd[pre + f"encoder.{nm}.bias"] = nrm(o)
        for nm, n in (("first_conv.1", 128), ("second_conv.1", 512)):
            ln(pre + "encoder." + nm, n)
            sd[pre + f"encoder.{nm}.running_mean"] = zeros(n); sd[pre + f"encoder.{nm}.running_var"] = ones(n)
        lin(pre + "reduce_dim", Tm, Ed); lin(pre + "pos_embed.0", 128, 3); lin(pre + "pos_embed.2", Tm, 128)
        for i in range(c["depth"]):
            q = f"{pre}blocks.blocks.{i}."
            ln(q + "norm1", Tm); ln(q + "norm2", Tm)
            lin(q + "mlp.fc1", 4 * Tm, Tm); lin(q + "mlp.fc2", Tm, 4 * Tm)
            lin(q + "attn.qkv", 3 * Tm, Tm, bias=False); lin(q + "attn.proj", Tm, Tm)
        ln(pre + "norm", Tm)
        mlp2x("point", Tm)
    if pending:
        from concurrent.futures import ThreadPoolExecutor

        def draw(job):
            st, kind, shape, scale = job
            gi = torch.Generator()
            gi.set_state(st)
            if kind == "randn":
                return (torch.randn(*shape, generator=gi, dtype=torch.float32) * scale).to(dtype)
            return ((torch.rand(*shape, generator=gi, dtype=torch.float32) * 2 - 1) * scale).to(dtype)
        nt = torch.get_num_threads()
        torch.set_num_threads(1)                  # the pool is the parallelism: no OpenMP team per worker
        try:
            with ThreadPoolExecutor(max_workers=par) as pool:
                drawn = list(pool.map(draw, pending))
        finally:
            torch.set_num_threads(nt)
        for k, v in sd.items():
            if isinstance(v, _Later):
                sd[k] = drawn[v.idx]
    return sd


def synthetic_prompt(B: int, modal_sentinels: Sequence[int], vocab: int = 32000, n_before: int = 35, n_after: int = 60, seed: int = 0):
    """[1] + n_before ids + (sentinel, 13)* + n_after ids, ids U{3..vocab-1}, equal length across the batch."""
    g = torch.Generator().manual_seed(seed)
    parts = [torch.ones(B, 1, dtype=torch.long), torch.randint(3, vocab, (B, n_before), generator=g)]
    for s in modal_sentinels:
        parts += [torch.full((B, 1), s, dtype=torch.long), torch.full((B, 1), 13, dtype=torch.long)]
    parts.append(torch.randint(3, vocab, (B, n_after), generator=g))
    return torch.cat(parts, 1)
