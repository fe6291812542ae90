"""Forward-with-saved-activations and backward of the Q-Former projector for the stage-2 finetune step (audio recipe:
scripts/model_composition/train/run_finetune_audio_damc.sh:37-38 trains `qformer_32N_2L` on BEATs features).

Mirrors VideoLlamaAudioQformer.forward (modelcompose/model/multimodal_projector/builder.py:130-155) over BLIP-2's BertLayer
(multimodal_projector/Qformer.py:112-277, 403-475 - text FFN removed, cross-attention in every layer): learned position embedding on
the encoder tokens, learned queries -> LayerNorm -> [self-attention, cross-attention to the encoder tokens, query FFN] x L -> Linear.
Every parameter of the projector is trainable (train_multimodal.py:436-465 unfreezes the modal projectors).  All arithmetic is kernels
of libmc_hip.so: bf16 GEMMs (forward, input gradients through transposed packs, weight gradients by the TN kernel), flash attention
forward with LSE + its backward, LayerNorm / GELU backward, column sums for biases.

Dropout: the reference builds this Q-Former from a default BertConfig - hidden_dropout_prob = attention_probs_dropout_prob
= 0.1 - active in model.train() on the embeddings (Qformer.py:108), the attention probabilities (:259) and the three output dense
layers of a layer (:288 self / cross attention output, :374 query FFN output).  All of them are applied here with counter-based Philox
masks (the step's seed; stream ids below), regenerated in the backward pass: hidden dropouts by mc_dropout_bf16 (out = residual +
dropout(dense(x))), probability dropout inside the flash attention forward / backward kernels (mc_attn_prefill_dropout_bf16)."""

from __future__ import annotations

from typing import Dict, List

import torch

from .. import _lib, ops

BF16 = _lib.storage_dtype()      # the library's 16-bit storage element: bf16, or fp16 with MC_STORAGE_DTYPE=fp16 (_lib.set_storage_dtype)
F32 = torch.float32


class QformerTrainable:
    def __init__(self, step, modal: str, proj, raw: Dict[str, torch.Tensor]):
        """Registers every parameter of `proj` (a HipQformerProjector) in the step's flat master buffer under the reference's names."""
        self.step, self.modal, self.proj = step, modal, proj
        # BertConfig defaults unless the projector's config says otherwise (multimodal_projector/builder.py:118-128 builds a default config)
        self.p_hidden = float(getattr(proj, "hidden_dropout_prob", 0.1))
        self.p_attn = float(getattr(proj, "attention_probs_dropout_prob", 0.1))
        self.modal_idx = list(step.model.modal_names).index(modal) if modal in step.model.modal_names else 0
        self.pre = f"model.modal_projectors.{modal}."
        self.names: List[str] = []
        Dm = proj.hidden
        if Dm % 64 or proj.width % 64 or proj.inter % 64 or (Dm // proj.heads) not in (64, 128):
            raise NotImplementedError("Q-Former backward needs hidden / encoder width / FFN width multiples of 64 and head_dim 64 or 128")

        def reg(suffix):
            k = self.pre + suffix
            if k not in raw:
                raise ValueError(f"state dict lacks {k}")
            step._register(k, raw[k])
            self.names.append(k)
        reg("audio_query_tokens")
        reg("audio_position_embedding.weight")
        for s in ("weight", "bias"):
            reg("audio_Qformer.bert.embeddings.LayerNorm." + s)
        for i in range(proj.nl):
            p = f"audio_Qformer.bert.encoder.layer.{i}."
            for att in ("attention", "crossattention"):
                for lin in ("self.query", "self.key", "self.value", "output.dense"):
                    for s in ("weight", "bias"):
                        reg(f"{p}{att}.{lin}.{s}")
                for s in ("weight", "bias"):
                    reg(f"{p}{att}.output.LayerNorm.{s}")
            for lin in ("intermediate_query.dense", "output_query.dense"):
                for s in ("weight", "bias"):
                    reg(f"{p}{lin}.{s}")
            for s in ("weight", "bias"):
                reg(f"{p}output_query.LayerNorm.{s}")
        for s in ("weight", "bias"):
            reg("audio_llama_proj." + s)

    # ------------------------------------------------------------------ helpers over the step's buffers
    def w16(self, suffix):
        return self.step.view(self.step.P16, self.pre + suffix)

    def g32(self, suffix):
        return self.step.view(self.step.G, self.pre + suffix)

    def _lin(self, x, name):
        return ops.linear(x, ops.pack_weight(self.w16(name + ".weight"), self.w16(name + ".bias")))

    def _lin_bwd(self, x, dy, name, need_dx=True):
        """dW = dy^T x, db = colsum(dy) into the gradient buffer; returns dx = dy W."""
        self.step._wgrad([dy], [x], [self.g32(name + ".weight")])
        ops.colsum(dy, out=self.g32(name + ".bias"))
        return ops.linear(dy, ops.pack_weight_t(self.w16(name + ".weight"))) if need_dx else None

    def _ln(self, x, name):
        return ops.layernorm(x, self.w16(name + ".weight"), self.w16(name + ".bias"), self.proj.eps)

    def _ln_bwd(self, x, dy, name):
        dx, t = ops.layernorm_bwd(x, self.w16(name + ".weight"), dy, self.proj.eps)
        ops.colsum(t, out=self.g32(name + ".weight"))
        ops.colsum(dy, out=self.g32(name + ".bias"))
        return dx

    # dropout sites: Philox stream ids 0x40000000 | modality << 16 | layer << 8 | site (LoRA-input dropout uses ids < 2^16)
    SITE_EMB, SITE_SELF_OUT, SITE_CROSS_OUT, SITE_FFN_OUT, SITE_SELF_PROBS, SITE_CROSS_PROBS = range(6)

    def stream_id(self, layer: int, site: int) -> int:
        return 0x40000000 | (self.modal_idx << 16) | (layer << 8) | site

    def _drop(self, x, layer, site, residual=None):
        """residual + dropout(x) (residual None: dropout(x)); identity when the step runs without dropout (p = 0 / eval)."""
        p = self.p_hidden if self.training else 0.0
        if p <= 0.0:
            return x if residual is None else ops.add(x, residual)
        if residual is None:
            return ops.dropout(x, p, self.step._seed, self.stream_id(layer, site))
        out = residual.clone()
        return ops.dropout(x, p, self.step._seed, self.stream_id(layer, site), out=out, accumulate=True)

    def _lin_drop_res(self, x, name, layer, site, residual):
        """residual + dropout(dense(x)): BertSelfOutput / BertOutput before their LayerNorm (Qformer.py:286-289, :372-375)."""
        p = self.p_hidden if self.training else 0.0
        if p <= 0.0:
            return ops.linear(x, ops.pack_weight(self.w16(name + ".weight"), self.w16(name + ".bias")), residual=residual)
        return self._drop(self._lin(x, name), layer, site, residual=residual)

    def _drop_bwd(self, dy, layer, site):
        p = self.p_hidden if self.training else 0.0
        return dy if p <= 0.0 else ops.dropout(dy, p, self.step._seed, self.stream_id(layer, site))

    def _attn_drop(self, layer, site):
        p = self.p_attn if self.training else 0.0
        return None if p <= 0.0 else (p, self.step._seed, self.stream_id(layer, site))

    training = True            # the finetune step trains; set False to run the projector as in eval (no dropout)

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x (B, T, width) encoder tokens (frozen encoder: no gradient needed) -> (B, nq, llm hidden)."""
        pj, dev = self.proj, self.step.dev
        Dm, H, N = pj.hidden, pj.heads, pj.nq
        d = Dm // H
        B, T, W = x.shape
        sv = self.saved = {"B": B, "T": T}
        xe = x.to(dev, BF16).reshape(B * T, W).contiguous()
        idx = torch.arange(T, dtype=torch.int32, device=dev).repeat(B)
        xe = ops.add_rows(xe, self.w16("audio_position_embedding.weight"), idx)            # builder.py:136-140
        sv["xe"] = xe
        q0 = self.w16("audio_query_tokens").reshape(N, Dm).repeat(B, 1).contiguous()
        sv["q0"] = q0
        h0 = self._ln(q0, "audio_Qformer.bert.embeddings.LayerNorm")
        h = self._drop(h0, 0, self.SITE_EMB)                                                # Qformer.py:107-108
        M = B * N
        for i in range(pj.nl):
            p = f"audio_Qformer.bert.encoder.layer.{i}."
            L = sv[i] = {"h_in": h}
            # self-attention among the queries
            q, k, v = (self._lin(h, p + "attention.self." + n) for n in ("query", "key", "value"))
            a = torch.empty(M, Dm, dtype=BF16, device=dev)
            lse = torch.empty(B * H * N, dtype=F32, device=dev)
            st = (N * Dm, Dm, d)
            ops.attn_prefill_lse(q, k, v, a, lse, B, H, N, N, d, st, st, st, Dm, False, dropout=self._attn_drop(i, self.SITE_SELF_PROBS))
            s1 = self._lin_drop_res(a, p + "attention.output.dense", i, self.SITE_SELF_OUT, h)
            h1 = self._ln(s1, p + "attention.output.LayerNorm")
            L.update(q=q, k=k, v=v, a=a, lse=lse, s1=s1, h1=h1)
            # cross-attention to the encoder tokens
            cq = self._lin(h1, p + "crossattention.self.query")
            ck, cv = (self._lin(xe, p + "crossattention.self." + n) for n in ("key", "value"))
            ca = torch.empty(M, Dm, dtype=BF16, device=dev)
            clse = torch.empty(B * H * N, dtype=F32, device=dev)
            stk = (T * Dm, Dm, d)
            ops.attn_prefill_lse(cq, ck, cv, ca, clse, B, H, N, T, d, st, stk, stk, Dm, False, dropout=self._attn_drop(i, self.SITE_CROSS_PROBS))
            s2 = self._lin_drop_res(ca, p + "crossattention.output.dense", i, self.SITE_CROSS_OUT, h1)
            h2 = self._ln(s2, p + "crossattention.output.LayerNorm")
            L.update(cq=cq, ck=ck, cv=cv, ca=ca, clse=clse, s2=s2, h2=h2)
            # query FFN
            fpre = self._lin(h2, p + "intermediate_query.dense")
            f = ops.act(fpre, "gelu")
            s3 = self._lin_drop_res(f, p + "output_query.dense", i, self.SITE_FFN_OUT, h2)
            h = self._ln(s3, p + "output_query.LayerNorm")
            L.update(fpre=fpre, f=f, s3=s3)
        sv["h_out"] = h
        return self._lin(h, "audio_llama_proj").view(B, N, -1)

    # ------------------------------------------------------------------ backward
    def backward(self, dout: torch.Tensor):
        """dout (B * nq, llm hidden) bf16: gradient of the loss w.r.t. this projector's output rows."""
        pj, dev, sv = self.proj, self.step.dev, self.saved
        Dm, H, N = pj.hidden, pj.heads, pj.nq
        d = Dm // H
        B, T = sv["B"], sv["T"]
        M = B * N
        st, stk = (N * Dm, Dm, d), (T * Dm, Dm, d)
        dh = self._lin_bwd(sv["h_out"], dout.contiguous(), "audio_llama_proj")
        dxe = None                                                             # gradient w.r.t. the position-embedded encoder tokens
        for i in reversed(range(pj.nl)):
            p = f"audio_Qformer.bert.encoder.layer.{i}."
            L = sv[i]
            # h = LN(s3), s3 = fc2(gelu(fc1(h2))) + h2
            ds3 = self._ln_bwd(L["s3"], dh, p + "output_query.LayerNorm")
            df = self._lin_bwd(L["f"], self._drop_bwd(ds3, i, self.SITE_FFN_OUT), p + "output_query.dense")
            dfpre = ops.act(L["fpre"], "gelu", dy=df)
            dh2 = ops.add(self._lin_bwd(L["h2"], dfpre, p + "intermediate_query.dense"), ds3)
            # h2 = LN(s2), s2 = o(ca) + h1
            ds2 = self._ln_bwd(L["s2"], dh2, p + "crossattention.output.LayerNorm")
            dca = self._lin_bwd(L["ca"], self._drop_bwd(ds2, i, self.SITE_CROSS_OUT), p + "crossattention.output.dense")
            dcq = torch.empty(M, Dm, dtype=BF16, device=dev)
            dck, dcv = (torch.empty(B * T, Dm, dtype=BF16, device=dev) for _ in range(2))
            ops.attn_bwd(L["cq"], L["ck"], L["cv"], L["ca"], dca, L["clse"], dcq, dck, dcv, B, H, N, T, d, st, stk, stk, st, st, stk, stk, False,
                         dropout=self._attn_drop(i, self.SITE_CROSS_PROBS))
            dh1 = ops.add(self._lin_bwd(L["h1"], dcq, p + "crossattention.self.query"), ds2)
            for n_, g_ in (("key", dck), ("value", dcv)):
                dx_ = self._lin_bwd(sv["xe"], g_, p + "crossattention.self." + n_)
                dxe = dx_ if dxe is None else ops.add(dxe, dx_)
            # h1 = LN(s1), s1 = o(a) + h_in
            ds1 = self._ln_bwd(L["s1"], dh1, p + "attention.output.LayerNorm")
            da = self._lin_bwd(L["a"], self._drop_bwd(ds1, i, self.SITE_SELF_OUT), p + "attention.output.dense")
            dq, dk, dv = (torch.empty(M, Dm, dtype=BF16, device=dev) for _ in range(3))
            ops.attn_bwd(L["q"], L["k"], L["v"], L["a"], da, L["lse"], dq, dk, dv, B, H, N, N, d, st, st, st, st, st, st, st, False,
                         dropout=self._attn_drop(i, self.SITE_SELF_PROBS))
            dh = ds1
            for n_, g_ in (("query", dq), ("key", dk), ("value", dv)):
                dh = ops.add(dh, self._lin_bwd(L["h_in"], g_, p + "attention.self." + n_))
        dq0 = self._ln_bwd(sv["q0"], self._drop_bwd(dh, 0, self.SITE_EMB), "audio_Qformer.bert.embeddings.LayerNorm")
        # the learned queries are shared by the B samples (.expand): sum over the batch; likewise the position rows
        self.g32("audio_query_tokens").copy_(ops.colsum(dq0.view(B, N * Dm)).view(1, N, Dm))
        gpos = self.g32("audio_position_embedding.weight")
        gpos.zero_()
        gpos[:T].copy_(ops.colsum(dxe.view(B, T * pj.width)).view(T, pj.width))
