"""Oracle (TEST INFRASTRUCTURE): one stage-2 finetune step on the CPU — loss and gradients of the trainable set.

Restates the training forward of the reference, which is the same graph as inference with labels
(modelcompose/model/language_model/multimodal_llama.py:676-745: splice -> LocalLoRA model in branch form with per-token
mask-sum routing (:262-268, :335-336, :380-390) -> lm_head -> shifted CrossEntropyLoss :722-733), differentiated by torch
autograd.  Trainable set = train_multimodal.py:436-465 with lora_strategy='modal+language': every lora_A / lora_B, the modal
projectors, prefix / suffix tokens; everything else (base weights, norms, embeddings, lm_head, encoders) is frozen.
lora_dropout (nn.Dropout on the LoRA input, multimodal_llama.py:133): `dropout_masks` = {linear prefix: keep / (1 - p) as a float
tensor broadcastable to the LoRA input (B, L, K)} applies explicit masks (the HIP step's Philox masks, oracle/philox.py), so the
step stays deterministic and comparable."""
from __future__ import annotations

from typing import Dict

import torch

from . import llm, pipeline


def trainable_keys(sd: Dict[str, torch.Tensor]):
    return [k for k in sd if (".lora_A." in k or ".lora_B." in k or k.startswith("model.modal_projectors.")
                              or k.startswith("prefix_tokens.") or k.startswith("suffix_tokens.")) and sd[k].is_floating_point()]


def loss_and_grads(sd: Dict[str, torch.Tensor], meta: dict, input_ids, labels, modal_inputs, attention_mask=None, dropout_masks=None):
    """Returns (loss, logits, {param name: grad}) — parameters that do not reach the loss get no entry (autograd None)."""
    sd = {k: (v.clone().float() if v.is_floating_point() else v) for k, v in sd.items()}
    keys = trainable_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    om = pipeline.OracleModel.from_state_dict(sd, meta)
    am, emb, new_labels, mam = om.prepare(input_ids, modal_inputs, attention_mask, labels)
    if om.cfg.lora_strategy not in ("modal", "modal+language"):
        mam = None
    if dropout_masks is not None:
        llm.DROPOUT_FN = lambda prefix, adapter, x: x * dropout_masks[prefix].to(x.dtype).view(x.shape)
    try:
        h, _ = llm.model_forward(sd, om.cfg, inputs_embeds=emb, attention_mask=am, modal_attention_mask=mam)
    finally:
        llm.DROPOUT_FN = None
    logits = llm.lm_logits(h, sd)
    loss = llm.cross_entropy_shifted(logits, new_labels, om.cfg.vocab_size)
    grads = torch.autograd.grad(loss, [sd[k] for k in keys], allow_unused=True)
    return loss.detach(), logits.detach(), {k: g for k, g in zip(keys, grads) if g is not None}
