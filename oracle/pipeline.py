"""Oracle (TEST INFRASTRUCTURE): end-to-end composed model on the CPU.

encode (per-modality encoder -> projector -> prefix|feat|suffix)  multimodal_arch.py:197-268
splice                                                              multimodal_arch.py:287-459
LocalLoRA llama prefill + greedy decode                             multimodal_llama.py:676-767 and the
  transformers==4.31 greedy_search loop (third-party; restated: argmax of logits[:, -1], feed
  input_ids[:, -1:] + tuple KV cache, stop on EOS / max_new_tokens, finished rows emit pad).
Also the timed ``cpu_baseline`` of bench.py.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import device_path
from . import encoders as enc
from . import llm, splice


class OracleModel:
    def __init__(self, sd: Dict[str, torch.Tensor], cfg: llm.LLMConfig, meta: dict, emulate: Optional[str] = None,
                 device_opts: Optional[dict] = None):
        self.sd, self.cfg, self.meta, self.emulate = sd, cfg, meta, emulate
        # emulate == "device" only: {"lazy": compose a layer's weights when it runs, "rounding": {point: bool}} (oracle/device_path.py)
        self.device_opts = dict(device_opts or {})
        self.modals = [m for m in cfg.modal_names if m != "default"]
        self.prefix = {k.split(".", 1)[1]: v for k, v in sd.items() if k.startswith("prefix_tokens.")} or None
        self.suffix = {k.split(".", 1)[1]: v for k, v in sd.items() if k.startswith("suffix_tokens.")} or None
        # emulate == "device": the backbone follows the HIP path's rounding points (oracle/device_path.py) instead of the reference's
        # branch form; encoders stay fp32 - pass the device's own feature blocks (feats_blocks) to isolate the backbone
        self._dw = None

    @classmethod
    def from_state_dict(cls, sd, meta, emulate=None, device_opts=None):
        cfg = llm.LLMConfig(
            vocab_size=meta["vocab_size"], hidden_size=meta["hidden_size"], intermediate_size=meta["intermediate_size"],
            num_hidden_layers=meta["num_hidden_layers"], num_attention_heads=meta["num_attention_heads"],
            num_key_value_heads=meta["num_key_value_heads"], max_position_embeddings=meta["max_position_embeddings"],
            rms_norm_eps=meta["rms_norm_eps"], lora_r=meta["lora_r"], lora_alpha=meta["lora_alpha"],
            lora_strategy=meta.get("lora_strategy"), modal_names=tuple(meta["modal_names"]),
            reset_scaling_weights=meta.get("reset_scaling_weights"), pad_token_id=meta.get("pad_token_id", 0),
            eos_token_id=meta.get("eos_token_id", 2))
        return cls(sd, cfg, meta, emulate, device_opts)

    # -- encoders -----------------------------------------------------------
    def _sub(self, prefix):
        n = len(prefix)
        return {k[n:]: v for k, v in self.sd.items() if k.startswith(prefix)}

    def encode_modal(self, modal: str, x):
        if modal == "vision":
            c = enc.ClipVisionConfig(**self.meta["clip"])
            sdv = self._sub("model.modal_encoders.vision.vision_tower.")
            f = enc.clip_vision_tower(x, sdv, c, self.meta.get("mm_vision_select_layer", -2),
                                      self.meta.get("mm_vision_select_feature", "patch"))
            return enc.projector(f, self.sd, "model.modal_projectors.vision", self.meta.get("mm_projector_type", "linear"))
        from . import encoders_extra as ex      # audio / video / point (added with their fixtures)
        return ex.encode(self, modal, x)

    # -- forward ------------------------------------------------------------
    def prepare(self, input_ids, modal_inputs, attention_mask=None, labels=None, feats_blocks=None):
        """feats_blocks: {modal: (n_items, T, hidden)} already projected and wrapped in prefix / suffix tokens (what
        encode_modal_inputs returns); replaces the encoders for the modalities it names."""
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids, dtype=torch.bool)
        if feats_blocks is not None:
            fns = {m: (lambda x, m=m: feats_blocks[m].float()) for m in self.modals}
            feats, fmask = splice.encode_modal_inputs(modal_inputs, self.modals, fns, None, None, skip_absent=True)
        else:
            fns = {m: (lambda x, m=m: self.encode_modal(m, x)) for m in self.modals}
            feats, fmask = splice.encode_modal_inputs(modal_inputs, self.modals, fns, self.prefix, self.suffix, skip_absent=True)
        return splice.prepare_inputs_labels_for_multimodal(input_ids, attention_mask, labels, list(modal_inputs), feats,
                                                           fmask, self.sd["model.embed_tokens.weight"])

    def device_weights(self):
        if self._dw is None:
            rnd = self.device_opts.get("rounding") or {}
            self._dw = device_path.DeviceWeights(self.sd, self.cfg, lazy=bool(self.device_opts.get("lazy")),
                                                 round_weights=rnd.get("weights", True))
        return self._dw

    def prefill(self, input_ids, modal_inputs, attention_mask=None, last_only=False, feats_blocks=None):
        am, emb, _, mam = self.prepare(input_ids, modal_inputs, attention_mask, feats_blocks=feats_blocks)
        if self.cfg.lora_strategy not in ("modal", "modal+language"):          # multimodal_llama.py:703-704
            mam = None
        if self.emulate == "device":
            assert bool(am.all()), "the device-path restatement takes unpadded batches"
            rnd = self.device_opts.get("rounding")
            e0 = emb.float() if rnd and not rnd.get("embed", True) else device_path.bf(emb.float())
            logits, kv = device_path.forward(self.device_weights(), e0, mam, last_only=last_only, rounding=rnd)
            return logits, kv, am
        h, kv = llm.model_forward(self.sd, self.cfg, inputs_embeds=emb, attention_mask=am, modal_attention_mask=mam,
                                  emulate=self.emulate)
        if last_only:
            h = h[:, -1:]
        return llm.lm_logits(h, self.sd), kv, am

    def decode_step(self, token_ids, kv, am, keep_mask=False):
        if keep_mask:
            # modal_inputs is None: prepare_inputs_labels_for_multimodal returns the caller's mask untouched (multimodal_arch.py:290-293), i.e.
            # HF's greedy loop mask = the prompt's mask extended by one attended position per generated token (third-party, restated)
            am = torch.cat([am, torch.ones((am.shape[0], 1), dtype=am.dtype)], 1)
        else:
            am = torch.ones((am.shape[0], kv[-1][-1].shape[-2] + 1), dtype=am.dtype)  # multimodal_arch.py:290-293
        if self.emulate == "device":
            dw = self.device_weights()
            logits, kv = device_path.forward(dw, dw.embed[token_ids][:, None], None, past_kv=kv, last_only=True,
                                             rounding=self.device_opts.get("rounding"))
            return logits[:, -1], kv, am
        h, kv = llm.model_forward(self.sd, self.cfg, input_ids=token_ids[:, None], attention_mask=am, past_key_values=kv,
                                  emulate=self.emulate)
        return llm.lm_logits(h, self.sd)[:, -1], kv, am

    def generate(self, input_ids, modal_inputs, max_new_tokens=128, ignore_eos=False, return_logits=False, feats_blocks=None,
                 attention_mask=None, keep_mask=None, forced_ids=None):
        """Greedy; returns the NEW ids (B, n).  With ignore_eos=False rows that hit EOS emit pad afterwards
        and the loop stops once every row has finished (transformers 4.31 greedy_search).  attention_mask (B, L_text): the prompt's
        padding mask; modal_inputs None (not {}) keeps it in force over the decode steps, as the reference does (keep_mask=None); with
        modal_inputs passed the reference replaces it by all ones on decode steps (multimodal_arch.py:290-293) - keep_mask=True is the
        reference WITHOUT that replacement, which is what the HIP path implements.  forced_ids (B, >= n - 1): TEACHER
        FORCING - decode step s is fed forced_ids[:, s] instead of this model's own argmax (the returned ids stay the argmaxes), so that two
        implementations can be compared step by step on one history."""
        if keep_mask is None:
            keep_mask = modal_inputs is None
        keep_mask = bool(keep_mask) and attention_mask is not None
        logits, kv, am = self.prefill(input_ids, modal_inputs or {}, attention_mask=attention_mask, last_only=True, feats_blocks=feats_blocks)
        last = logits[:, -1]
        B = input_ids.shape[0]
        unfinished = torch.ones(B, dtype=torch.long)
        pad = self.cfg.pad_token_id if self.cfg.pad_token_id is not None else self.cfg.eos_token_id
        out, all_logits = [], []
        for step in range(max_new_tokens):
            all_logits.append(last)
            nxt = last.argmax(-1)
            if not ignore_eos:
                nxt = nxt * unfinished + pad * (1 - unfinished)
                unfinished = unfinished * (nxt != self.cfg.eos_token_id).long()
            out.append(nxt)
            if (not ignore_eos and unfinished.max() == 0) or step == max_new_tokens - 1:
                break
            feed = nxt if forced_ids is None else forced_ids[:, step]
            last, kv, am = self.decode_step(feed, kv, am, keep_mask=keep_mask)
        ids = torch.stack(out, 1)
        if return_logits:
            return ids, torch.stack(all_logits, 1)
        return ids
