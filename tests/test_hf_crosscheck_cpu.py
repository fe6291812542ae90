"""The cross-check SURVEY §8(c) names as the strongest pin obtainable for the third-party arithmetic the reference takes from
transformers==4.31 (absent here): {shimmed reference, installed transformers' LlamaForCausalLM, the oracle's restatement} must agree on the
same weights.  RoPE, the causal + padding mask, RMSNorm and the decoder block did not change between 4.31 and the installed release.

  * oracle/llm.py (restatement of multimodal_llama.py:210-342, :363-468, :488-619 + the 4.31 helpers) vs installed LlamaForCausalLM: prefill
    logits with a full and with a LEFT-PADDED attention mask (positions ignore padding in both: multimodal_llama.py:526-531), and a cached
    decode step;
  * oracle/refshim.py's restated 4.31 rotary embedding / apply_rotary_pos_emb vs the installed ones;
  * when /root/reference is present (build container): the UNMODIFIED reference MultimodalLlamaForCausalLM, imported through the shim
    (its _prepare_decoder_attention_mask and rotary classes are the shim's), loaded with the same weights and LoRA B = 0, vs both.

The installed model runs in a child interpreter: refshim.install() replaces names inside transformers.models.llama.modeling_llama, so the
two cannot share a process with a pristine transformers."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HF_SCRIPT = r'''
import sys, numpy as np, torch
from transformers import LlamaConfig, LlamaForCausalLM
import transformers.models.llama.modeling_llama as ml
torch.manual_seed(5)
cfg = LlamaConfig(vocab_size=128, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                  max_position_embeddings=64, rms_norm_eps=1e-5, rope_theta=10000.0, pad_token_id=0, bos_token_id=1, eos_token_id=2,
                  attention_bias=False, mlp_bias=False, tie_word_embeddings=False)
cfg._attn_implementation = "eager"
m = LlamaForCausalLM(cfg).eval().float()
with torch.no_grad():
    for p in m.parameters():
        p.copy_(torch.randn_like(p) * 0.08)
    for n, p in m.named_parameters():
        if n.endswith("layernorm.weight") or n.endswith("norm.weight"):
            p.copy_(1.0 + 0.1 * torch.randn_like(p))
g = torch.Generator().manual_seed(9)
ids = torch.randint(3, 128, (2, 10), generator=g)
full = torch.ones(2, 10, dtype=torch.long)
left = full.clone(); left[0, :3] = 0; left[1, :1] = 0
out = {}
with torch.no_grad():
    o = m(input_ids=ids, attention_mask=full, use_cache=True)
    out["logits_full"] = o.logits
    nxt = o.logits[:, -1].argmax(-1)
    o2 = m(input_ids=nxt[:, None], attention_mask=torch.ones(2, 11, dtype=torch.long), past_key_values=o.past_key_values, use_cache=True)
    out["next_ids"] = nxt
    out["logits_step"] = o2.logits[:, -1]
    out["logits_left"] = m(input_ids=ids, attention_mask=left).logits
    # the pieces: rotary embedding on random q / k, RMSNorm
    q = torch.randn(2, 4, 10, 16, generator=g); k = torch.randn(2, 4, 10, 16, generator=g)
    pos = torch.arange(10)[None].expand(2, 10)
    cos, sin = m.model.rotary_emb(q, pos)
    qe, ke = ml.apply_rotary_pos_emb(q, k, cos, sin)
    out.update(rope_q=q, rope_k=k, rope_qe=qe, rope_ke=ke)
    x = torch.randn(3, 64, generator=g)
    out.update(norm_x=x, norm_y=m.model.norm(x))
arrs = {k: v.numpy() for k, v in out.items()}
arrs.update(input_ids=ids.numpy(), mask_left=left.numpy())
for k, v in m.state_dict().items():
    arrs["sd::" + k] = v.numpy()
np.savez(sys.argv[1], **arrs)
'''


@pytest.fixture(scope="module")
def hf(tmp_path_factory):
    path = str(tmp_path_factory.mktemp("hf") / "hf.npz")
    r = subprocess.run([sys.executable, "-c", HF_SCRIPT, path], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    z = np.load(path)
    arr = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("sd::")}
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    return arr, sd


def _cfg():
    from oracle import llm
    return llm.LLMConfig(vocab_size=128, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                         max_position_embeddings=64, rms_norm_eps=1e-5, lora_r=4, lora_alpha=8, lora_strategy=None, modal_names=("default",),
                         reset_scaling_weights=None, pad_token_id=0, eos_token_id=2)


def _close(a, b, tol=2e-5):
    scale = b.abs().max().item()
    err = (a - b).abs().max().item() / scale
    assert err < tol, err
    return err


def test_oracle_llama_equals_installed_transformers(hf):
    from oracle import llm
    arr, sd = hf
    cfg = _cfg()
    ids = arr["input_ids"]
    with torch.no_grad():
        h, kv = llm.model_forward(sd, cfg, input_ids=ids, attention_mask=torch.ones(2, 10, dtype=torch.bool))
        _close(llm.lm_logits(h, sd), arr["logits_full"])
        # cached decode step (tuple cache of the reference: torch.cat growth, multimodal_llama.py:284-289)
        h2, _ = llm.model_forward(sd, cfg, input_ids=arr["next_ids"][:, None], attention_mask=torch.ones(2, 11, dtype=torch.bool), past_key_values=kv)
        _close(llm.lm_logits(h2, sd)[:, -1], arr["logits_step"])
        # left padding: additive finfo.min mask, position ids ignore it (multimodal_llama.py:526-531, :543-545).  Compared on attended
        # positions (a fully masked query row is don't-care: the two libraries disagree on what garbage it holds)
        left = arr["mask_left"].bool()
        hl, _ = llm.model_forward(sd, cfg, input_ids=ids, attention_mask=left)
        got, ref = llm.lm_logits(hl, sd), arr["logits_left"]
        _close(got[left], ref[left])
        assert (ref[left] - arr["logits_full"][left]).abs().max() > 1e-3, "the padded run must differ from the unpadded one"


def test_shim_rotary_and_rmsnorm_equal_installed_transformers(hf):
    from oracle import llm, refshim
    arr, sd = hf
    rot = refshim._LlamaRotaryEmbedding431(16, max_position_embeddings=64, base=10000)
    q, k = arr["rope_q"], arr["rope_k"]
    cos, sin = rot(q, seq_len=10)
    pos = torch.arange(10)[None].expand(2, 10)
    qe, ke = refshim._apply_rotary_pos_emb431(q, k, cos, sin, pos)
    _close(qe, arr["rope_qe"], 1e-6)
    _close(ke, arr["rope_ke"], 1e-6)
    # the oracle's own tables / rotation (oracle/llm.py) against the same
    c2, s2 = llm.rope_tables(16, 64, 10000.0)
    qo, ko = llm.apply_rope(q, k, c2, s2, pos)
    _close(qo, arr["rope_qe"], 1e-6)
    _close(ko, arr["rope_ke"], 1e-6)
    _close(llm.rms_norm(arr["norm_x"], sd["model.norm.weight"], 1e-5), arr["norm_y"], 1e-6)
    # the shim's 4.31 mask: 0 where a query may look, finfo.min elsewhere (causal AND key padding) - the semantics the installed eager
    # attention implements (checked end to end above and below)
    am = arr["mask_left"].bool()
    m4 = refshim._prepare_decoder_attention_mask(None, am, (2, 10), torch.zeros(2, 10, 4), 0)
    allowed = torch.ones(10, 10, dtype=torch.bool).tril()[None, None] & am[:, None, None, :]
    assert torch.equal(m4 == 0, allowed) and bool((m4[~allowed] <= torch.finfo(torch.float32).min).all())
    assert torch.equal(llm.decoder_attention_mask(am, 2, 10, 0) == 0, allowed)


REF_SCRIPT = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[2])
from oracle import gen_golden, llm, refshim
z = np.load(sys.argv[1])
arr = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("sd::")}
sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
ml = refshim.import_ref("modelcompose.model.language_model.multimodal_llama")
torch.manual_seed(0)
cfg = gen_golden.tiny_llm_config(ml, modal=("vision",), reset="default-vision=0.5", layers=2, hidden=64, heads=4, inter=128, vocab=128)
cfg.max_position_embeddings = 64
model = ml.MultimodalLlamaForCausalLM(cfg).eval()
missing, unexpected = model.load_state_dict(sd, strict=False)
assert not unexpected and all("lora_" in k or "prefix_tokens" in k or "suffix_tokens" in k for k in missing), (missing, unexpected)
with torch.no_grad():
    for n, p in model.named_parameters():
        if ".lora_B." in n:
            p.zero_()        # untrained LoRA (peft 0.4.0 initialises B = 0; the installed transformers' post_init re-draws every nn.Linear)
def close(a, b, tol=2e-5):
    err = (a - b).abs().max().item() / b.abs().max().item()
    assert err < tol, err
ids = arr["input_ids"]
left = arr["mask_left"].bool()
with torch.no_grad():
    o = model(input_ids=ids, attention_mask=torch.ones(2, 10, dtype=torch.bool), use_cache=True)
    close(o.logits, arr["logits_full"])
    o2 = model(input_ids=arr["next_ids"][:, None], attention_mask=torch.ones(2, 11, dtype=torch.bool), past_key_values=o.past_key_values, use_cache=True)
    close(o2.logits[:, -1], arr["logits_step"])
    ol = model(input_ids=ids, attention_mask=left)
    close(ol.logits[left], arr["logits_left"][left])
    # and the restatement against the shimmed reference on EVERY position, masked rows included (same mask arithmetic on both sides)
    cfg_o = llm.LLMConfig(vocab_size=128, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                          max_position_embeddings=64, rms_norm_eps=1e-5, lora_r=4, lora_alpha=8, lora_strategy=None, modal_names=("default",),
                          reset_scaling_weights=None, pad_token_id=0, eos_token_id=2)
    hl, _ = llm.model_forward(sd, cfg_o, input_ids=ids, attention_mask=left)
    close(llm.lm_logits(hl, sd), ol.logits)
print("ok")
'''


@pytest.mark.skipif(not os.path.isdir("/root/reference/modelcompose"), reason="the reference sources exist in the build container only")
def test_shimmed_reference_equals_installed_transformers_and_the_oracle(hf, tmp_path):
    """The unmodified reference model class, made importable by the shim, with the SAME base weights and untrained LoRA (B = 0).  In a child
    interpreter: the shim replaces sys.modules['modelcompose'] and names inside transformers - neither may leak into the other tests."""
    arr, sd = hf
    path = str(tmp_path / "hf_in.npz")
    np.savez(path, **{k: v.numpy() for k, v in arr.items()}, **{"sd::" + k: v.numpy() for k, v in sd.items()})
    r = subprocess.run([sys.executable, "-c", REF_SCRIPT, path, ROOT], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-3000:]
