"""Beam search (eval/model_multimodal_qa_loader.py:94-102 forwards --num_beams to model.generate): oracle/beam.py restates transformers
4.31's beam_search + BeamSearchScorer; pinned here against the installed transformers' beam search on a tiny Llama, where the releases agree
(length_penalty = 0: 4.31 normalises a hypothesis by its whole length, later releases by the generated length), and by hand-checkable
properties of the 4.31 rule itself."""
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
out = {}
for case, (seed, vocab, eos) in enumerate([(3, 40, 2), (4, 24, 2), (5, 24, 2), (6, 64, 63), (7, 16, 2)]):
    torch.manual_seed(seed)
    cfg = LlamaConfig(vocab_size=vocab, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
                      max_position_embeddings=64, rms_norm_eps=1e-5, pad_token_id=0, bos_token_id=1, eos_token_id=eos, tie_word_embeddings=False)
    cfg._attn_implementation = "eager"
    m = LlamaForCausalLM(cfg).eval().float()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * 0.6)
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, vocab, (3, 5), generator=g)
    for k in (2, 3):
        with torch.no_grad():
            seq = m.generate(ids, attention_mask=torch.ones_like(ids), num_beams=k, do_sample=False, max_new_tokens=8, length_penalty=0.0,
                             early_stopping=False, num_return_sequences=1, pad_token_id=0, eos_token_id=eos)
        out[f"c{case}_k{k}"] = seq.numpy()
    out[f"c{case}_ids"] = ids.numpy()
    out[f"c{case}_cfg"] = np.array([seed, vocab, eos])
    for n, v in m.state_dict().items():
        out[f"c{case}_sd::" + n] = v.numpy()
np.savez(sys.argv[1], **out)
'''


@pytest.fixture(scope="module")
def hf(tmp_path_factory):
    path = str(tmp_path_factory.mktemp("hfbeam") / "beam.npz")
    r = subprocess.run([sys.executable, "-c", HF_SCRIPT, path], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    z = np.load(path)
    return {k: torch.from_numpy(z[k]) for k in z.files}


def _logits_fn(sd, vocab):
    from oracle import llm
    cfg = llm.LLMConfig(vocab_size=vocab, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
                        max_position_embeddings=64, rms_norm_eps=1e-5, lora_r=4, lora_alpha=8, lora_strategy=None, modal_names=("default",),
                        reset_scaling_weights=None, pad_token_id=0, eos_token_id=2)

    def fn(ids):
        with torch.no_grad():
            h, _ = llm.model_forward(sd, cfg, input_ids=ids, attention_mask=torch.ones_like(ids, dtype=torch.bool))
            return llm.lm_logits(h[:, -1:], sd)[:, 0]
    return fn


def test_restated_beam_search_equals_installed_transformers(hf):
    from oracle import beam
    n_eos_closed = 0
    for case in range(5):
        seed, vocab, eos = (int(v) for v in hf[f"c{case}_cfg"])
        sd = {k.split("::", 1)[1]: v for k, v in hf.items() if k.startswith(f"c{case}_sd::")}
        fn = _logits_fn(sd, vocab)
        ids = hf[f"c{case}_ids"]
        for k in (2, 3):
            want = hf[f"c{case}_k{k}"]
            got = beam.beam_search(fn, ids, k, 8, eos, 0, length_penalty=0.0, early_stopping=False)
            assert got.shape == want.shape, (case, k, got, want)
            for b in range(ids.shape[0]):
                # the hypothesis itself: the prompt + the new tokens up to and including the first EOS.  (What follows differs by release: 4.31
                # fills with pad_token_id - BeamSearchScorer.finalize - the installed one repeats the EOS.)
                def cut(row):
                    row = row.tolist()
                    gen = row[5:]
                    return row[:5] + (gen[:gen.index(eos) + 1] if eos in gen else gen)
                assert cut(got[b]) == cut(want[b]), (case, k, b, got[b], want[b])
                tail = got[b].tolist()[len(cut(got[b])):]
                assert all(t == 0 for t in tail)                                  # 4.31: padded with pad_token_id
            n_eos_closed += int((want[:, 5:] == eos).any())
    assert n_eos_closed >= 2, "the pin must include hypotheses closed by EOS"


def test_beam_search_rules_of_4_31():
    """Hand-checkable: a 'model' whose next-token distribution depends on the last token only."""
    from oracle import beam
    V = 6
    T = torch.full((V, V), -20.0)
    # from token 3: 4 (p ~ .6) or 5 (p ~ .4); from 4: always 3 ; from 5: EOS (2) with p ~ .9, else 3
    T[3, 4], T[3, 5] = 0.6, 0.2
    T[4, 3] = 5.0
    T[5, 2], T[5, 3] = 3.0, 0.8
    T[2, 2] = 5.0
    fn = lambda ids: T[ids[:, -1]]
    ids = torch.tensor([[1, 3]])
    # one beam = greedy: 3 -> 4 -> 3 -> 4 ...
    g1 = beam.beam_search(fn, ids, 1, 4, 2, 0)
    assert g1[0].tolist()[:6] == [1, 3, 4, 3, 4, 3]
    # two beams, no length normalisation: sums of log-probs decide, and a hypothesis closed by EOS gets its EOS back
    out = beam.beam_search(fn, ids, 2, 4, 2, 0, length_penalty=0.0)
    lp = torch.log_softmax(T, -1)
    greedy = lp[3, 4] + lp[4, 3] + lp[3, 4] + lp[4, 3]
    closed = lp[3, 5] + lp[5, 2]
    want = [1, 3, 5, 2] if closed > greedy else [1, 3, 4, 3, 4, 3]
    assert out[0].tolist()[:len(want)] == want
    # length_penalty 1 (the default): a hypothesis scores sum_logprobs / len(ids incl. the prompt) - the 4.31 rule
    out1 = beam.beam_search(fn, ids, 2, 4, 2, 0, length_penalty=1.0)
    s_closed, s_greedy = closed / 3, greedy / 6            # [1, 3, 5] closed without its EOS (3 ids); the running beam at max length (6 ids)
    want1 = [1, 3, 5, 2] if s_closed > s_greedy else [1, 3, 4, 3, 4, 3]
    assert out1[0].tolist()[:len(want1)] == want1


def test_done_heuristic_counts_the_new_token():
    """BeamSearchScorer.process of 4.31 hands is_done `cur_len = input_ids.shape[-1] + 1` ("add up to the length which the next_scores is
    calculated on"); with length_penalty != 0 and early_stopping False the other reading (len(ids) alone) stops a step earlier and can return
    another hypothesis.  The rule on one hypothesis set, then a run where the two readings part (gaussian 7-token 'model', seed 23)."""
    from oracle import beam
    from modelcompose_amd.beam import BeamHypotheses
    h, o = BeamHypotheses(1, 1.0, False), beam._Hyps(1, 1.0, False)
    ids3 = torch.tensor([1, 3, 5])
    h.add(ids3, -1.0); o.add(ids3, -1.0)                              # kept score: -1 / 3
    for hy in (h, o):
        assert hy.is_done(-1.0, 3) and not hy.is_done(-1.0, 4)        # -1/3 >= -1/3, but a 4-token continuation could still score -1/4
    T = torch.randn(7, 7, generator=torch.Generator().manual_seed(23)) * 2.5
    fn = lambda ids: T[ids[:, -1]]
    ids = torch.tensor([[1, 3]])
    with_new = beam.beam_search(fn, ids, 2, 6, 2, 0, length_penalty=1.0)
    without = beam.beam_search(fn, ids, 2, 6, 2, 0, length_penalty=1.0, _done_len_offset=0)
    assert with_new.tolist() == [[1, 3, 4, 1, 4, 1, 4, 1]] and without.tolist() == [[1, 3, 4, 1, 2]]
