"""Cached decoding driven from OUTSIDE the model (VERDICT r2 missing #3: forward(past_key_values=...) / inputs_embeds, the contract HF's
generate loop uses, multimodal_llama.py:676-688, :747-767) and the continuous-batching worker engine built on it (VERDICT r2 missing #1,
SURVEY §8(f)4, serve/model_worker.py:123-194), on the tiny reference fixture model g4 (CLIP tower + projector + 2-layer composed LLM)."""
import json
import threading

import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g4_model():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from modelcompose_amd.model.builder import build_from_state_dict
    a, meta, sd = load_golden("g4_e2e_vision")
    return build_from_state_dict(meta, sd), a, meta, sd


def test_external_greedy_loop_over_forward_with_past_key_values_equals_generate(g4_model):
    """A caller's own loop - forward(use_cache=True), then forward(input_ids[:, -1:], past_key_values=...) per token, as transformers'
    greedy_search drives the reference - produces generate()'s ids and step logits; the cache handle answers the tuple-cache questions
    the reference asks of it (past_key_values[-1][-1].shape[-2], multimodal_arch.py:290-293)."""
    model, a, meta, sd = g4_model
    ids, px = a["input_ids"].cuda(), a["pixels"].cuda()
    n_new = 8
    ref, ref_lg = model.generate(ids, modal_inputs={"vision": px}, max_new_tokens=n_new, ignore_eos=True, return_step_logits=True)
    out = model(input_ids=ids, modal_inputs={"vision": px}, use_cache=True)
    pkv = out.past_key_values
    L = out.logits.shape[1]
    assert pkv is not None and len(pkv) == model.config.num_hidden_layers and pkv[-1][-1].shape[-2] == L and pkv.get_seq_length() == L
    cur = ids
    nxt = out.logits[:, -1].argmax(-1)
    lgs = [out.logits[:, -1]]
    toks = [nxt]
    for step in range(n_new - 1):
        cur = torch.cat([cur, nxt[:, None]], 1)
        inp = model.prepare_inputs_for_generation(cur, past_key_values=pkv, modal_inputs={"vision": px}, use_cache=True)
        assert inp["input_ids"].shape == (ids.shape[0], 1)
        o = model(**inp)
        assert o.logits.shape == (ids.shape[0], 1, model.config.vocab_size) and o.past_key_values is pkv
        assert pkv[-1][-1].shape[-2] == L + step + 1
        lgs.append(o.logits[:, 0])
        nxt = o.logits[:, 0].argmax(-1)
        toks.append(nxt)
    got = torch.stack(toks, 1)
    assert torch.equal(got, ref[:, ids.shape[1]:])
    # the all-rows prefill of forward() and generate()'s last-token tail differ in fp32 summation order only; the decode
    # steps are the same launch sequence on the same cache
    got_lg = torch.stack(lgs, 1)
    scale = ref_lg.abs().max()
    assert ((got_lg[:, 0] - ref_lg[:, 0]).abs().max() / scale).item() < 5e-3
    assert ((got_lg[:, 1:] - ref_lg[:, 1:]).abs().max() / scale).item() < 5e-3


def test_cache_grows_past_its_reserve(g4_model):
    model, a, meta, sd = g4_model
    ids, px = a["input_ids"].cuda(), a["pixels"].cuda()
    ref = model.generate(ids, modal_inputs={"vision": px}, max_new_tokens=12, ignore_eos=True)[:, ids.shape[1]:]
    out = model(input_ids=ids, modal_inputs={"vision": px}, use_cache=True, cache_reserve=2)
    pkv = out.past_key_values
    s0 = pkv.st["Smax"]
    nxt = out.logits[:, -1].argmax(-1)
    toks = [nxt]
    for _ in range(11):
        o = model(input_ids=nxt[:, None], past_key_values=pkv)
        nxt = o.logits[:, 0].argmax(-1)
        toks.append(nxt)
    assert pkv.st["Smax"] > s0 or s0 >= out.logits.shape[1] + 12
    assert torch.equal(torch.stack(toks, 1), ref)


def test_forward_with_inputs_embeds_equals_forward_with_input_ids(g4_model):
    """inputs_embeds entry (multimodal_llama.py:696-710): the spliced embeddings + the per-modality masks that
    prepare_inputs_labels_for_multimodal returns, fed back through forward(inputs_embeds=, modal_attention_mask=), give the logits of the
    input_ids call bit for bit (same rows, same routing, same kernels); without masks every row runs on the default adapter."""
    model, a, meta, sd = g4_model
    ids, px = a["input_ids"].cuda(), a["pixels"].cuda()
    ref = model(input_ids=ids, modal_inputs={"vision": px}).logits
    _, am, _, emb, _, mam = model.prepare_inputs_labels_for_multimodal(ids, torch.ones_like(ids, dtype=torch.bool), None, None, {"vision": px},
                                                                       model.prefix_tokens, model.suffix_tokens)
    got = model(inputs_embeds=emb, attention_mask=am, modal_attention_mask=mam).logits
    assert got.shape == ref.shape and torch.equal(got, ref)
    plain = model(inputs_embeds=emb).logits
    assert plain.shape == ref.shape and not torch.equal(plain, ref)           # vision rows on the default adapter: another function
    with pytest.raises(ValueError):
        model(input_ids=ids, inputs_embeds=emb)


def _solo(model, ids, px, n):
    return model.generate(ids, modal_inputs=({"vision": px} if px is not None else None), max_new_tokens=n, ignore_eos=True)[0, ids.shape[1]:].cpu().tolist()


def test_continuous_batching_rows_equal_solo_generation(g4_model):
    """Five requests of different prompts, lengths and budgets through a 3-row engine: every request's tokens equal generate() of that
    request alone (greedy), although requests join and leave the decode batch at different iterations; finished rows are re-used; a
    request whose budget is 1 token never enters a decode step; a broken request fails alone."""
    from modelcompose_amd.serve import ContinuousBatcher, GenerationRequest
    model, a, meta, sd = g4_model
    ids_all, px_all = a["input_ids"], a["pixels"]
    V = model.config.vocab_size
    g = torch.Generator().manual_seed(3)
    text_only = torch.randint(3, V, (1, 9), generator=g)
    reqs = [GenerationRequest(ids_all[0:1], {"vision": px_all[0:1].cuda()}, max_new_tokens=7),
            GenerationRequest(ids_all[1:2], {"vision": px_all[1:2].cuda()}, max_new_tokens=3),
            GenerationRequest(text_only, None, max_new_tokens=9),
            GenerationRequest(ids_all[0:1], {"vision": px_all[1:2].cuda()}, max_new_tokens=1),
            GenerationRequest(ids_all[1:2], {"vision": px_all[0:1].cuda()}, max_new_tokens=6)]
    model.config.eos_token_id, eos_was = -12345, model.config.eos_token_id      # compare full budgets (random-weight models hit EOS at random)
    try:
        want = [_solo(model, r.input_ids.cuda(), None if r.modal_inputs is None else r.modal_inputs["vision"], r.max_new_tokens) for r in reqs]
        eng = ContinuousBatcher(model, max_batch=3, max_seq_len=256)
        bad = GenerationRequest(ids_all[0:1], {}, max_new_tokens=4)             # sentinel without its modality: plan_splice raises
        for r in reqs[:2] + [bad]:
            eng.submit(r)
        for _ in range(2):
            eng.step()
        for r in reqs[2:]:
            eng.submit(r)
        eng.run_until_idle(max_iters=200)
    finally:
        model.config.eos_token_id = eos_was
    assert all(r.finished.is_set() for r in reqs) and bad.finished.is_set()
    assert isinstance(bad.error, ValueError) and all(r.error is None for r in reqs)
    for r, w in zip(reqs, want):
        assert r.new_ids == w, (r.new_ids, w)
    assert eng.queue_length() == 0 and eng.steps < sum(r.max_new_tokens for r in reqs)      # rows decoded together, not one after another


class _Tok:
    """The slice of a tokenizer the worker uses (oracle/toy_tokenizer.py stands in for LLaMA's in the fixtures; here ids are words)."""
    bos_token_id = 1

    def __call__(self, text):
        class R:
            pass
        r = R()
        r.input_ids = [1] + [3 + (sum(map(ord, w)) % 100) for w in text.split()]          # < vocab_size 128 of the fixture model
        return r

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(f"t{int(i)}" for i in ids)

    def batch_decode(self, ids, skip_special_tokens=True):
        return [self.decode(row) for row in ids.tolist()]


def test_model_worker_generate_stream_chunks(g4_model):
    """generate_stream(params) with the reference's parameter names: NUL-terminated JSON chunks whose text grows token by token from the
    prompt, concurrent requests served by one engine thread, stop string honoured, the status call."""
    from modelcompose_amd.serve import ModelWorker
    model, a, meta, sd = g4_model
    model.config.eos_token_id, eos_was = -12345, model.config.eos_token_id
    worker = ModelWorker(model, _Tok(), image_processor=None, max_batch=2, max_seq_len=256)
    try:
        outs = {}

        def run(name, params):
            outs[name] = [json.loads(c[:-1].decode()) for c in worker.generate_stream_gate(params)]
        px = a["pixels"][0:1]
        p1 = {"prompt": "describe <image> now please", "images": px, "temperature": 0.0, "max_new_tokens": 5, "stop": "</s>"}
        p2 = {"prompt": "a text only question", "temperature": 0.0, "max_new_tokens": 4, "stop": "</s>"}
        ths = [threading.Thread(target=run, args=(n, p), daemon=True) for n, p in (("a", p1), ("b", p2))]
        for t in ths:
            t.start()
        for t in ths:
            t.join(60)
        assert not any(t.is_alive() for t in ths) and worker.engine.dead is None, worker.engine.dead
        assert all(c["error_code"] == 0 for c in outs["a"] + outs["b"])
        assert len(outs["a"]) == 5 and len(outs["b"]) == 4
        for name, p in (("a", p1), ("b", p2)):
            texts = [c["text"] for c in outs[name]]
            assert all(t.startswith(p["prompt"]) for t in texts) and all(len(x) < len(y) for x, y in zip(texts, texts[1:]))
        # a stop string that is the decoded form of the first generated token ends the stream after it
        first_tok = outs["b"][0]["text"][len(p2["prompt"]):]
        stop = [json.loads(c[:-1].decode()) for c in worker.generate_stream_gate(dict(p2, stop=first_tok))]
        assert len(stop) == 1 and stop[0]["text"] == p2["prompt"]
        st = worker.get_status()
        assert st["model_names"] == ["modelcompose-hip"] and st["queue_length"] == 0
        bad = [json.loads(c[:-1].decode()) for c in worker.generate_stream_gate({"prompt": "two <image> <image>", "images": px})]
        assert bad[-1]["error_code"] == 1
    finally:
        worker.engine.shutdown()
        model.config.eos_token_id = eos_was
