"""Sampled decoding step on the GPU (csrc/sampling.hip) against the oracle's restatement of the transformers 4.31 warpers
(oracle/sampling.py, pinned to the installed transformers' classes in tests/test_oracle_golden.py)."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

CASES = [(0.2, 50, 1.0), (1.0, 5, 0.9), (0.7, 0, 0.5), (1.3, 50, 0.05), (1.0, 0, 1.0), (0.5, 1, 0.3), (1.0, 0, 0.0), (2.0, 31999, 0.95)]


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _logits(M, V, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, V, generator=g) * 3
    x[0, 10:14] = x[0].topk(5)[0][-1]                           # ties at the top-5 boundary
    return x


@pytest.mark.parametrize("V", [32000, 1000, 4099])
def test_final_probabilities_match_the_warpers(V):
    from modelcompose_amd import ops
    from oracle import sampling
    x = _logits(8, V, 5)
    for (T, k, p) in CASES:
        if k >= V:
            continue
        ref = sampling.probabilities(x, T, k, p)
        _, got = ops.sample_step(x.cuda(), T, k, p, seed=1, want_probs=True)
        got = got.cpu()
        # kept sets identical (the cut is exact integer arithmetic on floor(p * 2^40)); probabilities to fp32 softmax accuracy
        gk, rk = got > 0, ref > 0
        # row 0 has equal logits: when the top-p cut falls inside such a tie group torch drops an arbitrary subset of it, the kernel
        # keeps the group (documented deviation) -> extra tokens only, all with the logit of the smallest kept one
        extra = gk[0] & ~rk[0]
        assert not (rk[0] & ~gk[0]).any() and (x[0][extra] == x[0][rk[0]].min()).all(), (V, T, k, p)
        # the integer masses have 2^-40 resolution: kept tokens whose probability is below it carry no mass (they would be drawn less
        # than once in 10^12 samples); everything else must agree exactly
        diff = gk[1:] != rk[1:]
        assert (ref[1:][diff] < 2.0 ** -39).all() and not (gk[1:] & ~rk[1:]).any(), (V, T, k, p, int(diff.sum()))
        assert (got[1:] - ref[1:]).abs().max().item() < 2e-6, (V, T, k, p)
        assert abs(got.sum(1) - 1).max().item() < 1e-5


def test_draw_is_the_inverse_cdf_of_the_given_uniform_and_ties_at_top_p_cut_are_kept():
    from modelcompose_amd import ops
    from oracle import sampling
    x = _logits(64, 32000, 9)
    g = torch.Generator().manual_seed(10)
    u = torch.rand(64, generator=g)
    u[0], u[1] = 0.0, 0.99999994
    ids, pr = ops.sample_step(x.cuda(), 0.9, 50, 0.8, uniform=u.cuda(), want_probs=True)
    assert torch.equal(ids.cpu(), sampling.pick(pr.cpu(), u))
    # equal probabilities straddling the top-p cut: torch drops an arbitrary subset, the kernel keeps the whole tie group
    t = torch.full((1, 1000), -30.0)
    t[0, :4] = 0.0
    _, pr = ops.sample_step(t.cuda(), 1.0, 0, 0.6, want_probs=True)
    assert torch.allclose(pr[0, :4].cpu(), torch.full((4,), 0.25), atol=1e-6) and pr[0, 4:].sum().item() == 0


def test_rng_stream_is_reproducible_and_distribution_matches():
    from modelcompose_amd import ops
    from oracle import sampling
    x = _logits(1, 1000, 3).repeat(4096, 1).cuda()
    a = ops.sample_step(x, 0.8, 20, 0.95, seed=123, step=0)
    b = ops.sample_step(x, 0.8, 20, 0.95, seed=123, step=0)
    c = ops.sample_step(x, 0.8, 20, 0.95, seed=124, step=0)
    d = ops.sample_step(x, 0.8, 20, 0.95, seed=123, step=1)
    assert torch.equal(a, b) and not torch.equal(a, c) and not torch.equal(a, d)
    draws = torch.cat([ops.sample_step(x, 0.8, 20, 0.95, seed=7, step=s) for s in range(16)]).cpu()      # 65536 draws
    ref = sampling.probabilities(x[:1].cpu(), 0.8, 20, 0.95)[0]
    freq = torch.bincount(draws, minlength=1000).double() / draws.numel()
    assert (freq[ref == 0] == 0).all()
    kept = ref > 0
    # binomial standard error per token: 5 sigma
    se = (ref[kept].double() * (1 - ref[kept].double()) / draws.numel()).sqrt()
    assert ((freq[kept] - ref[kept].double()).abs() <= 5 * se + 1e-4).all()


def test_generate_with_sampling_end_to_end():
    from modelcompose_amd.model.builder import build_from_state_dict
    a, meta, sd = load_golden("g4_e2e_vision")
    model = build_from_state_dict(meta, sd)
    ids, px = a["input_ids"].cuda(), a["pixels"].cuda()
    n = 8
    greedy = model.generate(ids, modal_inputs={"vision": px}, max_new_tokens=n, ignore_eos=True)
    # top_k = 1 leaves only the arg-max token: sampling must reproduce the greedy ids bit-exactly (graph-replayed sampled decode)
    s1 = model.generate(ids, modal_inputs={"vision": px}, do_sample=True, temperature=0.7, top_p=0.9, top_k=1, max_new_tokens=n, ignore_eos=True)
    assert torch.equal(s1, greedy)
    kw = dict(modal_inputs={"vision": px}, do_sample=True, temperature=1.5, top_p=0.95, max_new_tokens=n, ignore_eos=True)
    r1 = model.generate(ids, seed=11, **kw)
    r2 = model.generate(ids, seed=11, **kw)
    r3 = model.generate(ids, seed=12, **kw)
    assert torch.equal(r1, r2) and not torch.equal(r1, r3)
    torch.manual_seed(5)
    q1 = model.generate(ids, **kw)
    torch.manual_seed(5)
    q2 = model.generate(ids, **kw)
    assert torch.equal(q1, q2)
    # greedy afterwards is unaffected by the sampling mode of the previous call
    assert torch.equal(model.generate(ids, modal_inputs={"vision": px}, max_new_tokens=n, ignore_eos=True), greedy)
    with pytest.raises(ValueError):
        model.generate(ids, modal_inputs={"vision": px}, do_sample=True, temperature=0.0, max_new_tokens=2)
    with pytest.raises(NotImplementedError):                  # beam search scores greedily: the two modes do not combine
        model.generate(ids, modal_inputs={"vision": px}, num_beams=3, do_sample=True, temperature=0.7, max_new_tokens=2)


def test_generate_streamer_and_stopping_criteria_hooks():
    """transformers' per-token hooks (serve/model_worker.py:160-185): the streamer sees the prompt, then every new token, then end();
    a stopping criterion that fires after n tokens truncates the output there; neither changes the tokens."""
    from modelcompose_amd.model.builder import build_from_state_dict
    a, meta, sd = load_golden("g4_e2e_vision")
    model = build_from_state_dict(meta, sd)
    ids, px = a["input_ids"][:1].cuda(), a["pixels"][:1].cuda()
    plain = model.generate(ids, modal_inputs={"vision": px}, max_new_tokens=7, ignore_eos=True)

    class Streamer:
        def __init__(self):
            self.items, self.ended = [], False

        def put(self, v):
            assert not v.is_cuda
            self.items.append(v.clone())

        def end(self):
            self.ended = True

    st = Streamer()
    res = model.generate(ids, modal_inputs={"vision": px}, max_new_tokens=7, ignore_eos=True, streamer=st)
    assert torch.equal(res, plain) and st.ended
    assert torch.equal(st.items[0], ids.cpu()) and len(st.items) == 1 + 7
    assert torch.equal(torch.stack([v.view(-1) for v in st.items[1:]], dim=1), plain[:, ids.shape[1]:].cpu())
    calls = []

    def after_three(output_ids, scores, **kw):
        calls.append(output_ids.shape[1])
        return output_ids.shape[1] >= ids.shape[1] + 3

    res = model.generate(ids, modal_inputs={"vision": px}, max_new_tokens=7, ignore_eos=True, stopping_criteria=[after_three])
    assert torch.equal(res, plain[:, :ids.shape[1] + 3]) and calls == [ids.shape[1] + 1, ids.shape[1] + 2, ids.shape[1] + 3]


def test_sampling_edge_cases():
    """top_k >= vocab disables the filter; a tiny temperature collapses the distribution onto the arg-max; top_p = 1 keeps everything;
    invalid parameters raise like the transformers warpers (ValueError through the C ABI's status 1)."""
    from modelcompose_amd import ops
    from oracle import sampling
    x = _logits(4, 1000, 21)
    _, pr = ops.sample_step(x.cuda(), 1.0, 5000, 1.0, want_probs=True)
    assert (pr.cpu() - torch.softmax(x, -1)).abs().max().item() < 2e-6 and (pr > 0).all()
    ids = ops.sample_step(x.cuda(), 1e-4, 0, 1.0, seed=3)
    assert torch.equal(ids.cpu(), x.argmax(-1))
    _, pr = ops.sample_step(x.cuda(), 0.9, 0, 1.0, want_probs=True)
    assert (pr.cpu() - sampling.probabilities(x, 0.9, 0, 1.0)).abs().max().item() < 2e-6
    for bad in (dict(temperature=0.0), dict(temperature=-1.0), dict(top_p=1.5), dict(top_k=-2)):
        with pytest.raises(ValueError):
            ops.sample_step(x.cuda(), **{**dict(temperature=1.0, top_k=0, top_p=1.0), **bad})
