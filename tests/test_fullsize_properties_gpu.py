"""Size-independent properties at the real Vicuna-7B / CLIP-L widths (hidden 4096, 32 heads x 128, FFN 11008, vocab 32000, r = 128;
2 decoder layers to keep the test short) where no oracle run is affordable:
  * decode == prefill: the logits of decode step k (skinny split-K GEMMs, fused RoPE/append attention, device-resident loop) equal
    the last-row logits of a fresh prefill over prompt + the k tokens generated so far (256x256 / 128x128 GEMMs, flash attention);
  * batch invariance: every sample of a batch reproduces the prefill logits it gets when run alone bit for bit, the decode logits
    within bf16 rounding;
  * composition identity: all reset coefficients zero -> the composed weights are the base weights."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from modelcompose_amd import synthetic
    from modelcompose_amd.model.builder import build_from_state_dict
    meta = synthetic.vicuna7b_meta(("vision",), None, layers=2)
    sd = synthetic.synthetic_state_dict(meta, device="cuda", seed=5)
    model = build_from_state_dict(meta, sd)
    B = 3
    ids = synthetic.synthetic_prompt(B, [-200], seed=9).cuda()
    px = torch.randn(B, 3, 336, 336, generator=torch.Generator(device="cuda").manual_seed(1), device="cuda").to(torch.bfloat16)
    return model, meta, ids, px


def test_decode_steps_equal_fresh_prefill(big):
    model, meta, ids, px = big
    n_new = 4
    res, lg = model.generate(ids, modal_inputs={"vision": px}, max_new_tokens=n_new, ignore_eos=True, return_step_logits=True)
    new = res[:, ids.shape[1]:]
    scale = lg.abs().max().item()
    for k in range(1, n_new):
        ids_k = torch.cat([ids, new[:, :k]], dim=1)
        _, lg_k = model.generate(ids_k, modal_inputs={"vision": px}, max_new_tokens=1, ignore_eos=True, return_step_logits=True)
        # same math through two different kernel families; bf16 hidden state through 2 layers: 1 % of the logit scale
        err = (lg[:, k] - lg_k[:, 0]).abs().max().item()
        assert err < 1e-2 * scale, (k, err, scale)
        top2 = lg_k[:, 0].topk(2, dim=-1).values
        safe = (top2[:, 0] - top2[:, 1]) > 2e-2 * scale
        assert torch.equal(lg[:, k].argmax(-1)[safe], lg_k[:, 0].argmax(-1)[safe])


def test_batch_invariance(big):
    """A sample's logits and tokens do not depend on the batch it is generated in - bit for bit, prefill AND decode steps.  Prefill: every
    output element is one fixed-order K reduction of the tile GEMM family, attention is per (batch, head).  Decode (round 6): the strip GEMM
    family's one K order per (N, K) for every M <= 64 and the decode attention's per-sequence chunk schedule (any number of workgroups per
    head: the single-row runs below launch more of them than the batched run).  Also across a different Smax: max_new_tokens = 9 gives
    another cache stride, the first 4 steps must still be the same bits."""
    model, meta, ids, px = big
    res, lg = model.generate(ids, modal_inputs={"vision": px}, max_new_tokens=4, ignore_eos=True, return_step_logits=True)
    for b in range(ids.shape[0]):
        r1, l1 = model.generate(ids[b:b + 1], modal_inputs={"vision": px[b:b + 1]}, max_new_tokens=4, ignore_eos=True, return_step_logits=True)
        assert torch.equal(l1[0], lg[b]), (b, (l1[0].float() - lg[b].float()).abs().max().item())
        assert torch.equal(r1[0], res[b]), b
    r9, l9 = model.generate(ids[:2], modal_inputs={"vision": px[:2]}, max_new_tokens=9, ignore_eos=True, return_step_logits=True)
    assert torch.equal(l9[:, :4], lg[:2]) and torch.equal(r9[:, :ids.shape[1] + 4], res[:2])
    rg = model.generate(ids, modal_inputs={"vision": px}, max_new_tokens=4, ignore_eos=True)              # the replayed decode graph
    assert torch.equal(rg, res)


def test_zero_coefficients_compose_to_the_base_weights():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from modelcompose_amd import ops
    N, K, r = 4096, 11008, 128
    g = torch.Generator(device="cuda").manual_seed(3)
    w = (torch.randn(N, K, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    a = torch.randn(r, K, generator=g, device="cuda").to(torch.bfloat16)
    b = torch.randn(N, r, generator=g, device="cuda").to(torch.bfloat16)
    pw = ops.compose_weight(w, [(a, b, 0.0), (a, b, 0.0)], N, K)
    assert torch.equal(ops.unpack_weight(pw), w)
    # and linearity in the coefficients: compose(c1 + c2) == compose with two terms c1, c2 of the same adapter (fp32 sums, one rounding)
    p1 = ops.unpack_weight(ops.compose_weight(w, [(a, b, 0.75)], N, K))
    p2 = ops.unpack_weight(ops.compose_weight(w, [(a, b, 0.5), (a, b, 0.25)], N, K))
    assert (p1.float() - p2.float()).abs().max().item() <= 2 ** -7 * w.float().abs().max().item()


def test_merge_kernels_at_full_adapter_size():
    """TIES merging and the interference metrics on task vectors of the real rank-128 default-adapter size (327 M elements, no oracle
    run affordable): identities that hold for any size.
      * two identical checkpoints merge to themselves (mean / max) or to twice themselves (sum) on the entries K keeps - bit-exact
        (a power-of-two count: torch's bf16 sum-then-divide is exact only then);
      * K = 100 % with one sign-agreeing pair is the plain mean;
      * metrics: d(x, x) = 0 for L2 / cosine / SSD; L2 scales linearly, cosine and SSD are scale-invariant; SSD(x, -x) = 1."""
    from modelcompose_amd import compose
    d = 327_155_712
    g = torch.Generator(device="cuda").manual_seed(3)
    x = (torch.randn(d, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    flat = torch.stack([x, x])
    m = compose.ties_merge_vectors(flat, 20, "mean")
    kept = m != 0
    assert torch.equal(m[kept], x[kept]) and 0.19 < kept.float().mean().item() < 0.21           # top 20 % survive, untouched
    assert torch.equal(compose.ties_merge_vectors(flat, 20, "max")[kept], x[kept])
    s = compose.ties_merge_vectors(flat, 20, "sum")
    assert torch.equal(s[kept], (x[kept].float() * 2).to(torch.bfloat16))
    del flat, m, s
    xf = x.float()
    y = (torch.randn(d, generator=g, device="cuda") * 0.02)
    same = compose.interference_metrics(torch.stack([xf, xf]), 50)
    assert same["L2"] == 0.0 and abs(same["Cosine"]) < 1e-6 and abs(same["SSD"]) < 1e-9 and abs(same["TSSD"]) < 1e-9
    a = compose.interference_metrics(torch.stack([xf, y]), 50)
    b = compose.interference_metrics(torch.stack([xf * 4, y * 4]), 50)
    assert abs(b["L2"] - 4 * a["L2"]) <= 1e-6 * b["L2"]
    for k in ("Cosine", "SSD", "TSSD"):
        assert abs(a[k] - b[k]) <= 1e-9, k                                                       # powers of two: exact scale invariance
    assert abs(a["Cosine"] - 1.0) < 1e-3 and 0.26 < a["SSD"] < 0.30          # independent Gaussians: E|x+y|/(|x|+|y|) = 0.72
    opp = compose.interference_metrics(torch.stack([xf, -xf]), 50)
    assert abs(opp["SSD"] - 1.0) < 1e-12 and abs(opp["Cosine"] - 2.0) < 1e-6
