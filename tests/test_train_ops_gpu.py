"""Kernels of the training step (csrc/train.hip, csrc/attention_bwd.hip) against torch autograd in fp32 on the same bf16-valued
inputs.  Gradients are stored in bf16: tolerance 2^-6 of the gradient scale unless stated."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from modelcompose_amd import ops as o
    return o


def rb(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF).cuda()


def close(got, ref, rel=2 ** -6):
    got, ref = got.float().cpu(), ref.float().cpu()
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    assert err <= rel * scale + 1e-12, f"max err {err} vs scale {scale}"


@pytest.mark.parametrize("B,H,L,D,causal", [(2, 2, 70, 64, True), (1, 3, 200, 128, True), (2, 2, 64, 64, False), (1, 2, 130, 128, False)])
def test_attention_backward_matches_autograd(ops, B, H, L, D, causal):
    q, k, v = rb(B, L, H, D, seed=1), rb(B, L, H, D, seed=2), rb(B, L, H, D, seed=3)
    d_o = rb(B, L, H, D, scale=0.5, seed=4)
    o = torch.empty(B * L, H * D, dtype=BF, device="cuda")
    lse = torch.empty(B * H * L, dtype=torch.float32, device="cuda")
    st = (L * H * D, H * D, D)
    ops.attn_prefill_lse(q, k, v, o, lse, B, H, L, L, D, st, st, st, H * D, causal)
    qf, kf, vf = (t.float().transpose(1, 2).requires_grad_(True) for t in (q, k, v))           # [B, H, L, D]
    ref = F.scaled_dot_product_attention(qf, kf, vf, is_causal=causal)
    close(o.view(B, L, H, D).transpose(1, 2), ref)
    s = (qf @ kf.transpose(-1, -2)) / math.sqrt(D)
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(L, L, dtype=torch.bool, device="cuda"), 1), float("-inf"))
    assert torch.allclose(lse.view(B, H, L), torch.logsumexp(s, -1) * 1.4426950408889634, rtol=2e-2, atol=2e-2)
    ref.backward(d_o.float().transpose(1, 2))
    dq, dk, dv = (torch.empty(B, L, H, D, dtype=BF, device="cuda") for _ in range(3))
    ops.attn_bwd(q, k, v, o, d_o, lse, dq, dk, dv, B, H, L, L, D, st, st, st, st, st, st, st, causal)
    close(dq.transpose(1, 2), qf.grad, rel=2 ** -5)
    close(dk.transpose(1, 2), kf.grad, rel=2 ** -5)
    close(dv.transpose(1, 2), vf.grad, rel=2 ** -5)


def test_rmsnorm_swiglu_act_backward(ops):
    M, D, I = 37, 256, 192
    x, g, dy, dres = rb(M, D, seed=5), (1 + 0.1 * torch.randn(D)).to(BF).cuda(), rb(M, D, seed=6), rb(M, D, seed=7)
    xf = x.float().requires_grad_(True)
    y = g.float() * xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)
    y.backward(dy.float())
    close(ops.rmsnorm_bwd(x, g, dy, 1e-5, dres=dres), xf.grad + dres.float())
    close(ops.rmsnorm_bwd(x, g, dy, 1e-5), xf.grad)
    gu, di = rb(M, 2 * I, seed=8), rb(M, I, seed=9)
    guf = gu.float().requires_grad_(True)
    (F.silu(guf[:, :I]) * guf[:, I:]).backward(di.float())
    close(ops.swiglu_bwd(gu, di), guf.grad)
    pre, d2 = rb(M, D, seed=10), rb(M, D, seed=11)
    pf = pre.float().requires_grad_(True)
    F.gelu(pf).backward(d2.float())
    close(ops.act(pre, "gelu"), F.gelu(pre.float()), rel=2 ** -7)
    close(ops.act(pre, "gelu", dy=d2), pf.grad)


def test_ce_loss_and_gradient(ops):
    M, V = 19, 512
    logits = (torch.randn(M, V, generator=torch.Generator().manual_seed(12)) * 3).cuda()
    labels = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(13)).cuda()
    labels[[2, 7]] = -100
    n = int((labels >= 0).sum())
    lf = logits.clone().requires_grad_(True)
    ref = F.cross_entropy(lf, labels, ignore_index=-100)
    ref.backward()
    rows, dl = ops.ce_loss(logits, labels, 1.0 / n)
    assert abs(rows.sum().item() / n - ref.item()) < 1e-4 * abs(ref.item())
    close(dl, lf.grad, rel=2 ** -7)
    assert torch.count_nonzero(dl[[2, 7]]) == 0


def test_transpose_mask_colsum_rope_adamw(ops):
    x = rb(70, 200, seed=14)
    t = ops.transpose(x, Rp=128)
    assert t.shape == (200, 128) and torch.equal(t[:, :70], x.t()) and torch.count_nonzero(t[:, 70:]) == 0
    # strided input view
    big = rb(70, 400, seed=15)
    assert torch.equal(ops.transpose(big[:, 200:]), big[:, 200:].t())
    tt = rb(50, 96, seed=16)
    ra = torch.randint(0, 3, (50,), generator=torch.Generator().manual_seed(17)).int().cuda()
    ref = tt.clone().view(50, 3, 32)
    keep = torch.zeros(50, 3, 1, dtype=torch.bool, device="cuda")
    keep[torch.arange(50), ra.long()] = True
    ref = torch.where(keep, ref, torch.zeros_like(ref)).view(50, 96)
    assert torch.equal(ops.lora_mask_rows(tt.clone(), ra, 32, 3), ref)
    close(ops.colsum(x), x.float().sum(0), rel=1e-5)
    # rope: forward then inverse is the identity up to bf16 rounding; forward equals the rotate-half formula
    H, D, M = 3, 64, 21
    v = rb(M, 2 * H * D, seed=18)
    pos = torch.randint(0, 50, (M,), generator=torch.Generator().manual_seed(19)).int().cuda()
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    ang = torch.outer(torch.arange(64).float(), inv)
    cos, sin = ang.cos().cuda().contiguous(), ang.sin().cuda().contiguous()
    w = ops.rope_inplace(v.clone(), pos, cos, sin, H, D, 1.0)
    xh = v.float()[:, :H * D].view(M, H, D)
    c, s_ = cos[pos.long()][:, None, :], sin[pos.long()][:, None, :]
    x1, x2 = xh[..., :D // 2], xh[..., D // 2:]
    ref = torch.cat([x1 * c - x2 * s_, x2 * c + x1 * s_], -1).reshape(M, H * D)
    close(w[:, :H * D], ref, rel=2 ** -7)
    assert torch.equal(w[:, H * D:], v[:, H * D:])
    back = ops.rope_inplace(w.clone(), pos, cos, sin, H, D, -1.0)
    close(back[:, :H * D], v[:, :H * D], rel=2 ** -6)
    # AdamW against torch.optim.AdamW
    p0 = torch.randn(1000, generator=torch.Generator().manual_seed(20)).cuda()
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([p_ref], lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    p, m, vv, p16 = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0), torch.empty(1000, dtype=BF, device="cuda")
    for step in range(1, 4):
        g = torch.randn(1000, generator=torch.Generator().manual_seed(30 + step)).cuda()
        p_ref.grad = g.clone()
        opt.step()
        ops.adamw(p, g * 4.0, m, vv, p16, 1e-2, 0.9, 0.95, 1e-8, 0.1, step, grad_scale=0.25)
    assert torch.allclose(p, p_ref.detach(), rtol=1e-5, atol=1e-6)
    assert torch.equal(p16, p.to(BF))


@pytest.mark.parametrize("M,P,Q,n", [(2728, 4096, 256, 1), (2728, 512, 4096, 1), (300, 72, 136, 1), (1000, 128, 256, 3), (64, 64, 64, 2), (2728, 256, 4096, 1)])
def test_gemm_tn_weight_gradient_form(M, P, Q, n):
    """out = alpha * a^T b from row-major activations (transposing LDS reads): against fp32 torch on the same bf16 inputs; batched
    problems are strided views of one buffer (the q / k / v slices of dqkv); split launches reduce in fixed order (bitwise repeatable)."""
    from modelcompose_amd import ops
    g = torch.Generator().manual_seed(M + P + Q)
    abuf = (torch.randn(M, n * P, generator=g) * 0.5).to(torch.bfloat16).cuda()
    bbuf = (torch.randn(M, n * Q, generator=g) * 0.5).to(torch.bfloat16).cuda()
    a_list = [abuf[:, i * P:(i + 1) * P] for i in range(n)]
    b_list = [bbuf[:, i * Q:(i + 1) * Q] for i in range(n)]
    outs = [torch.full((P, Q), float("nan"), device="cuda") for _ in range(n)]
    ops.gemm_tn(a_list, b_list, outs, alpha=0.5)
    for a, b, o in zip(a_list, b_list, outs):
        ref = 0.5 * (a.float().t() @ b.float())
        assert (o - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-6      # fp32 accumulation in a different order
    again = [torch.empty(P, Q, device="cuda") for _ in range(n)]
    ops.gemm_tn(a_list, b_list, again, alpha=0.5)
    assert all(torch.equal(x, y) for x, y in zip(outs, again))


def test_pack_weight_from_transposed_source():
    from modelcompose_amd import ops
    g = torch.Generator().manual_seed(3)
    for (K, N) in ((256, 4096), (100, 72), (4096, 256)):
        w_t = torch.randn(K, N, generator=g).to(torch.bfloat16).cuda()
        assert torch.equal(ops.unpack_weight(ops.pack_weight_t(w_t)), w_t.t().contiguous())
        assert torch.equal(ops.pack_weight_t(w_t).data, ops.pack_weight(w_t.t().contiguous()).data)


def test_linear_auto_split_k_matches_unsplit_and_is_deterministic():
    """The opt-in split-K of under-filled launches (LoRA rank projections: few output tiles, long K): same result as the unsplit kernel
    up to fp32 summation order, bitwise repeatable, all epilogue features intact; the default path is untouched."""
    from modelcompose_amd import ops
    g = torch.Generator().manual_seed(17)
    for (M, N, K) in ((2728, 256, 11008), (2728, 768, 4096), (300, 128, 2048)):
        x = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).cuda()
        w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
        res = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
        pw = ops.pack_weight(w)
        base = ops.linear(x, pw, residual=res, alpha=0.5)
        a = ops.linear(x, pw, residual=res, alpha=0.5, auto_split=True)
        b = ops.linear(x, pw, residual=res, alpha=0.5, auto_split=True)
        assert torch.equal(a, b)
        assert (a.float() - base.float()).abs().max().item() <= 2 ** -7 * base.float().abs().max().item()       # one bf16 ulp of the output scale
        ref = 0.5 * (x.float() @ w.float().t()) + res.float()
        assert (a.float() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item()
