"""Can an HBM-streaming kernel with few registers share CUs with gemm_tile256_kernel (2 waves x ~232 VGPRs per SIMD, 128 KiB LDS)?
Stream A: back-to-back o_proj-shaped GEMMs; stream B: a low-register streaming kernel (torch's elementwise copy / add: ~16-24 VGPRs,
256-thread workgroups, no LDS).  Times: each alone, then both together."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from modelcompose_amd import ops
M, N, K = 44656, 4096, 4096
x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
w = ops.pack_weight(torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
src = torch.randn(1 << 29, device="cuda", dtype=torch.bfloat16)      # 1 GiB
dst = torch.empty_like(src)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
NG, NC = 20, 40
def gemms():
    with torch.cuda.stream(sa):
        for _ in range(NG): ops.linear_ex(x, w, out=out)
def copies(kind):
    with torch.cuda.stream(sb):
        for _ in range(NC):
            if kind == "copy": dst.copy_(src)
            else: torch.sum(src)
def timed(fa, fb):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if fa: fa()
    if fb: fb()
    torch.cuda.synchronize(); return time.perf_counter() - t0
from modelcompose_amd import _lib
force = int(sys.argv[1]) if len(sys.argv) > 1 else 0          # 1: every large GEMM on the 186-register 192-column tiles (round 4)
_lib.check(_lib.lib().mc_gemm_set_option(b"force_tile192", force), "force_tile192")
print(f"force_tile192 = {force}")
gemms(); copies("copy"); copies("sum"); torch.cuda.synchronize()
tg = timed(gemms, None)
print(f"GEMMs alone: {tg*1e3:.1f} ms ({2*M*N*K*NG/tg/1e12:.0f} TFLOP/s)")
for kind, by in (("copy", 2 * src.numel() * 2), ("sum", src.numel() * 2)):
    tc = timed(None, lambda: copies(kind))
    tb = timed(gemms, lambda: copies(kind))
    print(f"{kind}: alone {tc*1e3:.1f} ms ({by*NC/tc/1e12:.2f} TB/s); together {tb*1e3:.1f} ms (sum of both alone {1e3*(tg+tc):.1f}, max {1e3*max(tg,tc):.1f})")
