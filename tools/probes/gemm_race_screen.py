import os, sys
sys.path.insert(0, os.getcwd())
import torch
from modelcompose_amd import _lib, ops
BF = torch.bfloat16
L = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(3)
bad_total = 0
for (M, N, K, res) in ((16500, 4096, 4096, True), (44656, 12288, 4096, False), (9000, 4096, 11008, True), (30000, 1024, 1024, False)):
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(BF)
    x = torch.randn(M, K, device="cuda", generator=g).to(BF)
    r = torch.randn(M, N, device="cuda", generator=g).to(BF) if res else None
    pw = ops.pack_weight(w)
    L.mc_gemm_debug(2); ref = ops.linear(x, pw, residual=r); L.mc_gemm_debug(0)      # 128x128 kernel
    bad = 0
    n = 600 if M * N * K < 1.5e12 else 150
    for _ in range(n):
        bad += int(not torch.equal(ops.linear(x, pw, residual=r), ref))
    print(f"M={M} N={N} K={K} residual={res}: {bad} of {n} launches differ from the 128x128 kernel's bits", flush=True)
    bad_total += bad
print("RACE SCREEN", "CLEAN" if bad_total == 0 else "FAILED")
