"""A/B of the n-slab raster of gemm_tile256_kernel (mc_gemm_set_option("raster_slab", SW)): interleaved rounds, random data, the LLM shapes of
the headline workload; outputs compared bit for bit with the slab-less raster."""
import json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modelcompose_amd import _lib, ops
BF = torch.bfloat16
L = _lib.lib()
shapes = [(44656, 22016, 4096), (44656, 12288, 4096), (44656, 4096, 11008), (44656, 4096, 4096)]
variants = [0, 32, 16, 24, 48]
res = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for (M, N, K) in shapes:
    w = ops.pack_weight((torch.randn(N, K, device="cuda") * K ** -0.5).to(BF))
    x = torch.randn(M, K, device="cuda").to(BF)
    out = torch.empty(M, N, dtype=BF, device="cuda")
    for _ in range(5):
        ops.linear(x, w, out=out)
    ref = None
    ts = {}
    same = True
    for rnd in range(6):
        for sw in variants:
            L.mc_gemm_set_option(b"raster_slab", sw)
            ops.linear(x, w, out=out)
            e0.record()
            for _ in range(4):
                ops.linear(x, w, out=out)
            e1.record(); torch.cuda.synchronize()
            ts.setdefault(sw, []).append(e0.elapsed_time(e1) / 4 * 1e-3)
            if rnd == 0:
                if ref is None: ref = out.clone()
                else: same &= bool(torch.equal(out, ref))
    L.mc_gemm_set_option(b"raster_slab", 0)
    row = {"M": M, "N": N, "K": K, "bit_identical": same}
    for sw in variants:
        row[f"slab{sw}"] = round(2.0 * M * N * K / statistics.median(ts[sw]) / 1e12, 1)
    print(json.dumps(row), flush=True)
    res.append(row)
    del w, x, out
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/gemm_raster_slab_ab.json", "w"), indent=1)
