"""A/B of two BUILDS of libmc_hip.so in one process (interleaved, random order per round): python tools/probes/two_build_ab.py old.so new.so
Each library is loaded through its own handle (ctypes loads with RTLD_LOCAL: two copies of every symbol and of the library's static state);
modelcompose_amd._lib's cached handle is swapped between launches.  Shapes: the LLM GEMMs of the headline workload."""
import json, os, random, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modelcompose_amd import _lib, ops
BF = torch.bfloat16
paths = {"old": os.path.abspath(sys.argv[1]), "new": os.path.abspath(sys.argv[2])}
handles = {}
for nm, pth in paths.items():
    _lib.LIB_PATH = pth
    _lib._lib = None
    handles[nm] = _lib.lib()
shapes = [(8192, 8192, 8192), (44656, 12288, 4096), (44656, 4096, 4096), (44656, 22016, 4096), (44656, 4096, 11008), (98688, 3072, 1024), (98688, 4096, 1024), (10928, 4096, 4096)]
res, outs = {}, {}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
bufs = {}
_lib._lib = handles["new"]
for shp in shapes:
    M, N, K = shp
    bufs[shp] = (ops.pack_weight((torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)), torch.randn(M, K, device="cuda").to(BF), torch.empty(M, N, dtype=BF, device="cuda"))
for r in range(6):
    for shp in shapes:
        w, x, out = bufs[shp]
        order = list(handles); random.Random(31 * r + len(res)).shuffle(order)
        for nm in order:
            _lib._lib = handles[nm]
            for _ in range(3): ops.linear(x, w, out=out)
            e0.record()
            for _ in range(6): ops.linear(x, w, out=out)
            e1.record(); torch.cuda.synchronize()
            res.setdefault((shp, nm), []).append(e0.elapsed_time(e1) / 6 * 1e-3)
            outs[(shp, nm)] = out.clone()
for shp in shapes:
    M, N, K = shp
    o, n = statistics.median(res[(shp, "old")]), statistics.median(res[(shp, "new")])
    print(json.dumps({"M": M, "N": N, "K": K, "old_tflops": round(2.0 * M * N * K / o / 1e12, 1), "new_tflops": round(2.0 * M * N * K / n / 1e12, 1),
                      "new_vs_old_pct": round((o / n - 1) * 100, 2), "identical": bool(torch.equal(outs[(shp, "old")], outs[(shp, "new")]))}), flush=True)
