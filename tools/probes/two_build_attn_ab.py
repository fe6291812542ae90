"""Two builds of libmc_hip.so in one process, LLM prefill attention (B = 48 and 16, L = 2793, causal): python tools/probes/two_build_attn_ab.py old.so new.so"""
import json, os, random, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modelcompose_amd import _lib, ops
BF = torch.bfloat16
handles = {}
for nm, pth in (("old", sys.argv[1]), ("new", sys.argv[2])):
    _lib.LIB_PATH = os.path.abspath(pth); _lib._lib = None; handles[nm] = _lib.lib()
H, D = 32, 128
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for (B, L) in ((48, 2793), (16, 2793), (4, 3317), (16, 683)):
    S = (L + 32 + 63) // 64 * 64
    q = torch.randn(B, L, H, D, device="cuda").to(BF); k = torch.randn(B, H, S, D, device="cuda").to(BF); v = torch.randn(B, H, S, D, device="cuda").to(BF)
    lens = torch.full((B,), L, dtype=torch.int32, device="cuda")
    res, outs = {}, {}
    for r in range(6):
        order = list(handles); random.Random(r).shuffle(order)
        for nm in order:
            _lib._lib = handles[nm]
            out = torch.empty(B * L, H * D, dtype=BF, device="cuda")
            f = lambda: ops.attn_prefill(q, k, v, out, B, H, H, L, S, D, (L*H*D, H*D, D), (H*S*D, D, S*D), (H*S*D, D, S*D), H*D, True, kv_lens=lens)
            for _ in range(3): f()
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(nm, []).append(e0.elapsed_time(e1) / 10 * 1e-3)
            outs[nm] = out
    fl = 4.0 * L * L * D * H * B / 2
    o, n = statistics.median(res["old"]), statistics.median(res["new"])
    print(json.dumps({"B": B, "L": L, "old_tflops": round(fl / o / 1e12, 1), "new_tflops": round(fl / n / 1e12, 1), "new_vs_old_pct": round((o / n - 1) * 100, 2),
                      "identical": bool(torch.equal(outs["old"], outs["new"]))}), flush=True)
