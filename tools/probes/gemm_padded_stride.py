"""Does a row stride that is not a multiple of 4 KiB help gemm_tile256_kernel?  The X operand's DMA pieces (8 rows x 128 bytes per wave instruction) and the
epilogue's stores (16 rows x 64 bytes) address rows K x 2 / N x 2 bytes apart: with K, N multiples of 2048 every row of an instruction sits on the same
L2 channel.  Same GEMM with x / out as views of buffers whose rows are 64 elements longer."""
import json, os, random, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modelcompose_amd import ops
BF = torch.bfloat16
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for (M, N, K) in ((44656, 4096, 4096), (44656, 12288, 4096), (44656, 4096, 11008), (98688, 4096, 1024)):
    w = ops.pack_weight((torch.randn(N, K, device="cuda") * K ** -0.5).to(BF))
    xs = {"dense": torch.randn(M, K, device="cuda").to(BF)}
    xp = torch.zeros(M, K + 64, dtype=BF, device="cuda"); xp[:, :K].copy_(xs["dense"]); xs["pad"] = xp[:, :K]
    outs = {"dense": torch.empty(M, N, dtype=BF, device="cuda"), "pad": torch.empty(M, N + 64, dtype=BF, device="cuda")[:, :N]}
    res = {}
    combos = [("dense", "dense"), ("pad", "dense"), ("dense", "pad"), ("pad", "pad")]
    for r in range(5):
        order = list(combos); random.Random(r).shuffle(order)
        for (xi, oi) in order:
            for _ in range(3): ops.linear(xs[xi], w, out=outs[oi])
            e0.record()
            for _ in range(6): ops.linear(xs[xi], w, out=outs[oi])
            e1.record(); torch.cuda.synchronize()
            res.setdefault((xi, oi), []).append(e0.elapsed_time(e1) / 6 * 1e-3)
    row = {"M": M, "N": N, "K": K, "same_result": bool(torch.equal(outs["dense"], outs["pad"]))}
    for c in combos: row["x_%s__out_%s" % c] = round(2.0 * M * N * K / statistics.median(res[c]) / 1e12, 1)
    print(json.dumps(row), flush=True)
