"""compose_tile_kernel (round 6, MC_COMPOSE_TILE unset) against compose_multi_kernel (round 5, MC_COMPOSE_TILE=0): run this file once under
each setting - one kernel choice per process - and compare the SHA-256 digests of every output (they must be equal: same MFMA chains, same
term order, one rounding) and the retention ratios; timings of the model's shapes as batched launches.
    python tools/compose_ab.py tile > a.json; MC_COMPOSE_TILE=0 python tools/compose_ab.py general > b.json; python tools/compose_ab.py --compare a.json b.json"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--compare":
    a, b = json.load(open(sys.argv[2])), json.load(open(sys.argv[3]))
    bad = 0
    for k in a["digests"]:
        if a["digests"][k] != b["digests"].get(k):
            bad += 1
            print("DIFFERENT:", k)
    for k in a["retention"]:
        if abs(a["retention"][k] - b["retention"][k]) > 1e-5 * max(1.0, abs(a["retention"][k])):
            bad += 1
            print("RETENTION:", k, a["retention"][k], b["retention"][k])
    print(json.dumps({"cases": len(a["digests"]), "different": bad, "timing_us": {k: (a["timing_us"].get(k), b["timing_us"].get(k)) for k in a["timing_us"]}}, indent=1))
    sys.exit(1 if bad else 0)
import torch
from modelcompose_amd import ops
from modelcompose_amd.model.multimodal_llama import _compose_flush, _compose_multi_into
BF = ops.BF16
tag = sys.argv[1] if len(sys.argv) > 1 else "run"
rep = {"tag": tag, "digests": {}, "retention": {}, "timing_us": {}}
sha = lambda t: hashlib.sha256(t.contiguous().view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:24]
for (N, K, r, stride, off, nterms, masks, name) in (
        (4096, 4096, 128, 1, 0, 6, [7, 8, 16, 32, 0], "o_proj"), (4096, 11008, 128, 1, 0, 6, [7, 8, 16, 32], "down"),
        (11008, 4096, 128, 2, 1, 6, [7, 8, 16, 32], "up_interleaved"), (352, 1024, 64, 1, 0, 3, [7, 1, 2], "small_r64"),
        (1024, 1024, 256, 1, 0, 2, [3, 2], "r256_two_chunks"), (4096, 4096, 128, 1, 0, 0, [0], "copy"), (1000, 4096, 128, 1, 0, 2, [1, 3], "ragged_n"),
        (100, 200, 32, 1, 0, 2, [3], "general_only")):
    g = torch.Generator(device="cuda").manual_seed(N + K + r)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(BF)
    terms = [((torch.randn(r, K, device="cuda", generator=g) * K ** -0.5).to(BF), (torch.randn(N, r, device="cuda", generator=g) * 0.01).to(BF), 0.5 + 0.25 * i)
             for i in range(nterms)]
    cs = (1.0 + 0.1 * torch.randn(K, device="cuda", generator=g)).float()
    for use_cs in (False, True):
        outs = [torch.zeros(ops.packed_elems(N, K) * stride, dtype=BF, device="cuda") for _ in masks]
        rets = [[] for _ in masks]
        _compose_multi_into(w, terms, masks, N, K, outs, col_scale=cs if use_cs else None, nb_stride=stride, nb_offset=off, retentions=rets)
        torch.cuda.synchronize()
        for oi, o in enumerate(outs):
            rep["digests"][f"{name}_cs{int(use_cs)}_out{oi}"] = sha(o)
            if rets[oi]:
                p_ = rets[oi][0].double().sum(0).cpu()
                rep["retention"][f"{name}_cs{int(use_cs)}_out{oi}"] = float(p_[0] / p_[1])
if os.environ.get("MC_COMPOSE_AB_QUICK") == "1":           # tests/test_ops_gpu.py: digests only
    print(json.dumps(rep))
    sys.exit(0)
# timings: one decoder layer's seven linears of the 3-way composed model (4 outputs, 6 terms) as ONE batched call
Hd, I, r = 4096, 11008, 128
g = torch.Generator(device="cuda").manual_seed(3)
def mk(N, K):
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(BF)
    terms = [((torch.randn(r, K, device="cuda", generator=g) * K ** -0.5).to(BF), (torch.randn(N, r, device="cuda", generator=g) * 0.01).to(BF), 0.5) for _ in range(6)]
    return w, terms
shapes = [(Hd, Hd)] * 4 + [(I, Hd), (I, Hd), (Hd, I)]
lin = [mk(N, K) for N, K in shapes]
masks = [7, 8, 16, 32]
outs = [[torch.empty(ops.packed_elems(N, K), dtype=BF, device="cuda") for _ in masks] for N, K in shapes]
byts = sum(2.0 * N * K * (1 + len(masks)) for N, K in shapes)
for label, batched in (("layer_batched", True), ("layer_per_linear", False)):
    ts = []
    for it in range(6):
        batch = [] if batched else None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev = []
        if not batched:
            e0.record()
        for (w, terms), (N, K), o in zip(lin, shapes, outs):
            _compose_multi_into(w, terms, masks, N, K, o, batch=batch)
        if batched:
            _compose_flush(batch, ev)
            torch.cuda.synchronize()
            ts.append(ev[0][0].elapsed_time(ev[0][1]) * 1e3)
        else:
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
    us = sorted(ts[1:])[len(ts[1:]) // 2]
    rep["timing_us"][label] = {"us": round(us, 1), "GBs": round(byts / us / 1e3, 1), "frac_of_8TBs": round(byts / us / 1e3 / 8000, 4)}
for name, masks_ in (("copy_1out_0terms", [0]), ("4out_6terms", [7, 8, 16, 32])):
    w, terms = lin[0]
    o = [torch.empty(ops.packed_elems(Hd, Hd), dtype=BF, device="cuda") for _ in masks_]
    ev = []
    for _ in range(8):
        _compose_multi_into(w, terms, masks_, Hd, Hd, o, events=ev)
    torch.cuda.synchronize()
    us = sorted(a.elapsed_time(b) * 1e3 for a, b in ev[2:])[3]
    rep["timing_us"][name] = {"us": round(us, 1), "GBs": round(2.0 * Hd * Hd * (1 + len(masks_)) / us / 1e3, 1)}
print(json.dumps(rep))
