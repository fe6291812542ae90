"""Co-residency of a second queue's workgroups with gemm_tile256_kernel (round 4).  See spin_probe.hip.
    python tools/probes/spin_beside_gemm.py          -> JSON: spin launch time alone / beside the 256-register GEMM / beside the 186-register
                                                         (192-column) GEMM, for 24- and 104-register spin kernels, and the GEMMs' own time"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from modelcompose_amd import _lib, ops  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libspin_probe.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, os.path.join(here, "spin_probe.hip")], check=True)
spin = C.CDLL(so)
spin.spin_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]

M, N, K = 44656, 4096, 4096
x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
w = ops.pack_weight(torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
NWG, US, NG = int(sys.argv[1]) if len(sys.argv) > 1 else 10240, 500, 20
buf = torch.zeros(NWG * 4, dtype=torch.int64, device="cuda")
PRIO = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # 1: the GEMM stream gets the high priority, the spin stream the low one
sa, sb = (torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)) if PRIO else (torch.cuda.Stream(), torch.cuda.Stream())


def run(gemm: bool, regs: int):
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ga, gb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t_host = time.perf_counter()
    if gemm:
        with torch.cuda.stream(sa):
            ga.record()
            for _ in range(NG):
                ops.linear_ex(x, w, out=out)
            gb.record()
        time.sleep(0.004)                       # the GEMMs are under way when the spin kernel is enqueued
    with torch.cuda.stream(sb):
        ea.record()
        spin.spin_launch(buf.data_ptr(), NWG, US, regs, sb.cuda_stream)
        eb.record()
    torch.cuda.synchronize()
    total_ms = (time.perf_counter() - t_host) * 1e3
    t = buf.view(NWG, 4).cpu()
    span = (t[:, 1].max() - t[:, 0].min()).item() / 100.0           # us, first start -> last end (100 MHz ticks)
    return {"spin_launch_ms": round(ea.elapsed_time(eb), 3), "spin_span_us": round(span, 1), "both_streams_wall_ms": round(total_ms, 2),
            "gemms_ms": round(ga.elapsed_time(gb), 3) if gemm else None}


res = {"probe": "spin_beside_gemm", "gemm_stream_high_priority": bool(PRIO), "spin": {"workgroups": NWG, "threads": 256, "us": US}, "gemm": {"M": M, "N": N, "K": K, "launches": NG}}
for force in (0, 1):
    _lib.check(_lib.lib().mc_gemm_set_option(b"force_tile192", force), "force_tile192")
    for _ in range(3):
        ops.linear_ex(x, w, out=out)
    torch.cuda.synchronize()
    tag = "gemm_186_regs_192_cols" if force else "gemm_256_regs_256_cols"
    res[tag] = {}
    for regs in (24, 104):
        run(False, regs)
        res[tag][f"spin_{regs}_regs"] = {"alone": run(False, regs), "beside": run(True, regs), "beside_again": run(True, regs)}
    ga, gb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ga.record()
    for _ in range(NG):
        ops.linear_ex(x, w, out=out)
    gb.record()
    torch.cuda.synchronize()
    res[tag]["gemms_alone_ms"] = round(ga.elapsed_time(gb), 3)
_lib.check(_lib.lib().mc_gemm_set_option(b"force_tile192", 0), "force_tile192")
print(json.dumps(res))
