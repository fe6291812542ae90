"""Build-container run of the FULL CPU baseline sample: the oracle (torch fp32 CPU port of the reference path) on one sample of the
metric's workload - 3-way composed Vicuna-7B, all 32 decoder layers, 336 px image + 10 s audio + 8-frame video, 32 greedy tokens.
bench.py times a bounded sample of this (1 and 3 layers, extrapolated); this script records the un-extrapolated number for DESIGN.md."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from modelcompose_amd import synthetic  # noqa: E402
from oracle import pipeline  # noqa: E402


def main():
    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    modals, sentinels = bench.WORKLOADS["iav"][0], bench.WORKLOADS["iav"][1]
    meta = bench.workload_meta("iav", layers)
    t0 = time.perf_counter()
    sd = synthetic.synthetic_state_dict(meta, device="cpu", seed=7, dtype=torch.float32)
    t_gen = time.perf_counter() - t0
    om = pipeline.OracleModel.from_state_dict(sd, meta)
    ids = synthetic.synthetic_prompt(1, sentinels)
    mi = bench.synthetic_inputs(modals, 1, "cpu", 3)
    mi = {k: ({kk: (vv.float() if vv.is_floating_point() else vv) for kk, vv in v.items()} if isinstance(v, dict) else v.float()) for k, v in mi.items()}
    with torch.no_grad():
        t0 = time.perf_counter()
        out = om.generate(ids, mi, max_new_tokens=32, ignore_eos=True)
        dt = time.perf_counter() - t0
    print(json.dumps({"layers": layers, "cores": cores, "seconds_per_sample": round(dt, 1), "samples_per_s": 1.0 / dt, "weight_generation_s": round(t_gen, 1),
                      "new_tokens": int(out.shape[1])}), flush=True)


if __name__ == "__main__":
    main()
