"""Build-container tool: rank candidate row seeds of a full-width case by their smallest top-2 margin over the 17 greedy steps (fp32
branch-form oracle only).  Rows with the largest minimum margin are the ones whose greedy path is least sensitive to bf16 noise.

    python tools/rank_rows_by_margin.py <case> <first seed> <count>"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fullwidth_cases as fc  # noqa: E402
from oracle import pipeline  # noqa: E402


def main():
    name, start, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    meta, sd = fc.build_weights(name)
    sdf = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}
    del sd
    o32 = pipeline.OracleModel.from_state_dict(sdf, meta)
    res = []
    for seed in range(start, start + count):
        t0 = time.time()
        ids, mi = fc.build_rows(name, [seed])
        if "point" in mi:
            meta["fps_start"] = [0]
        with torch.no_grad():
            _, l32 = o32.generate(ids, fc.to_f32(mi), max_new_tokens=fc.N_NEW, ignore_eos=True, return_logits=True)
        m = fc.margins(l32).min().item()
        res.append((m, seed))
        print(f"{name} row seed {seed}: min margin {m:.4f} ({time.time() - t0:.0f}s)", flush=True)
    res.sort(reverse=True)
    print("best:", [(s, round(m, 4)) for m, s in res[:8]])


if __name__ == "__main__":
    main()
