"""Build-container tool: find seeds for tests/fullwidth_cases.py whose greedy paths are margin-safe.

    python tools/screen_fullwidth_seeds.py configs1_vision <rows wanted> <first row seed>

The case's weights are built once; every candidate row seed is run alone (batch 1) through the fp32 branch-form oracle and the
device-rounding restatement on the CPU.  A row is accepted when both produce the same ids and every step's top-2 margin is
>= fullwidth_cases.MIN_MARGIN of the logit scale; write the accepted seeds into CASES[...]["row_seeds"]."""
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
    name = sys.argv[1]
    want = int(sys.argv[2])
    start = int(sys.argv[3])
    meta, sd = fc.build_weights(name)
    sdf = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}
    del sd
    o32 = pipeline.OracleModel.from_state_dict(sdf, meta)
    od = pipeline.OracleModel.from_state_dict(sdf, meta, emulate="device")
    good = []
    seed = start
    while len(good) < want and seed < start + 200:
        t0 = time.time()
        ids, mi = fc.build_rows(name, [seed])
        if "point" in mi:
            meta["fps_start"] = [0]
        mif = fc.to_f32(mi)
        with torch.no_grad():
            i32, l32 = o32.generate(ids, mif, max_new_tokens=fc.N_NEW, ignore_eos=True, return_logits=True)
            m32 = fc.margins(l32).min().item()
            if m32 < fc.MIN_MARGIN:
                print(f"{name} row seed {seed}: fp32 min margin {m32:.4f} reject ({time.time() - t0:.0f}s)", flush=True)
                seed += 1
                continue
            idv, ldv = od.generate(ids, mif, max_new_tokens=fc.N_NEW, ignore_eos=True, return_logits=True)
        same = bool(torch.equal(i32, idv))
        mdv = fc.margins(ldv).min().item()
        err = ((l32 - ldv).abs().max() / l32.abs().max()).item()
        ok = same and mdv >= fc.MIN_MARGIN
        print(f"{name} row seed {seed}: ids_equal={same} min_margin fp32={m32:.4f} device={mdv:.4f} |fp32-device|/scale={err:.2e} "
              f"{'ACCEPT' if ok else 'reject'} ({time.time() - t0:.0f}s)", flush=True)
        if ok:
            good.append(seed)
        seed += 1
    print("accepted row seeds:", good)


if __name__ == "__main__":
    main()
