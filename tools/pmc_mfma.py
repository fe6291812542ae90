"""MFMA utilisation per kernel from one rocprofv3 PMC pass (kernel trace + counters, no other tracing):

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY \
              SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -- python3 bench.py --steps 1 --warmup 0 --new-tokens 2 --no-cpu-baseline --no-profile
    python tools/pmc_mfma.py <counter_collection.csv> <kernel_trace.csv> <out.json> <round tag>

mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs) (GRBM_GUI_ACTIVE is summed over the 8 XCDs);
effective_clock_ghz = GRBM_GUI_ACTIVE / 8 / duration (reads high on dispatches shorter than ~0.3 ms, MI355X_MICROARCH.md "DVFS give-back");
tflops = SQ_INSTS_VALU_MFMA_MOPS_BF16 * 512 / duration (all issued MFMAs incl. masked / padded tiles, profiled pass)."""
import csv
import json
import re
import sys
from collections import defaultdict

KEEP = ("gemm_tile256_kernel", "gemm_tile_kernel", "gemm_tile2_kernel", "attn_prefill_kernel", "attn_prefill32_kernel", "attn_decode_kernel", "gemm_skinny2_kernel",
        "gemm_rows_kernel", "compose_multi_kernel")


def short(name):
    m = re.search(r"(\w+(?:<[^>]*>)?)\(", name)
    return m.group(1) if m else name[:60]


def main():
    cc, kt, out, tag = sys.argv[1:5]
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    ctr = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for r in csv.DictReader(open(cc)):
        k = short(r["Kernel_Name"])
        if not k.startswith(KEEP):
            continue
        ctr[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    res = {}
    for k, c in ctr.items():
        ns = sum(dur[d][0] for d in disp[k] if d in dur)
        e = {"dispatches": len(disp[k]), "duration_ms": round(ns / 1e6, 3), "counters": dict(c)}
        if ns and c.get("GRBM_GUI_ACTIVE"):
            e["effective_clock_ghz"] = round(c["GRBM_GUI_ACTIVE"] / 8 / ns, 3)
            e["mfma_busy_frac"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
        if ns and c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16"):
            e["tflops_from_mops"] = round(c["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512 / ns / 1e3, 1)
        if c.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in c:
            e["wait_any_frac_of_wave_cycles"] = round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4)
        if ns and c.get("GRBM_GUI_ACTIVE") and "SQ_LDS_IDX_ACTIVE" in c:
            # LDS-array cycles per CU-cycle (256 CUs; GRBM_GUI_ACTIVE is summed over the 8 XCDs).  Units as the counters report them: if SQ
            # counts quad-cycles on this part the fraction reads 1/4 - compare with the analytic LDS cycles of the tile
            cu_cycles = c["GRBM_GUI_ACTIVE"] / 8 * 256
            e["lds_idx_active_per_cu_cycle"] = round(c["SQ_LDS_IDX_ACTIVE"] / cu_cycles, 4)
            e["lds_bank_conflict_frac_of_lds_cycles"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c["SQ_LDS_IDX_ACTIVE"], 1.0), 4)
            if c.get("SQ_WAVE_CYCLES"):
                e["wait_inst_lds_frac_of_wave_cycles"] = round(c.get("SQ_WAIT_INST_LDS", 0.0) / c["SQ_WAVE_CYCLES"], 4)
        res[k] = e
    json.dump({"round": tag, "note": __doc__.split("\n\n")[-1].replace("\n", " "), "kernels": res}, open(out, "w"), indent=1)
    for k, e in sorted(res.items(), key=lambda kv: -kv[1]["duration_ms"])[:8]:
        print(k, {x: e.get(x) for x in ("dispatches", "duration_ms", "mfma_busy_frac", "effective_clock_ghz", "tflops_from_mops", "wait_any_frac_of_wave_cycles")})


if __name__ == "__main__":
    main()
