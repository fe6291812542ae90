"""profiles/traffic.json from the standalone layer-GEMM PMC passes (tools/pmc_gemm_layer.sh):

    python tools/pmc_gemm_traffic.py <fetch csv> <write csv> <launches json> <tag> [out json, default profiles/traffic.json]

bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per dispatch of gemm_tile256_kernel: on gfx950 FETCH_SIZE tallies the 128-byte
requests of wide coalesced reads at 64 bytes (MI355X_MICROARCH.md, HBM section) and both counters are in KiB.  The counters sit on the
fabric side of the L2, so Infinity-Cache hits are included: L2-fill traffic, an upper bound on HBM bytes."""
import csv
import datetime
import json
import sys


def vals(path, counter):
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "gemm_tile256_kernel" in r["Kernel_Name"]]


def main():
    fcsv, wcsv, lj, tag = sys.argv[1:5]
    info = json.loads(open(lj).read())
    L = info["launch_order"]
    f, w = vals(fcsv, "FETCH_SIZE"), vals(wcsv, "WRITE_SIZE")
    per = {}
    for i, (a, b) in enumerate(zip(f, w)):
        per.setdefault(L[i % len(L)]["gemm"], []).append((2 * a + b) * 1024)
    rows = {}
    tot_t = tot_a = 0.0
    for l in L:
        t = sum(per[l["gemm"]]) / len(per[l["gemm"]])
        rows[l["gemm"]] = {"M": l["M"], "N": l["N"], "K": l["K"], "traffic_bytes": t, "algorithmic_bytes": l["algorithmic_bytes"],
                           "ratio": round(t / l["algorithmic_bytes"], 2)}
        tot_t += t
        tot_a += l["algorithmic_bytes"]
    out = {"round": int("".join(ch for ch in tag[1:3] if ch.isdigit()) or 0), "tag": tag, "date": datetime.date.today().isoformat(), "workload": "iav", "per_gpu_batch": info.get("per_gpu_batch"),
           "kernel": "gemm_tile256_kernel",
           "gemm_tile256_kernel_bytes_per_launch": tot_t / len(L), "algorithmic_bytes_per_launch": tot_a / len(L), "ratio": round(tot_t / tot_a, 2),
           "per_gemm": rows,
           "scope": "the four routed decoder-layer GEMMs of the headline workload (same kernel, shapes, adapter groups and epilogues as "
                    "mc_llm_prefill launches them), measured standalone by tools/pmc_gemm_layer.sh: rocprofv3 --pmc aborts (SIGSEGV inside the "
                    "profiler, ROCm 7.2) on the full img+audio+video bench.py process; encoder launches of the same kernel are not included",
           "method": __doc__.split("\n\n")[-1].replace("\n", " ")}
    json.dump(out, open(sys.argv[5] if len(sys.argv) > 5 else "profiles/traffic.json", "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k not in ("method", "scope")}, indent=1))


if __name__ == "__main__":
    main()
