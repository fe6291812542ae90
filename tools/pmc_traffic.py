"""profiles/traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only) over
`bench.py --steps 1 --warmup 0 --new-tokens 1` (one full prefill: every gemm_tile256_kernel launch of a step).

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <bench json line file>

bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 averaged over the kernel's dispatches: on gfx950 FETCH_SIZE tallies the
128-byte requests of wide coalesced reads at 64 bytes (MI355X_MICROARCH.md, HBM section) and both counters are in KiB.  The
counters sit on the fabric side of the L2, so Infinity-Cache hits are included: an upper bound on HBM bytes."""
import csv
import json
import sys

KERNEL = "gemm_tile256_kernel"


def avg(path, counter):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            n += 1
    return tot / max(n, 1), n


def main():
    fetch, nf = avg(sys.argv[1], "FETCH_SIZE")
    write, nw = avg(sys.argv[2], "WRITE_SIZE")
    line = None
    for l in open(sys.argv[3]):
        if l.startswith("{") and "roofline" in l:
            line = json.loads(l)
    import datetime
    cfg = (line or {}).get("config", {})
    out = {"round": 2, "date": datetime.date.today().isoformat(), "workload": cfg.get("workload_name"), "per_gpu_batch": cfg.get("per_gpu_batch"),
           "kernel": KERNEL, "dispatches_fetch_pass": nf, "dispatches_write_pass": nw,
           "gemm_tile256_kernel_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
           "fetch_size_kib_avg": fetch, "write_size_kib_avg": write,
           "algorithmic_bytes_per_launch": line["roofline"].get("avg_algorithmic_bytes_per_launch") if line else None,
           "method": __doc__.split("\n\n")[-1].replace("\n", " ")}
    json.dump(out, open("profiles/traffic.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
