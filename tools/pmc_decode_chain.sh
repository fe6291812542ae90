#!/bin/bash
# Fabric-side read traffic of the decode GEMMs (gemm_strip_kernel at 48 rows):  tools/pmc_decode_chain.sh <tag> [rows]
# One rocprofv3 pass, kernel trace + FETCH_SIZE only; bytes per launch = 2 * FETCH_SIZE * 1024 (gfx950 correction, MI355X_MICROARCH.md HBM section).
set -u
tag=${1:-r06}
rows=${2:-48}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcs_${tag} -o run -- python3 tools/decode_chain_pmc.py $rows > gpurun_out/pmcs_${tag}.log 2>&1
echo "pmc decode chain rc=$?"
cc=$(find gpurun_out/pmcs_${tag} -name '*counter_collection.csv' | head -1)
kt=$(find gpurun_out/pmcs_${tag} -name '*kernel_trace.csv' | head -1)
python3 - "$cc" "$kt" gpurun_out/pmcs_${tag}.log gpurun_out/decode_chain_traffic_${tag}.json <<'PY'
import csv, json, sys
cc, kt, log, out = sys.argv[1:5]
order = json.loads([l for l in open(log) if l.startswith("{")][-1])["decode_chain"]
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt)) if "gemm_strip_kernel" in r["Kernel_Name"]}
rows = [(int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])) for r in csv.DictReader(open(cc)) if r["Counter_Name"] == "FETCH_SIZE" and "gemm_strip_kernel" in r["Kernel_Name"]]
rows.sort()
assert len(rows) == len(order["launch_order"]), (len(rows), len(order["launch_order"]))
res = {"rows": order["rows"], "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE; bytes = 2 x FETCH_SIZE x 1024 (gfx950 correction); 8 rotating weights per shape, the first 8 launches of a shape (first touch) dropped", "kernels": {}}
for (d, kn, v), o in zip(rows, order["launch_order"]):
    e = res["kernels"].setdefault(o["name"], {"kernel": kn.split("::")[-1].split("(")[0], "algorithmic_bytes": o["algorithmic_bytes"], "x_bytes": o["x_bytes"], "_b": [], "_t": []})
    e["_b"].append(2 * v * 1024); e["_t"].append(dur.get(str(d), 0) / 1e3)
for e in res["kernels"].values():
    b, t = e.pop("_b")[8:], e.pop("_t")[8:]
    e["launches"] = len(b)
    e["fetch_bytes_per_launch"] = sum(b) / len(b)
    e["ratio_to_algorithmic"] = round(sum(b) / len(b) / e["algorithmic_bytes"], 3)
    e["avg_us_profiled"] = round(sum(t) / len(t), 2)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
PY
rm -rf gpurun_out/pmcs_${tag}
