#!/bin/bash
# HBM traffic of the decode attention (back the 0.79-of-8-TB/s figure with counters):  tools/pmc_decode_attn.sh <tag>
# One rocprofv3 pass, kernel trace + FETCH_SIZE only.  bytes per launch = 2 * FETCH_SIZE * 1024 (gfx950 tallies the 128-byte requests of wide
# coalesced reads at 64 bytes: MI355X_MICROARCH.md, HBM section); duration from the same pass's kernel trace.
set -u
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcd_${tag} -o run -- python3 tools/decode_attn_pmc.py > gpurun_out/pmcd_${tag}.log 2>&1
echo "pmc decode attention rc=$?"
cc=$(find gpurun_out/pmcd_${tag} -name '*counter_collection.csv' | head -1)
kt=$(find gpurun_out/pmcd_${tag} -name '*kernel_trace.csv' | head -1)
python3 - "$cc" "$kt" gpurun_out/pmcd_${tag}.log gpurun_out/decode_attn_traffic_${tag}.json <<'PY'
import csv, json, sys
cc, kt, log, out = sys.argv[1:5]
alg = json.loads([l for l in open(log) if l.startswith("{")][-1])["decode_attention"]
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt)) if "attn_decode_kernel" in r["Kernel_Name"]}
rows = [(r["Dispatch_Id"], float(r["Counter_Value"])) for r in csv.DictReader(open(cc)) if r["Counter_Name"] == "FETCH_SIZE" and "attn_decode_kernel" in r["Kernel_Name"]]
rows = rows[8:]                                   # the first sweep over the 8 caches also pays their first touch
by = [2 * v * 1024 for _, v in rows]
us = [dur[d] / 1e3 for d, _ in rows if d in dur]
res = {"kernel": "attn_decode_kernel<128> (RoPE + append fused)", "shape": alg, "launches": len(by), "fetch_bytes_per_launch": sum(by) / len(by),
       "ratio_to_algorithmic": round(sum(by) / len(by) / alg["algorithmic_bytes"], 3), "avg_us_profiled": round(sum(us) / len(us), 1),
       "GBs_algorithmic_profiled": round(alg["algorithmic_bytes"] / (sum(us) / len(us)) / 1e3, 1),
       "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE; bytes = 2 x FETCH_SIZE x 1024 (gfx950 correction); 8 rotating KV caches of 2.3 GB"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
PY
rm -rf gpurun_out/pmcd_${tag}
