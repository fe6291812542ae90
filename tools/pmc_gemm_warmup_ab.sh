#!/bin/bash
# FETCH_SIZE of the four layer GEMMs with and without the next-tile L2 warm-up ("explain why down's fabric traffic rose
# 3.84x -> 4.36x with the L2 warm-up").  Two rocprofv3 passes, kernel trace + one counter each; run through gpurun.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for v in on off; do
  if [ $v = off ]; then export MC_GEMM_DEBUG=12344; else unset MC_GEMM_DEBUG; fi
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcw_$v -o run -- python3 tools/gemm_layer_pmc.py 2 > gpurun_out/pmcw_$v.log 2>&1
  echo "pass $v rc=$?"
  src=$(find gpurun_out/pmcw_$v -name '*counter_collection.csv' | head -1)
  (head -1 "$src"; grep gemm_tile256_kernel "$src") > gpurun_out/pmcw_${v}_FETCH.csv
  rm -rf gpurun_out/pmcw_$v
done
python3 - <<'PY'
import csv, json
out = {}
for v in ("on", "off"):
    rows = list(csv.DictReader(open(f"gpurun_out/pmcw_{v}_FETCH.csv")))
    per = {}
    for r in rows:
        per.setdefault(r["Dispatch_Id"], 0.0)
        per[r["Dispatch_Id"]] += float(r["Counter_Value"])
    vals = [per[k] for k in sorted(per, key=int)]
    # launch order per rep: qkv, o, gate_up, down; FETCH_SIZE in KiB, 128-byte requests tallied at 64 (x2)
    names = ["qkv", "o", "gate_up", "down"]
    out[v] = {n: [round(2 * vals[i] * 1024 / 1e9, 3) for i in range(len(vals)) if i % 4 == j] for j, n in enumerate(names)}
json.dump({"probe": "fetch bytes (GB per launch, 2 x FETCH_SIZE) of the four layer GEMMs with the next-tile L2 warm-up on / off (debug word 12344)", "GB": out},
          open("gpurun_out/r04_warmup_fetch_ab.json", "w"), indent=1)
print(json.dumps(out))
PY
