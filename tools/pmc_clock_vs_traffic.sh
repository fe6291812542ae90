#!/bin/bash
# Held clock against fabric traffic for gemm_tile256_kernel: the four routed decoder-layer GEMMs of the headline workload
# (tools/gemm_layer_pmc.py), the SAME FLOPs under tile orders that move different amounts of data through the fabric:
#   shared   the shipped order (32-tile blocks dealt round-robin over the XCDs, n-slabs of 32 tile columns on gate|up)
#   noslab   the same without the n-slabs
#   slab8    n-slabs of 8 tile columns on every launch
#   plain    blockIdx order (raster 0: consecutive tiles of one tile row land on 8 different XCDs)
# per order two rocprofv3 passes (counters only): FETCH_SIZE + WRITE_SIZE -> L2-fill bytes per launch; MFMA busy + GRBM_GUI_ACTIVE -> held clock.
#   tools/pmc_clock_vs_traffic.sh <tag>     ->  gpurun_out/clock_vs_traffic_<tag>.json
set -u
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run_order() {      # name, MC_GEMM_DEBUG, MC_GEMM_OPTIONS
  local name=$1
  export MC_GEMM_DEBUG=$2
  export MC_GEMM_OPTIONS=$3
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/cvt_${tag}_${name}_$c -o run -- python3 tools/gemm_layer_pmc.py 2 > gpurun_out/cvt_${tag}_${name}_$c.log 2>&1
    echo "$name pmc $c rc=$?"
    src=$(find gpurun_out/cvt_${tag}_${name}_$c -name '*counter_collection.csv' | head -1)
    (head -1 "$src"; grep gemm_tile256_kernel "$src") > gpurun_out/cvt_${tag}_${name}_${c}.csv
    rm -rf gpurun_out/cvt_${tag}_${name}_$c
  done
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/cvt_${tag}_${name}_mfma -o run -- python3 tools/gemm_layer_pmc.py 2 > gpurun_out/cvt_${tag}_${name}_mfma.log 2>&1
  echo "$name pmc mfma rc=$?"
  cc=$(find gpurun_out/cvt_${tag}_${name}_mfma -name '*counter_collection.csv' | head -1)
  kt=$(find gpurun_out/cvt_${tag}_${name}_mfma -name '*kernel_trace.csv' | head -1)
  python3 tools/pmc_mfma.py "$cc" "$kt" gpurun_out/cvt_${tag}_${name}_mfma.json ${tag}_${name} > /dev/null
  rm -rf gpurun_out/cvt_${tag}_${name}_mfma
  grep launch_order gpurun_out/cvt_${tag}_${name}_FETCH_SIZE.log > gpurun_out/cvt_${tag}_${name}_launches.json
  python3 tools/pmc_gemm_traffic.py gpurun_out/cvt_${tag}_${name}_FETCH_SIZE.csv gpurun_out/cvt_${tag}_${name}_WRITE_SIZE.csv gpurun_out/cvt_${tag}_${name}_launches.json $tag gpurun_out/cvt_${tag}_${name}_traffic.json > /dev/null
  rm -f gpurun_out/cvt_${tag}_${name}_FETCH_SIZE.csv gpurun_out/cvt_${tag}_${name}_WRITE_SIZE.csv gpurun_out/cvt_${tag}_${name}_*.log
}
run_order shared 0 "raster_shared=1"
run_order noslab 0 "raster_slab=0"
run_order slab8 0 "raster_slab=8"
run_order plain 65536 "raster_shared=1"
unset MC_GEMM_DEBUG MC_GEMM_OPTIONS
python3 - "$tag" <<'EOF'
import json, sys
tag = sys.argv[1]
out = {"tag": tag, "kernel": "gemm_tile256_kernel", "scope": "the four routed decoder-layer GEMMs of the headline workload, standalone (tools/gemm_layer_pmc.py), 2 repetitions",
       "orders": {}}
for name in ("shared", "noslab", "slab8", "plain"):
    try:
        t = json.load(open(f"gpurun_out/cvt_{tag}_{name}_traffic.json"))
        m = json.load(open(f"gpurun_out/cvt_{tag}_{name}_mfma.json"))
    except Exception as e:
        out["orders"][name] = {"error": str(e)}
        continue
    k = [v for kk, v in m.get("kernels", {}).items() if "gemm_tile256" in kk]
    e = k[0] if k else {}
    out["orders"][name] = {"l2_fill_bytes_per_launch": t["gemm_tile256_kernel_bytes_per_launch"], "ratio_to_algorithmic": t["ratio"],
                           "per_gemm_ratio": {g: r["ratio"] for g, r in t["per_gemm"].items()},
                           "held_clock_ghz": e.get("effective_clock_ghz"), "mfma_busy_frac": e.get("mfma_busy_frac"),
                           "tflops": e.get("tflops_from_mops"), "duration_ms": e.get("duration_ms"), "dispatches": e.get("dispatches")}
json.dump(out, open(f"gpurun_out/clock_vs_traffic_{tag}.json", "w"), indent=1)
print(json.dumps(out, indent=1))
EOF
