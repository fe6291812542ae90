#!/bin/bash
# Round-end evidence on the GPU box (run through gpurun from the repo root):  tools/profile_round.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of the default bench.py command           -> gpurun_out/prof_<tag>/run_kernel_stats.csv, bench line
#   2. tools/pmc_gemm_layer.sh <tag>: PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA / wait counters; separate runs, kernel trace only) over the
#      four routed layer GEMMs of the same workload, standalone -> profiles/traffic.json + gpurun_out/mfma_util_gemm_layer_<tag>.json
#      (rocprofv3 --pmc aborts with SIGSEGV inside the profiler on the full img+audio+video bench.py process; `--workload vision` is fine)
# The program itself follows `--` (no env / shell hop): the profiler initialises the GPU before the program starts.
set -u
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o run -- python3 bench.py --no-cpu-baseline --no-secondary > gpurun_out/bench_prof_$tag.log 2>&1
echo "stats rc=$?"
grep '^{' gpurun_out/bench_prof_$tag.log | tail -1 | cut -c1-300
rm -f gpurun_out/prof_$tag/*kernel_trace.csv
timeout 900 bash tools/pmc_gemm_layer.sh $tag
