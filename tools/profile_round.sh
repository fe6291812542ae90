#!/bin/bash
# Round-end evidence on the GPU box (run through gpurun from the repo root):  tools/profile_round.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of the default bench.py command            -> gpurun_out/prof_<tag>/run_kernel_stats.csv
#   2. two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, kernel trace only) over one prefill -> gpurun_out/pmc_<tag>_{FETCH,WRITE}_SIZE/
#   3. tools/pmc_traffic.py -> profiles/traffic.json (copied to gpurun_out/ for the merge back)
# The program itself follows `--` (no env / shell hop): the profiler initialises the GPU before the program starts.
set -u
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o run -- python3 bench.py --no-cpu-baseline > gpurun_out/bench_prof_$tag.log 2>&1
grep '^{' gpurun_out/bench_prof_$tag.log | tail -1 | cut -c1-300
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_${tag}_$c -o run -- python3 bench.py --steps 1 --warmup 0 --new-tokens 1 --no-cpu-baseline --no-profile > gpurun_out/pmc_${tag}_$c.log 2>&1
  echo "pmc $c rc=$?"
done
f=$(find gpurun_out/pmc_${tag}_FETCH_SIZE -name '*counter_collection.csv' | head -1)
w=$(find gpurun_out/pmc_${tag}_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python3 tools/pmc_traffic.py "$f" "$w" gpurun_out/bench_prof_$tag.log > gpurun_out/traffic_$tag.log 2>&1
cp profiles/traffic.json gpurun_out/traffic_$tag.json
tail -12 gpurun_out/traffic_$tag.log
# the raw counter CSVs are large: keep only the rows of the dominant kernel for the merge back
for c in FETCH_SIZE WRITE_SIZE; do
  src=$(find gpurun_out/pmc_${tag}_$c -name '*counter_collection.csv' | head -1)
  (head -1 "$src"; grep gemm_tile256_kernel "$src") > gpurun_out/pmc_${tag}_${c}_gemm_tile256.csv
  rm -rf gpurun_out/pmc_${tag}_$c
done
rm -f gpurun_out/prof_$tag/*kernel_trace.csv
# 4. MFMA utilisation: one PMC pass (SQ + GRBM counters; kernel trace only) over one prefill + one decode step -> gpurun_out/mfma_util_<tag>.json
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${tag}_mfma -o run -- python3 bench.py --steps 1 --warmup 0 --new-tokens 2 --no-cpu-baseline --no-profile > gpurun_out/pmc_${tag}_mfma.log 2>&1
echo "pmc mfma rc=$?"
cc=$(find gpurun_out/pmc_${tag}_mfma -name '*counter_collection.csv' | head -1)
kt=$(find gpurun_out/pmc_${tag}_mfma -name '*kernel_trace.csv' | head -1)
python3 tools/pmc_mfma.py "$cc" "$kt" gpurun_out/mfma_util_$tag.json $tag
rm -rf gpurun_out/pmc_${tag}_mfma
