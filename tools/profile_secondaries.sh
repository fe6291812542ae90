#!/bin/bash
# rocprofv3 --kernel-trace --stats of the two secondary configurations round 6 added (run through gpurun):  tools/profile_secondaries.sh <tag>
#   iav_b1_128: batch 1, 128 greedy tokens, sequential (the reference's eval geometry);  iav_fp16: the headline workload on fp16 storage
set -u
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_b1 -o run -- python3 bench.py --workload iav --batch 1 --no-pipeline --new-tokens 128 --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/bench_prof_${tag}_b1.log 2>&1
echo "b1_128 rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_fp16 -o run -- python3 bench.py --dtype fp16 --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/bench_prof_${tag}_fp16.log 2>&1
echo "fp16 rc=$?"
rm -f gpurun_out/prof_${tag}_b1/*kernel_trace.csv gpurun_out/prof_${tag}_fp16/*kernel_trace.csv
for k in b1 fp16; do grep '^{' gpurun_out/bench_prof_${tag}_$k.log | tail -1 | cut -c1-200; done
