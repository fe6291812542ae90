#!/bin/bash
# PMC evidence for gemm_tile256_kernel at the headline workload's prefill shapes (run through gpurun):  tools/pmc_gemm_layer.sh <tag>
# Three separate rocprofv3 passes (kernel trace + counters only): FETCH_SIZE, WRITE_SIZE, MFMA/wait counters -> gpurun_out/pmc_<tag>_*.csv
set -u
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmcg_${tag}_$c -o run -- python3 tools/gemm_layer_pmc.py 2 > gpurun_out/pmcg_${tag}_$c.log 2>&1
  echo "pmc $c rc=$?"
  src=$(find gpurun_out/pmcg_${tag}_$c -name '*counter_collection.csv' | head -1)
  (head -1 "$src"; grep gemm_tile256_kernel "$src") > gpurun_out/pmcg_${tag}_${c}_gemm_tile256.csv
  rm -rf gpurun_out/pmcg_${tag}_$c
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcg_${tag}_mfma -o run -- python3 tools/gemm_layer_pmc.py 2 > gpurun_out/pmcg_${tag}_mfma.log 2>&1
echo "pmc mfma rc=$?"
cc=$(find gpurun_out/pmcg_${tag}_mfma -name '*counter_collection.csv' | head -1)
kt=$(find gpurun_out/pmcg_${tag}_mfma -name '*kernel_trace.csv' | head -1)
python3 tools/pmc_mfma.py "$cc" "$kt" gpurun_out/mfma_util_gemm_layer_$tag.json $tag
rm -rf gpurun_out/pmcg_${tag}_mfma
# round 4: LDS pass - is the LDS array the bound of the main loop?  (SQ_LDS_IDX_ACTIVE = all LDS-array cycles, SQ_LDS_BANK_CONFLICT = the extra
# cycles of conflicts, SQ_WAIT_INST_LDS = issue stalls on the LDS queue; MI355X_MICROARCH.md "rocprofv3 PMC slots")
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcg_${tag}_lds -o run -- python3 tools/gemm_layer_pmc.py 2 > gpurun_out/pmcg_${tag}_lds.log 2>&1
echo "pmc lds rc=$?"
cc=$(find gpurun_out/pmcg_${tag}_lds -name '*counter_collection.csv' | head -1)
kt=$(find gpurun_out/pmcg_${tag}_lds -name '*kernel_trace.csv' | head -1)
python3 tools/pmc_mfma.py "$cc" "$kt" gpurun_out/lds_util_gemm_layer_$tag.json $tag
rm -rf gpurun_out/pmcg_${tag}_lds
grep launch_order gpurun_out/pmcg_${tag}_FETCH_SIZE.log > gpurun_out/pmcg_${tag}_launches.json
python3 tools/pmc_gemm_traffic.py gpurun_out/pmcg_${tag}_FETCH_SIZE_gemm_tile256.csv gpurun_out/pmcg_${tag}_WRITE_SIZE_gemm_tile256.csv gpurun_out/pmcg_${tag}_launches.json $tag
cp profiles/traffic.json gpurun_out/traffic_$tag.json
