#!/bin/bash
# PMC evidence for the LLM prefill attention kernel at the headline shape (B 16, L 2793, 32 x 128, causal): two rocprofv3 passes
# (kernel trace + counters only) -> gpurun_out/{mfma,lds}_util_attn_<tag>.json.  Run through gpurun:  tools/pmc_attn.sh <tag>
set -u
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmca_${tag}_mfma -o run -- python3 tools/attn_pmc.py > gpurun_out/pmca_${tag}_mfma.log 2>&1
echo "pmc mfma rc=$?"
cc=$(find gpurun_out/pmca_${tag}_mfma -name '*counter_collection.csv' | head -1)
kt=$(find gpurun_out/pmca_${tag}_mfma -name '*kernel_trace.csv' | head -1)
python3 tools/pmc_mfma.py "$cc" "$kt" gpurun_out/mfma_util_attn_$tag.json $tag
rm -rf gpurun_out/pmca_${tag}_mfma
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmca_${tag}_lds -o run -- python3 tools/attn_pmc.py > gpurun_out/pmca_${tag}_lds.log 2>&1
echo "pmc lds rc=$?"
cc=$(find gpurun_out/pmca_${tag}_lds -name '*counter_collection.csv' | head -1)
kt=$(find gpurun_out/pmca_${tag}_lds -name '*kernel_trace.csv' | head -1)
python3 tools/pmc_mfma.py "$cc" "$kt" gpurun_out/lds_util_attn_$tag.json $tag
rm -rf gpurun_out/pmca_${tag}_lds
