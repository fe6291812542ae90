#!/bin/bash
# Out-of-bounds screen on the GPU box (run through gpurun):  tools/oob_screen.sh [pytest args]
# PYTORCH_NO_HIP_MEMORY_CACHING=1 gives every tensor its own hipMalloc, so a kernel that reads or writes past the end of a tensor by more than the
# page slack faults instead of landing in a neighbour of the caching allocator's segment (how round 6 found an embedding gather fed ids beyond
# the table: a layout-dependent fault late in the full suite).  GPU AddressSanitizer is not available on this pool; this is the cheap substitute.
# ulimit -c 0: a GPU core dump of a 288-GB device fills the disk.
set -u
cd "$GRAFT_REPO_ROOT"
ulimit -c 0
if [ $# -eq 0 ]; then
  set -- tests/test_ops_gpu.py tests/test_e2e_gpu.py tests/test_encoders_gpu.py tests/test_serve_gpu.py tests/test_sampling_gpu.py tests/test_train_ops_gpu.py \
         tests/test_train_step_gpu.py tests/test_merge_gpu.py tests/test_fullsize_properties_gpu.py tests/test_fullwidth_parity_gpu.py
fi
PYTORCH_NO_HIP_MEMORY_CACHING=1 python -m pytest "$@" -q -m gpu -x 2>&1 | tail -15
