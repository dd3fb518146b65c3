#!/bin/bash
# PMC counter passes (one rocprofv3 run per counter group; kernel-trace only, as the pool requires).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-pmc}; VARIANT=${2:-4}
export TMPDIR=/tmp
REPO=$PWD
mkdir -p gpurun_out/$TAG
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$REPO/gpurun_out/$TAG/p$i" -o p -- python "$REPO/tools/pmc_frames.py" $VARIANT 6 > "$REPO/gpurun_out/$TAG/p$i.log" 2>&1)
  tail -1 gpurun_out/$TAG/p$i.log
done
python tools/pmc_summary.py gpurun_out/$TAG | tee gpurun_out/$TAG/summary.txt
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete
