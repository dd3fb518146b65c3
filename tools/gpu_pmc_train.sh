#!/bin/bash
# PMC counter passes over the training step (tools/bench_train.py --fused): one rocprofv3 run per counter group,
# kernel-trace only, as the pool requires.  Summary -> gpurun_out/<tag>/summary.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-pmc_train}
# PMC_CMD: another command to count instead of the training step (e.g. the closed loop: tools/gpu_round6.sh clpmc)
PMC_CMD=${PMC_CMD:-tools/bench_train.py --fused --steps 6 --warmup 2}
export TMPDIR=/tmp
REPO=$PWD
mkdir -p gpurun_out/$TAG
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$REPO/gpurun_out/$TAG/p$i" -o p -- python $REPO/$PMC_CMD > "$REPO/gpurun_out/$TAG/p$i.log" 2>&1)
  tail -1 gpurun_out/$TAG/p$i.log | cut -c1-120
done
python tools/pmc_summary.py gpurun_out/$TAG | tee gpurun_out/$TAG/summary.txt | cut -c1-400
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete
