#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4; export TMPDIR=/tmp
for q in 4 8 16; do
GPU_MAX_HW_QUEUES=$q timeout 600 python tools/ab_frame.py "render_blocks_per_cu=0" --layouts 1 --in-flight 3,2,4,5,6,8 --rounds 2 --tag hwq$q 2> gpurun_out/r4/inflight.err | grep -v "^{"
done
GPU_MAX_HW_QUEUES=8 timeout 600 python tools/ab_frame.py "render_blocks_per_cu=0" --layouts 1 --in-flight 1,3,4 --rounds 2 --tag hwq8_order 2> gpurun_out/r4/inflight.err | grep -v "^{"
