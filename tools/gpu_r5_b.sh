#!/bin/bash
# round 5: where a closed-loop step's time goes (kernel + copy trace of the batched loop, steps enqueued ahead / policy in loop)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5; export TMPDIR=/tmp
for v in 1,0 1,1; do
CL_ONLY=$v bash tools/gpu_trace_seq.sh cl_$v pack_transforms tools/ab_closed_loop.py > gpurun_out/r5/b_seq_$v.txt 2>&1
cat gpurun_out/r5/b_seq_$v.txt | tail -40
done
