#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5
( time timeout 1200 python -m pytest tests/test_closed_loop_gpu.py tests/test_batch_gpu.py tests/test_layout_gpu.py -q -m gpu -x ) > $OUT/g_tests.log 2>&1; echo "tests rc=$?" >> $OUT/g_tests.log
grep -E "passed|failed|FAILED|Error|rc=|real" $OUT/g_tests.log | tail
timeout 400 python tools/ab_closed_loop.py > $OUT/g_cl.jsonl 2> $OUT/g_cl.err; cat $OUT/g_cl.jsonl; tail -3 $OUT/g_cl.err
CL_ONLY=1,0 bash tools/gpu_trace_seq.sh cl_g pack_transforms tools/ab_closed_loop.py 2>&1 | tail -22
