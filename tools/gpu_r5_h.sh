#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5; REPO=$PWD
( time timeout 1500 python -m pytest tests/test_forward_gpu.py tests/test_batch_gpu.py tests/test_renderer_gpu.py tests/test_backward_gpu.py tests/test_fuzz_gpu.py -q -m gpu -x ) > $OUT/h_tests.log 2>&1; echo "tests rc=$?" >> $OUT/h_tests.log
grep -E "passed|failed|FAILED|rc=|real" $OUT/h_tests.log | tail -5
timeout 400 python tools/ab_batch.py --view sensor --configs batch1,batch4,3x4 > $OUT/h_ab.jsonl 2> $OUT/h_ab.err; cat $OUT/h_ab.jsonl
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_h -o k -- python $REPO/tools/ab_batch.py --eager --steps 400 --configs batch4 > $OUT/h_prof.log 2>&1)
f=$(find $OUT/p_h -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_batch4_h.csv; rm -rf $OUT/p_h
python tools/show_stats.py $OUT/kernel_stats_batch4_h.csv 11
