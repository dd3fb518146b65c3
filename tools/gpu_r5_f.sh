#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5; REPO=$PWD
cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.base.so
cp tools/variants/libgsr_hip.sstiming.so gsworld_amd/libgsr_hip.so
timeout 300 python tools/ss_stamps_closed_loop.py 40 > $OUT/f_stamps.txt 2>&1
cp /tmp/libgsr_hip.base.so gsworld_amd/libgsr_hip.so
grep -v amdgpu.ids $OUT/f_stamps.txt | grep -A2 "step 29\|step 39" | cut -c1-330
( time timeout 1200 python -m pytest tests/test_forward_gpu.py tests/test_layout_gpu.py tests/test_batch_gpu.py tests/test_renderer_gpu.py tests/test_fuzz_gpu.py -q -m gpu -x ) > $OUT/f_tests.log 2>&1; echo "tests rc=$?" >> $OUT/f_tests.log
grep -E "passed|failed|FAILED|rc=|real" $OUT/f_tests.log | tail
timeout 400 python tools/ab_closed_loop.py > $OUT/f_cl.jsonl 2> $OUT/f_cl.err; head -2 $OUT/f_cl.jsonl
(cd /tmp && CL_ONLY=1,0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_cl -o k -- python $REPO/tools/ab_closed_loop.py > $OUT/f_prof_cl.log 2>&1)
f=$(find $OUT/p_cl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_cl_f.csv; rm -rf $OUT/p_cl
python tools/show_stats.py $OUT/kernel_stats_cl_f.csv 8
