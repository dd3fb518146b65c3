#!/bin/bash
# round 5: the whole GPU suite (no -x), then the closed loop / batch A/B after the ss_compact prologue rewrite
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5; REPO=$PWD
( time timeout 2400 python -m pytest tests/ -q -m gpu ) > $OUT/e_tests.log 2>&1; echo "tests rc=$?" >> $OUT/e_tests.log
grep -E "passed|failed|FAILED|Error|rc=|real" $OUT/e_tests.log | tail -30
timeout 400 python tools/ab_closed_loop.py > $OUT/e_cl.jsonl 2> $OUT/e_cl.err; cat $OUT/e_cl.jsonl
timeout 400 python tools/ab_batch.py --view dense --configs streams1,batch4,3x4 > $OUT/e_ab_dense.jsonl 2> $OUT/e_ab_dense.err; cat $OUT/e_ab_dense.jsonl
(cd /tmp && CL_ONLY=1,0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_cl -o k -- python $REPO/tools/ab_closed_loop.py > $OUT/e_prof_cl.log 2>&1)
f=$(find $OUT/p_cl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_cl_e.csv; rm -rf $OUT/p_cl
python tools/show_stats.py $OUT/kernel_stats_cl_e.csv 14
timeout 300 python bench.py --no-extras --no-cpu-baseline > $OUT/e_bench.json 2> $OUT/e_bench.err; python -c "
import json; d=json.loads(open('$OUT/e_bench.json').read().strip().splitlines()[-1]); print('value', d['value'], 'one', d['config']['one_frame_in_flight_frames_per_s'], 'one_stream', d['config']['one_stream_frames_per_s'], 'p50', d['frame_roofline']['frame_ms_p50'], d['frame_roofline']['stage_ms_one_frame_per_launch'])"
