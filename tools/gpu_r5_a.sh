#!/bin/bash
# round 5, first GPU session: batched launches -- correctness, then frames per launch against frames per stream
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5; REPO=$PWD
timeout 900 python -m pytest tests/test_batch_gpu.py tests/test_renderer_gpu.py -x -q -m gpu > $OUT/a_tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/a_tests.log
tail -5 $OUT/a_tests.log
timeout 400 python tools/ab_batch.py --view sensor > $OUT/a_ab_sensor.jsonl 2> $OUT/a_ab_sensor.err; cat $OUT/a_ab_sensor.jsonl
timeout 400 python tools/ab_batch.py --view dense --configs streams1,streams4,batch1,batch4,batch8,2x4 > $OUT/a_ab_dense.jsonl 2> $OUT/a_ab_dense.err; cat $OUT/a_ab_dense.jsonl
timeout 400 python tools/ab_closed_loop.py > $OUT/a_cl.jsonl 2> $OUT/a_cl.err; cat $OUT/a_cl.jsonl
for cfg in batch1 batch4; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_$cfg -o k -- python $REPO/tools/ab_batch.py --eager --steps 200 --configs $cfg > $OUT/a_prof_$cfg.log 2>&1)
f=$(find $OUT/p_$cfg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_$cfg.csv
rm -rf $OUT/p_$cfg
python tools/show_stats.py $OUT/kernel_stats_$cfg.csv 2>/dev/null | head -16
done
