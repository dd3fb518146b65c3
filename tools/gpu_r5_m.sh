#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5; REPO=$PWD
( time timeout 2400 python -m pytest tests/ -q -m gpu ) > $OUT/m_tests.log 2>&1; echo "tests rc=$?" >> $OUT/m_tests.log
grep -E "passed|failed|FAILED|Error|rc=|real" $OUT/m_tests.log | tail -8
st() { local name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_$name -o k -- "$@" > $OUT/m_prof_$name.log 2>&1)
  f=$(find $OUT/p_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_m_$name.csv; rm -rf $OUT/p_$name
  echo "== $name"; python tools/show_stats.py $OUT/kernel_stats_m_$name.csv 16 | grep "ss_\|preprocess\|render_stream\|band_\|tile_"; }
st batch1 python $REPO/tools/ab_batch.py --eager --steps 300 --configs batch1
st batch8 python $REPO/tools/ab_batch.py --eager --steps 400 --configs batch8
st dense8 python $REPO/tools/ab_batch.py --eager --view dense --steps 200 --configs batch8
st dense1 python $REPO/tools/ab_batch.py --eager --view dense --steps 200 --configs batch1
CL_ONLY=1,0 st cl python $REPO/tools/ab_closed_loop.py
