#!/bin/bash
# round 5, session t: cooperative quadrants -- whole suite, closed loop on / off, kernel times of the dense and sensor view one frame at a time
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5; REPO=$PWD
timeout 900 python -m pytest tests/ -q -m gpu 2>&1 | tail -3
for t in "" "render_split=3" "" "render_split=3"; do echo "closed loop [$t]"; GSWORLD_AMD_TUNING="$t" CL_ONLY=1,0 timeout 300 python tools/ab_closed_loop.py 2>/dev/null | cut -c1-120; GSWORLD_AMD_TUNING="$t" CL_ONLY=1,1 timeout 300 python tools/ab_closed_loop.py 2>/dev/null | cut -c1-120; done
st() { local name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_$name -o k -- "$@" > $OUT/t_prof_$name.log 2>&1)
  f=$(find $OUT/p_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_t_$name.csv; rm -rf $OUT/p_$name
  echo "== $name"; python tools/show_stats.py $OUT/kernel_stats_t_$name.csv 6 | grep -v "at::"; }
st dense python $REPO/tools/prof_scene.py --view dense
st sensor python $REPO/tools/prof_scene.py --view sensor
GSWORLD_AMD_TUNING="render_split=3" st sensor_off python $REPO/tools/prof_scene.py --view sensor
