#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5; REPO=$PWD
cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.base.so
cp tools/variants/libgsr_hip.sstiming.so gsworld_amd/libgsr_hip.so
timeout 300 python tools/ss_stamps_closed_loop.py 60 2>&1 | grep -v amdgpu.ids > $OUT/n_stamps.txt
cp /tmp/libgsr_hip.base.so gsworld_amd/libgsr_hip.so
grep "^step" $OUT/n_stamps.txt | cut -c1-140; grep "buckets wg100" $OUT/n_stamps.txt | tail -4 | cut -c1-200
st() { local name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_$name -o k -- "$@" > $OUT/n_prof_$name.log 2>&1)
  f=$(find $OUT/p_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_n_$name.csv; rm -rf $OUT/p_$name
  echo "== $name"; python tools/show_stats.py $OUT/kernel_stats_n_$name.csv 16 | grep "ss_"; }
CL_ONLY=1,0 st cl python $REPO/tools/ab_closed_loop.py
st moving python $REPO/tools/prof_scene.py --view sensor --moving
timeout 300 python tools/ab_closed_loop.py 2>/dev/null | head -2
