#!/bin/bash
# round 5, session o: records per bucket 512 / 1024, two candidates per lane and round in the compositor, split quadrants -- dense and sensor view
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5
cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.base.so
run() { # run <lib> <label> <tuning> <view> <configs>
  cp "$1" gsworld_amd/libgsr_hip.so
  GSWORLD_AMD_TUNING="$3" timeout 300 python tools/ab_batch.py --view $4 --steps 600 --configs $5 2>/dev/null | sed "s/^/$2 /"
}
for v in dense sensor; do
  run /tmp/libgsr_hip.base.so base "" $v batch1,batch8,3x8
  run tools/variants/libgsr_hip.per1024.so per1024 "" $v batch1,batch8,3x8
  run tools/variants/libgsr_hip.items2.so items2 "" $v batch1,batch8
  run /tmp/libgsr_hip.base.so split "render_split=1" $v batch1,batch8
done 2>&1 | tee $OUT/o_ab.txt
for lib in /tmp/libgsr_hip.base.so tools/variants/libgsr_hip.per1024.so; do
  cp $lib gsworld_amd/libgsr_hip.so
  echo "closed loop $(basename $lib)"; CL_ONLY=1,0 timeout 300 python tools/ab_closed_loop.py 2>/dev/null | cut -c1-160
  CL_ONLY=1,1 timeout 300 python tools/ab_closed_loop.py 2>/dev/null | cut -c1-160
done 2>&1 | tee $OUT/o_cl.txt
cp /tmp/libgsr_hip.base.so gsworld_amd/libgsr_hip.so
