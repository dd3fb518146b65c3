#!/bin/bash
# round 5, session q: from how many frames per launch the larger buckets pay (closed loop with 2 / 4 environments, sensor view 4 per launch)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5
cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.base.so
for lib in /tmp/libgsr_hip.base.so tools/variants/libgsr_hip.ff8.so tools/variants/libgsr_hip.ff9.so; do
  cp $lib gsworld_amd/libgsr_hip.so; echo "== $(basename $lib)"
  for E in 2 4; do CL_ONLY=1,0 timeout 300 python tools/ab_closed_loop.py 1468850 $E 2>/dev/null | cut -c1-120; CL_ONLY=1,1 timeout 300 python tools/ab_closed_loop.py 1468850 $E 2>/dev/null | cut -c1-120; done
  timeout 300 python tools/ab_batch.py --view sensor --steps 600 --configs batch4,3x4,batch8 2>/dev/null
done 2>&1 | tee $OUT/q.txt
cp /tmp/libgsr_hip.base.so gsworld_amd/libgsr_hip.so
