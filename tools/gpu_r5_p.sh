#!/bin/bash
# round 5, session p: bucket size by launch occupancy (hdr->ss_B) -- tests + the three arrangements on both views + closed loop
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5
timeout 900 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -3
for v in dense sensor; do timeout 300 python tools/ab_batch.py --view $v --steps 600 --configs batch1,batch2,batch4,batch8,3x8 2>/dev/null; done | tee $OUT/p_ab.txt
CL_ONLY=1,0 timeout 300 python tools/ab_closed_loop.py 2>/dev/null | cut -c1-160
CL_ONLY=1,1 timeout 300 python tools/ab_closed_loop.py 2>/dev/null | cut -c1-160
CL_ONLY=1,0 timeout 300 python tools/ab_closed_loop.py 1468850 2 2>/dev/null | cut -c1-160
CL_ONLY=1,0 timeout 300 python tools/ab_closed_loop.py 1468850 4 2>/dev/null | cut -c1-160
