#!/bin/bash
# round 5, session r: cooperative quadrants -- tests that compare compositor paths bit for bit, then one frame at a time on both views, coop on (default) / off
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5
timeout 900 python -m pytest tests/test_renderer_gpu.py tests/test_forward_gpu.py tests/test_batch_gpu.py tests/test_layout_gpu.py tests/test_fuzz_gpu.py -q -m gpu -x 2>&1 | tail -5
for v in dense sensor; do
  for t in "" "render_split=3" "render_split=2"; do
    GSWORLD_AMD_TUNING="$t" timeout 300 python tools/ab_batch.py --view $v --steps 600 --configs batch1,batch2,batch8 2>/dev/null | sed "s/^/[$t] /"
  done
done | tee $OUT/r_ab.txt
for t in "" "render_split=3"; do echo "closed loop [$t]"; GSWORLD_AMD_TUNING="$t" CL_ONLY=1,0 timeout 300 python tools/ab_closed_loop.py 2>/dev/null | cut -c1-120; GSWORLD_AMD_TUNING="$t" CL_ONLY=1,1 timeout 300 python tools/ab_closed_loop.py 2>/dev/null | cut -c1-120; done
