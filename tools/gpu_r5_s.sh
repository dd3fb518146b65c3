#!/bin/bash
# round 5, session s: cooperative quadrants in front of the grid, marked in the deal's entries; threshold variants
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5
cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.base.so
timeout 900 python -m pytest tests/test_renderer_gpu.py tests/test_forward_gpu.py tests/test_batch_gpu.py -q -m gpu -x 2>&1 | tail -3
run() { cp "$1" gsworld_amd/libgsr_hip.so; for v in dense sensor; do GSWORLD_AMD_TUNING="$3" timeout 300 python tools/ab_batch.py --view $v --steps 800 --configs batch1,batch2 2>/dev/null | sed "s/^/[$2] /"; done; }
{
run /tmp/libgsr_hip.base.so off "render_split=3"
run /tmp/libgsr_hip.base.so f40 ""
run tools/variants/libgsr_hip.f28.so f28 ""
run tools/variants/libgsr_hip.f56.so f56 ""
run tools/variants/libgsr_hip.f80.so f80 ""
run /tmp/libgsr_hip.base.so off "render_split=3"
} | tee $OUT/s.txt
cp /tmp/libgsr_hip.base.so gsworld_amd/libgsr_hip.so
