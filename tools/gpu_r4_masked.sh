#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4; export TMPDIR=/tmp
OUT=gpurun_out/r4
timeout 1200 python -m pytest tests/test_renderer_gpu.py tests/test_forward_gpu.py tests/test_layout_gpu.py -x -q -m gpu > $OUT/pytest_masked.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_masked.log
cp gsworld_amd/libgsr_hip.so /tmp/new.so
for v in new select new select; do
  if [ $v = new ]; then cp /tmp/new.so gsworld_amd/libgsr_hip.so; else cp tools/variants/libgsr_hip.select.so gsworld_amd/libgsr_hip.so; fi
  timeout 600 python tools/ab_frame.py "render_blocks_per_cu=0" --layouts 1 --in-flight 3,1 --rounds 2 --tag $v 2>/dev/null | grep -v "^{"
done
cp /tmp/new.so gsworld_amd/libgsr_hip.so
timeout 600 python tools/ab_frame.py "render_blocks_per_cu=0" --layouts 1 --in-flight 3,1 --rounds 2 --view dense --tag new_dense 2>/dev/null | grep -v "^{"
