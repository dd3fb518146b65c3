#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4; export TMPDIR=/tmp
OUT=gpurun_out/r4
timeout 900 python -m pytest tests/test_layout_gpu.py -x -q -m gpu > $OUT/pytest_layout.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_layout.log
timeout 600 python tools/ab_frame.py "render_blocks_per_cu=0" --layouts 0,1 --in-flight 3,1 --tag sensor 2> $OUT/lay_sensor.err | grep -v "^{"; tail -2 $OUT/lay_sensor.err
timeout 600 python tools/ab_frame.py "render_blocks_per_cu=0" --layouts 0,1 --in-flight 3,1 --view dense --tag dense 2> $OUT/lay_dense.err | grep -v "^{"; tail -2 $OUT/lay_dense.err
bash tools/gpu_r4_prof.sh lay1 "render_blocks_per_cu=0" --layouts 1 --in-flight 1 --rounds 1 --steps 300
