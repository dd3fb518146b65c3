#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4; export TMPDIR=/tmp
OUT=gpurun_out/r4
timeout 1200 python -m pytest tests/test_layout_gpu.py tests/test_closed_loop_gpu.py tests/test_renderer_gpu.py -x -q -m gpu > $OUT/pytest_ties.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_ties.log
python tools/scratch/dbg_det.py 2>&1 | grep -v amdgpu.ids | tail -3
timeout 600 python tools/ab_frame.py "render_blocks_per_cu=0" --layouts 1 --in-flight 3,1 --rounds 2 --tag sensor 2>/dev/null | grep -v "^{"
timeout 600 python tools/ab_frame.py "render_blocks_per_cu=0" --layouts 1 --in-flight 3,1 --rounds 2 --view dense --tag dense 2>/dev/null | grep -v "^{"
bash tools/gpu_r4_prof.sh dense2 "render_blocks_per_cu=0" --layouts 1 --in-flight 1 --rounds 1 --steps 200 --view dense | head -8
