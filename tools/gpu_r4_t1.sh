#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_closed_loop_gpu.py tests/test_renderer_gpu.py tests/test_layout_gpu.py -x -q -m gpu > gpurun_out/r4/pytest_t1.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r4/pytest_t1.log
