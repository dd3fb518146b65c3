#!/bin/bash
# round 5, session y: soaks with cooperative quadrants (dense / sensor view, three states in flight) + whole suite + the evidence run
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5
SOAK_VIEW=dense timeout 300 python tools/soak_static_scene.py 6000 inference 2>&1 | tail -1
timeout 300 python tools/soak_static_scene.py 12000 inference 2>&1 | tail -1
timeout 600 python tools/fuzz_batch.py 200 21 2>&1 | tail -1
timeout 900 python -m pytest tests/ -q -m gpu 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
bash tools/gpu_round5.sh bench stats train pmc dist
