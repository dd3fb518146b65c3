#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4; export TMPDIR=/tmp
OUT=gpurun_out/r4
timeout 600 python tools/ab_frame.py "render_blocks_per_cu=0" --layouts 0,1,2 --in-flight 3,1 --tag sensor 2> $OUT/lay_sensor.err | grep -v "^{"; tail -3 $OUT/lay_sensor.err
timeout 600 python tools/ab_frame.py "render_blocks_per_cu=0" --layouts 0,1 --in-flight 3,1 --view dense --tag dense 2> $OUT/lay_dense.err | grep -v "^{"; tail -3 $OUT/lay_dense.err
