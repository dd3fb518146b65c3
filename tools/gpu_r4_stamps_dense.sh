#!/bin/bash
# per-quadrant stamps of the compositor for a view: tools/gpu_r4_stamps_dense.sh [sensor|dense]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4; export TMPDIR=/tmp
OUT=gpurun_out/r4; view=${1:-dense}
cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.new.so
cp tools/variants/libgsr_hip.stamps.so gsworld_amd/libgsr_hip.so
timeout 120 python tools/stream_stamps.py $OUT/stamps_$view.npz 0 --view $view 2>$OUT/stamps.err
timeout 60 python tools/stamps_report.py $OUT/stamps_$view.npz
cp /tmp/libgsr_hip.new.so gsworld_amd/libgsr_hip.so
tail -3 $OUT/stamps.err
