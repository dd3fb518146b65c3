#!/bin/bash
# round 5, session u: sort stamps of the closed loop on the committed library + fuzz sweeps + the evidence run
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5
cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.base.so
cp tools/variants/libgsr_hip.sstiming.so gsworld_amd/libgsr_hip.so
timeout 300 python tools/ss_stamps_closed_loop.py 60 2>&1 | grep -v amdgpu.ids > $OUT/n_stamps.txt
cp /tmp/libgsr_hip.base.so gsworld_amd/libgsr_hip.so
head -4 $OUT/n_stamps.txt | cut -c1-200
timeout 600 python tools/fuzz_forward_only.py 300 11 2>&1 | tail -1
timeout 600 python tools/fuzz_batch.py 150 12 2>&1 | tail -1
timeout 600 python tools/fuzz_parity.py 60 13 2>&1 | tail -1
bash tools/gpu_round5.sh bench stats train pmc dist
