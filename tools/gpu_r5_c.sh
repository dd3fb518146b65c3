#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5
cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.base.so
cp tools/variants/libgsr_hip.sstiming.so gsworld_amd/libgsr_hip.so
timeout 300 python tools/ss_stamps_closed_loop.py 40 > gpurun_out/r5/c_stamps.txt 2>&1
cp /tmp/libgsr_hip.base.so gsworld_amd/libgsr_hip.so
cat gpurun_out/r5/c_stamps.txt | tail -50
