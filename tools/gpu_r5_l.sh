#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5
cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.base.so
cp tools/variants/libgsr_hip.sstiming.so gsworld_amd/libgsr_hip.so
timeout 300 python tools/ss_stamps_view.py dense 6 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/l_stamps_dense.txt
timeout 300 python tools/ss_stamps_view.py sensor 6 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/l_stamps_sensor.txt
cp /tmp/libgsr_hip.base.so gsworld_amd/libgsr_hip.so
