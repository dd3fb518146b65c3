#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4; export TMPDIR=/tmp
OUT=gpurun_out/r4
cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.new.so
for v in stamps stamps_prio; do cp tools/variants/libgsr_hip.$v.so gsworld_amd/libgsr_hip.so
 for bpc in 6 4; do echo "== $v"; timeout 300 python tools/stream_stamps.py $OUT/stamps_${v}_$bpc.npz $bpc 2>$OUT/stamps.err; done; done
cp /tmp/libgsr_hip.new.so gsworld_amd/libgsr_hip.so
tail -3 $OUT/stamps.err
