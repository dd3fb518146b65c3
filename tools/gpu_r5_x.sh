#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.base.so
for rep in 1 2; do for l in head cur; do cp tools/variants/libgsr_hip.$l.so gsworld_amd/libgsr_hip.so; for v in dense sensor; do timeout 100 python tools/ab_batch.py --view $v --steps 800 --configs batch1,3x8 2>/dev/null | sed "s/^/[$l] /" | cut -c1-90; done; done; done
cp /tmp/libgsr_hip.base.so gsworld_amd/libgsr_hip.so
