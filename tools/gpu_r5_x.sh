#!/bin/bash
# A/B of two library builds in one box: tools/variants/libgsr_hip.head.so against libgsr_hip.cur.so
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.base.so
timeout 600 python -m pytest tests/test_forward_gpu.py tests/test_renderer_gpu.py tests/test_layout_gpu.py tests/test_fuzz_gpu.py -q -m gpu 2>&1 | tail -1
for rep in 1 2; do for l in head cur; do cp tools/variants/libgsr_hip.$l.so gsworld_amd/libgsr_hip.so; for v in dense sensor; do timeout 100 python tools/ab_batch.py --view $v --steps 800 --configs ${AB_CONFIGS:-batch1,batch8} 2>/dev/null | sed "s/^/[$l] /" | cut -c1-90; done; done; done
cp /tmp/libgsr_hip.base.so gsworld_amd/libgsr_hip.so
