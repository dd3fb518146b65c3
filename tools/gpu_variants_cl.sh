#!/bin/bash
# On the GPU box: closed-loop surrogate + moving-camera / dense figures for every tools/variants/libgsr_hip.<name>.so
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.base.so
for lib in /tmp/libgsr_hip.base.so tools/variants/libgsr_hip.*.so; do
  name=$(basename $lib .so); name=${name#libgsr_hip.}
  cp $lib gsworld_amd/libgsr_hip.so
  timeout 250 python bench.py > gpurun_out/vcl_$name.json 2> gpurun_out/vcl_$name.err
  python - gpurun_out/vcl_$name.json $name <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "headline", round(d['value']), "closed_loop", round(d['closed_loop']['frames_per_s']), "moving", round(d['moving_camera']['frames_per_s']), "dense", round(d['dense_view']['frames_per_s']), "train", round(d['train_step']['fused_packing']['ms_per_step'], 3))
PY
done
cp /tmp/libgsr_hip.base.so gsworld_amd/libgsr_hip.so
