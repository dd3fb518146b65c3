#!/bin/bash
# On the GPU box: for every tools/variants/libgsr_hip.<name>.so run the one-frame-in-flight bench with the stage table.
# usage: tools/gpu_variants.sh "<env assignments>" [bench args...]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
ENVS=${1:-}; shift || true
cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.base.so
for lib in /tmp/libgsr_hip.base.so tools/variants/libgsr_hip.*.so; do
  name=$(basename $lib .so); name=${name#libgsr_hip.}
  cp $lib gsworld_amd/libgsr_hip.so
  env $ENVS timeout 300 python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline --in-flight 1 --breakdown "$@" > gpurun_out/var_$name.log 2> gpurun_out/var_$name.err
  echo "== $name: $(python -c "import json,sys; d=json.loads(open('gpurun_out/var_$name.log').read().strip().splitlines()[-1]); print(round(d['value']), 'fps', round(d['ms_per_step']*1000,1), 'us')" 2>/dev/null) $(grep 'stage ms' gpurun_out/var_$name.err | sed 's/.*stage ms: //')"
done
cp /tmp/libgsr_hip.base.so gsworld_amd/libgsr_hip.so
