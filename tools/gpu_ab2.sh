#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "forward_only=0" "forward_only=1" "forward_only=1,binning_path=4"; do
  for inf in 1 3; do
    GSWORLD_AMD_TUNING=$cfg timeout 300 python bench.py --steps 300 --warmup 30 --no-extras --no-cpu-baseline --in-flight $inf --breakdown > gpurun_out/ab2.log 2> gpurun_out/ab2.err
    echo "== $cfg in-flight $inf: $(python -c "import json; d=json.loads(open('gpurun_out/ab2.log').read().strip().splitlines()[-1]); print(round(d['value']), 'fps', round(d['ms_per_step']*1000,1), 'us')") $(grep 'stage ms' gpurun_out/ab2.err | sed 's/.*stage ms: //' | python -c "import sys,ast; d=ast.literal_eval(sys.stdin.read() or '{}'); print({k: round(v*1000,1) for k,v in d.items()})")"
  done
done
