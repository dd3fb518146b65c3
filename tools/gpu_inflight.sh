#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for k in 1 2 3 4 6 8; do
  timeout 200 python bench.py --no-extras --no-cpu-baseline --steps 600 --in-flight $k 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('in_flight', $k, 'fps', round(d['value']), 'ms', round(d['ms_per_step'],4))"
done
