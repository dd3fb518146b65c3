#!/bin/bash
# usage: sweep_bench.sh "inflight bpc" ...   -> one line per configuration
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for cfg in "$@"; do
  set -- $cfg
  out=$(timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --in-flight $1 --render-bpc $2 2>/dev/null)
  echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('inflight $1 bpc $2 fps', round(d['value'],1))"
done
