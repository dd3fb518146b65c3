#!/bin/bash
# kernel stats (eager, one frame in flight) for a list of tuning configs: tools/gpu_prof2.sh "cfgA" "cfgB" ...
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$PWD
for cfg in "$@"; do
  tag=$(echo "$cfg" | tr '=,' '__')
  (cd /tmp && GSWORLD_AMD_TUNING=$cfg timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_$tag" -o k -- python "$REPO/bench.py" --steps 100 --warmup 10 --no-graph --no-cpu-baseline --no-extras --in-flight 1 > "$REPO/gpurun_out/rocprof_$tag.log" 2>&1)
  echo "== $cfg"
  f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:34]
    if 'at::' in n or 'rocclr' in n: continue
    print(f"  {n:34s} {float(r['AverageNs'])/1000:7.1f} us  (min {float(r['MinNs'])/1000:6.1f})")
PY
  find gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete
done
