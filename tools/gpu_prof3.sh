#!/bin/bash
# kernel stats of tools/prof_scene.py for several argument sets: tools/gpu_prof3.sh "args1" "args2" ...
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$PWD
i=0
for a in "$@"; do
  i=$((i+1)); tag=ps$i
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_$tag" -o k -- python "$REPO/tools/prof_scene.py" $a > "$REPO/gpurun_out/rocprof_$tag.log" 2>&1)
  echo "== prof_scene.py $a  $(tail -1 gpurun_out/rocprof_$tag.log)"
  f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:34]
    if 'at::' in n or 'rocclr' in n: continue
    tot+=float(r['AverageNs'])/1000
    print(f"  {n:34s} {float(r['AverageNs'])/1000:7.1f} us  (min {float(r['MinNs'])/1000:6.1f})")
print(f"  sum {tot:.1f} us")
PY
  find gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete
done
