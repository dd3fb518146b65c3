#!/bin/bash
# kernel stats of an arbitrary python command line: tools/gpu_prof_cmd.sh <tag> <script> [args...]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$PWD; tag=$1; shift
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_$tag" -o k -- python "$REPO/$1" "${@:2}" > "$REPO/gpurun_out/rocprof_$tag.log" 2>&1)
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:int(14)]:
    n = r['Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:44]
    print(f"  {n:44s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1000:7.1f} us  total {float(r['TotalDurationNs'])/1e6:7.2f} ms")
PY
find gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete
