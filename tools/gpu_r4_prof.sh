#!/bin/bash
# kernel stats of tools/ab_frame.py runs: tools/gpu_r4_prof.sh <tag> <ab_frame args...>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4; REPO=$PWD; tag=$1; shift
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_$tag -o k -- python $REPO/tools/ab_frame.py "$@" > $OUT/prof_$tag.log 2>&1)
f=$(find $OUT/p_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_$tag.csv
rm -rf $OUT/p_$tag
python - $OUT/kernel_stats_$tag.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:13]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:36]
    print(f"  {n:36s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:7.1f} max {float(r['MaxNs'])/1e3:7.1f}")
PY
