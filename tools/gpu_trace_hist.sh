#!/bin/bash
# per-launch duration histogram of one kernel: tools/gpu_trace_hist.sh <tag> <kernel-substring> <script> [args...]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$PWD; tag=$1; kern=$2; shift 2
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d "$REPO/gpurun_out/prof_$tag" -o k -- python "$REPO/$1" "${@:2}" > "$REPO/gpurun_out/rocprof_$tag.log" 2>&1)
f=$(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python - "$f" "$kern" <<'PY'
import csv, sys
import numpy as np
d = [ (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000 for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r['Kernel_Name']]
d = np.array(d)
print(sys.argv[2], "launches", len(d), "mean %.1f us" % d.mean(), "percentiles 10/50/90/99/max:", np.percentile(d, [10, 50, 90, 99, 100]).round(1).tolist())
print("  launches > 2x median:", int((d > 2 * np.median(d)).sum()), " their share of the total time: %.2f" % (d[d > 2 * np.median(d)].sum() / d.sum()))
PY
find gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete
