#!/bin/bash
# the kernel sequence of the LAST occurrence window: tools/gpu_trace_seq.sh <tag> <anchor-kernel-substring> <script> [args...]
# prints every kernel launched between the last two launches of the anchor kernel (one step / frame of a loop)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$PWD; tag=$1; anchor=$2; shift 2
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$REPO/gpurun_out/prof_$tag" -o k -- python "$REPO/$1" "${@:2}" > "$REPO/gpurun_out/rocprof_$tag.log" 2>&1)
f=$(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python - "$f" "$anchor" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:90]
    print(f"{(int(r['Start_Timestamp'])-t0)/1000:8.1f} us  +{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000:6.1f}  {n}")
print("step span us", (int(rows[b]['Start_Timestamp']) - t0) / 1000)
PY
find gpurun_out/prof_$tag -name "*_trace.csv" -delete
