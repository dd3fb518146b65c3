#!/bin/bash
# rocprofv3 kernel trace + stats of the eager bench (CSV summaries land in gpurun_out/prof_<tag>/)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r1}
shift || true
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_$TAG" -o "$TAG" -- python "$REPO/bench.py" --steps 100 --warmup 10 --no-graph --no-cpu-baseline "$@" > "$REPO/gpurun_out/rocprof_$TAG.log" 2>&1)
tail -2 gpurun_out/rocprof_$TAG.log
find gpurun_out/prof_$TAG -type f | head -20
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cat "$f" | cut -c1-200 | head -40
# keep only the small summaries (the raw trace can be large)
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete
