#!/bin/bash
# rocprofv3 kernel trace + stats of the training-step bench (config 5); CSV summary lands in gpurun_out/prof_train_<tag>/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-t1}
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
shift || true
python tools/bench_train.py --steps 50 --warmup 5 "$@" 2>/dev/null | tail -1
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_train_$TAG" -o "$TAG" -- python "$REPO/tools/bench_train.py" --steps 30 --warmup 3 "$@" > "$REPO/gpurun_out/rocprof_train_$TAG.log" 2>&1)
f=$(find gpurun_out/prof_train_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python tools/show_stats.py "$f" | head -30
find gpurun_out/prof_train_$TAG -name "*kernel_trace.csv" -size +20M -delete
