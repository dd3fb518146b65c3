#!/bin/bash
# One gpurun call: diagnostics, GPU tests, bench, rocprof.  Everything lands in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== diag"; timeout 600 python tools/gpu_diag.py > gpurun_out/diag.log 2>&1; tail -8 gpurun_out/diag.log
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -15 gpurun_out/pytest_gpu.log
echo "== bench"; timeout 600 python bench.py --steps 100 --warmup 10 --breakdown > gpurun_out/bench.log 2> gpurun_out/bench.err; cat gpurun_out/bench.log; tail -5 gpurun_out/bench.err
echo "== bench eager"; timeout 600 python bench.py --steps 100 --warmup 10 --no-graph --no-cpu-baseline > gpurun_out/bench_eager.log 2> gpurun_out/bench_eager.err; cat gpurun_out/bench_eager.log; tail -3 gpurun_out/bench_eager.err
echo "== rocprof"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o r1 -- python "$OLDPWD/bench.py" --steps 50 --warmup 5 --no-graph --no-cpu-baseline > "$OLDPWD/gpurun_out/rocprof.log" 2>&1); tail -3 gpurun_out/rocprof.log
find gpurun_out/prof -name "*stats*" | head; 
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f"
