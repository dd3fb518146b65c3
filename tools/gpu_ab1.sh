#!/bin/bash
# round 3, call 1: new tests + A/B of the chunk placement against the band placement
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest (new tests)"
timeout 900 python -m pytest tests -m gpu -x -q -k "chunk or alternative or glue or asset or fixture or reference_glue or config_json" > gpurun_out/pytest_new.log 2>&1; tail -15 gpurun_out/pytest_new.log
for bp in 0 4; do
  echo "== bench binning_path=$bp (3 in flight)"
  GSWORLD_AMD_TUNING=binning_path=$bp timeout 300 python bench.py --steps 300 --warmup 30 --no-extras --no-cpu-baseline --breakdown > gpurun_out/bench_bp$bp.log 2> gpurun_out/bench_bp$bp.err; cat gpurun_out/bench_bp$bp.log | cut -c1-600; grep -i "stage\|ms\|us" gpurun_out/bench_bp$bp.err | tail -12
  echo "== bench binning_path=$bp (1 in flight)"
  GSWORLD_AMD_TUNING=binning_path=$bp timeout 300 python bench.py --steps 300 --warmup 30 --no-extras --no-cpu-baseline --in-flight 1 > gpurun_out/bench1_bp$bp.log 2> gpurun_out/bench1_bp$bp.err; cat gpurun_out/bench1_bp$bp.log | cut -c1-400
done
echo "== rocprof path 4"
REPO=$PWD
(cd /tmp && GSWORLD_AMD_TUNING=binning_path=4 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_bp4" -o bp4 -- python "$REPO/bench.py" --steps 100 --warmup 10 --no-graph --no-cpu-baseline --no-extras --in-flight 1 > "$REPO/gpurun_out/rocprof_bp4.log" 2>&1)
f=$(find gpurun_out/prof_bp4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-8 "$f" | head -20
find gpurun_out/prof_bp4 -name "*kernel_trace.csv" -size +20M -delete
