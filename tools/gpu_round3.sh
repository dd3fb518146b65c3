#!/bin/bash
# Round-3 evidence run: everything profiles/round3/ holds comes out of this one script (see profiles/round3/README.md).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/round3/pmc; export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out/round3
stats() {  # stats <name> <python script> [args...]: rocprofv3 --kernel-trace --stats of one command -> $OUT/kernel_stats_<name>.csv
  local name=$1; shift
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tmp_$name" -o k -- python "$REPO/$1" "${@:2}" > "$OUT/rocprof_$name.log" 2>&1)
  f=$(find "$OUT/tmp_$name" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_$name.csv"
  rm -rf "$OUT/tmp_$name"; echo "== $name"; head -12 "$OUT/kernel_stats_$name.csv" | cut -d, -f1-4 | cut -c1-150
}
echo "== bench (default invocation)"; timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -2 $OUT/bench_default.err
stats bench_one_frame_in_flight bench.py --steps 100 --warmup 10 --no-graph --no-cpu-baseline --no-extras --in-flight 1
stats bench_three_frames_in_flight bench.py --steps 100 --warmup 10 --no-graph --no-cpu-baseline --no-extras --in-flight 3
stats default_mode_frame tools/prof_scene.py --view sensor --default-mode
stats dense_view tools/prof_scene.py --view dense
stats moving_camera tools/prof_scene.py --view sensor --moving
stats train_step_fused tools/bench_train.py --fused --steps 30
stats closed_loop tools/closed_loop_surrogate.py --graph
echo "== pmc (inference frame)"; bash tools/gpu_pmc.sh round3/pmc_raw 4 > $OUT/pmc/frame_config2.txt 2>&1; python tools/pmc_summary.py gpurun_out/round3/pmc_raw --json $OUT/pmc_render.json | tail -1
grep -E "render_stream|preprocess|band_place" $OUT/pmc/frame_config2.txt | cut -c1-400
rm -rf gpurun_out/round3/pmc_raw/p*/
