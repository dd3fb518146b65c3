#!/bin/bash
# Round-4 evidence run: everything profiles/round4/ holds (except the CPU-only files) comes out of this one script
# (see profiles/round4/README.md).  usage: gpurun -- 'bash tools/gpu_round4.sh [part ...]'   parts: bench stats pmc train dist
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/round4/pmc; export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out/round4
PARTS="${*:-bench stats pmc train dist}"
has() { [[ " $PARTS " == *" $1 "* ]]; }
stats() {  # stats <name> <python script> [args...]: rocprofv3 --kernel-trace --stats of one command -> $OUT/kernel_stats_<name>.csv
  local name=$1; shift
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tmp_$name" -o k -- python "$REPO/$1" "${@:2}" > "$OUT/rocprof_$name.log" 2>&1)
  f=$(find "$OUT/tmp_$name" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_$name.csv"
  rm -rf "$OUT/tmp_$name"; echo "== $name"; head -12 "$OUT/kernel_stats_$name.csv" | cut -d, -f1-4 | cut -c1-150
}
if has bench; then
  echo "== bench (default invocation)"; timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -2 $OUT/bench_default.err
  python - <<'PY'
import json
d = json.loads(open('gpurun_out/round4/bench_default.json').read().strip().splitlines()[-1])
print("value", round(d["value"]), "blocks", [round(x) for x in d["config"]["timed_blocks"]["frames_per_s"]], "one", d["config"]["one_frame_in_flight_frames_per_s"], "moving", d["config"]["moving_camera_frames_per_s"])
print("roofline", {k: d["roofline"][k] for k in ("frac", "kernel_ms", "valu_issue_frac")}, "frame", d["frame_roofline"]["frac_of_8TBs"], "p50", d["frame_roofline"]["frame_ms_p50"])
for k in ("dense_view", "closed_loop", "parity", "cpu_baseline"):
    print(k, {kk: vv for kk, vv in d.get(k, {}).items() if kk not in ("workload", "per_scene_worst", "sample", "against")})
print("train", {k: (v.get("ms_per_step") if isinstance(v, dict) else None) for k, v in d.get("train_step", {}).items()})
PY
fi
if has stats; then
  stats bench_one_frame_in_flight bench.py --steps 100 --warmup 10 --no-graph --no-cpu-baseline --no-extras --in-flight 1 --blocks 1
  stats bench_three_frames_in_flight bench.py --steps 100 --warmup 10 --no-graph --no-cpu-baseline --no-extras --in-flight 3 --blocks 1
  stats model_as_given bench.py --steps 100 --warmup 10 --no-graph --no-cpu-baseline --no-extras --in-flight 1 --blocks 1 --no-layout
  stats default_mode_frame tools/prof_scene.py --view sensor --default-mode
  stats dense_view tools/prof_scene.py --view dense
  stats moving_camera tools/prof_scene.py --view sensor --moving
  stats closed_loop tools/closed_loop_surrogate.py --graph
fi
if has train; then
  stats train_step_fused tools/bench_train.py --fused --steps 30
  echo "== pmc (training step)"; bash tools/gpu_pmc_train.sh round4/pmc_train_raw > $OUT/pmc/train_step.txt 2>&1; tail -12 $OUT/pmc/train_step.txt | cut -c1-300
  rm -rf gpurun_out/round4/pmc_train_raw/p*/
fi
if has pmc; then
  echo "== pmc (inference frame)"; bash tools/gpu_pmc.sh round4/pmc_raw 4 > $OUT/pmc/frame_config2.txt 2>&1; python tools/pmc_summary.py gpurun_out/round4/pmc_raw --json $OUT/pmc_render.json | tail -1
  grep -E "render_stream|preprocess|band_place" $OUT/pmc/frame_config2.txt | cut -c1-400
  rm -rf gpurun_out/round4/pmc_raw/p*/
fi
if has dist; then
  echo "== two ranks on this GPU over gloo (the N > 1 record's shape; RCCL needs an 8-GPU node)"
  GSWORLD_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 64 --warmup 16 --no-extras --no-cpu-baseline > $OUT/bench_two_ranks_gloo.json 2> $OUT/bench_two_ranks_gloo.err
  tail -1 $OUT/bench_two_ranks_gloo.json | cut -c1-600
fi
