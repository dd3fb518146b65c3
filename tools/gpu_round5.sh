#!/bin/bash
# Round-5 evidence run: everything profiles/round5/ holds comes out of this one script (see profiles/round5/README.md).
# usage: gpurun -- 'bash tools/gpu_round5.sh [part ...]'   parts: bench stats pmc train dist
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/round5/pmc; export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out/round5
PARTS="${*:-bench stats pmc train dist}"
has() { [[ " $PARTS " == *" $1 "* ]]; }
stats() {  # stats <name> <python script> [args...]: rocprofv3 --kernel-trace --stats of one command -> $OUT/kernel_stats_<name>.csv
  local name=$1; shift
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tmp_$name" -o k -- python "$REPO/$1" "${@:2}" > "$OUT/rocprof_$name.log" 2>&1)
  f=$(find "$OUT/tmp_$name" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_$name.csv"
  rm -rf "$OUT/tmp_$name"; echo "== $name"; python tools/show_stats.py "$OUT/kernel_stats_$name.csv" 13
}
if has bench; then
  echo "== bench (default invocation)"; ( time timeout 1200 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -4 $OUT/bench_default.err
  echo "== bench (the driver's flags)"; timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err
  python - <<'PY'
import json
for f in ("bench_default", "bench_driver_flags"):
    d = json.loads(open(f'gpurun_out/round5/{f}.json').read().strip().splitlines()[-1])
    tb = d["config"]["timed_blocks"]
    print(f, "value", round(d["value"]), "blocks", tb["blocks"], [round(x) for x in tb["frames_per_s_min_median_max"]], "total-time", round(tb["frames_per_s_total_time"]),
          "one", round(d["config"]["one_frame_in_flight_frames_per_s"] or 0), "one stream", round(d["config"]["one_stream_frames_per_s"] or 0))
    print("  roofline", {k: d["roofline"][k] for k in ("frac", "kernel_ms", "frames_per_launch")}, "one/launch", d["roofline"]["one_frame_per_launch"]["frac"],
          "frame", d["frame_roofline"]["frac_of_8TBs"], "p50", d["frame_roofline"]["frame_ms_p50"])
    for k in ("dense_view", "closed_loop", "moving_camera", "cpu_baseline"):
        if k in d:
            print("  ", k, {kk: vv for kk, vv in d[k].items() if kk not in ("workload", "sample", "three_steps_in_flight")})
    if "parity" in d:
        p = d["parity"]
        print("   parity", p.get("worst_pixel_all_scenes"), p.get("worst_pixel_off_borderline"), p.get("per_scene_worst"), (p.get("config5_gradients") or {}).get("worst_normalised_error"))
    if "train_step" in d:
        print("   train", {k: (v.get("ms_per_step") if isinstance(v, dict) else None) for k, v in d["train_step"].items()})
PY
fi
if has stats; then
  # the step's own launches, alone on the chip: B = 8 frames per launch, one stream, eager (what roofline.kernel_ms times)
  stats bench_step_launches bench.py --steps 200 --warmup 10 --no-graph --no-cpu-baseline --no-extras --batch 8 --streams 1 --blocks 1 --min-seconds 0 --only-steps
  stats bench_one_frame_per_launch bench.py --steps 100 --warmup 10 --no-graph --no-cpu-baseline --no-extras --batch 1 --streams 1 --blocks 1 --min-seconds 0
  stats bench_headline_arrangement bench.py --steps 200 --warmup 10 --no-graph --no-cpu-baseline --no-extras --blocks 1 --min-seconds 0 --only-steps
  stats dense_view tools/prof_scene.py --view dense
  stats dense_view_step_launches tools/ab_batch.py --eager --view dense --steps 200 --configs batch8
  stats moving_camera tools/prof_scene.py --view sensor --moving
  CL_ONLY=1,0 stats closed_loop tools/ab_closed_loop.py
  stats default_mode_frame tools/prof_scene.py --view sensor --default-mode
fi
if has train; then
  stats train_step_fused tools/bench_train.py --fused --steps 30
fi
if has pmc; then
  echo "== pmc (the step's launches: 8 frames per launch)"
  PMC_BATCH=8 bash tools/gpu_pmc.sh round5/pmc_raw 4 > $OUT/pmc/step_config2.txt 2>&1
  PMC_BATCH=8 python tools/pmc_summary.py gpurun_out/round5/pmc_raw --json $OUT/pmc_render.json | tail -1
  grep -E "render_stream|preprocess|band_place|ss_compact|ss_buckets" $OUT/pmc/step_config2.txt | cut -c1-420
  rm -rf gpurun_out/round5/pmc_raw/p*/
fi
if has dist; then
  echo "== two ranks on this GPU over gloo (the N > 1 record's shape; RCCL needs an 8-GPU node)"
  GSWORLD_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 16 --warmup 6 --no-extras --no-cpu-baseline > $OUT/bench_two_ranks_gloo.json 2> $OUT/bench_two_ranks_gloo.err
  tail -1 $OUT/bench_two_ranks_gloo.json | cut -c1-400
fi
