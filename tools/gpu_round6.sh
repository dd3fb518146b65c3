#!/bin/bash
# Round-6 GPU sessions, one script with targets (round 5 left 23 one-off gpu_r5_*.sh files: folded into this form).
# usage: gpurun -- 'bash tools/gpu_round6.sh <target> [<target> ...]'
#   suite        the -m gpu test suite + smoke
#   cl           closed-loop surrogate (configs[2]) on the committed library, the variants in tools/variants/ and the
#                round-5 tree (.r5ref/, a worktree of round 5's last commit built next to this one): same box, same minute
#   cl_stats     rocprofv3 --kernel-trace --stats of the closed-loop surrogate -> gpurun_out/round6/kernel_stats_closed_loop.csv
#   headline     bench.py --no-extras on this tree and on .r5ref
#   bench        the driver's invocation(s) of bench.py -> gpurun_out/round6/bench_*.json
#   stats        kernel stats of the step's launches, one frame per launch, dense view, moving camera
#   train        training step: stats (+ pmc with `trainpmc`)
#   pmc          counters of the eight-frame step -> pmc_render.json
#   clpmc        counters of the closed-loop step -> pmc_closed_loop.json
#   knnssim      rocprofv3 stats of gsr_knn_dist2 at 1.47 M and gsr_ssim_forward / backward at 800 x 800
#   dist         two ranks over gloo on the one GPU (the N > 1 record's shape)
#   rccl1        RCCL itself under the frame gather, in a process group of one rank
#   cl_waited    policy in the loop: the graph's one submission against eleven launches from the kept pack
#   tiles        tiles the tile reuse leaves per camera and step on the surrogate
#   expacc       -DGSR_EXP_ACCURATE=1 on the eight full-size scenes of configs[3] (tools/variants/libgsr_hip.expacc.so)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/round6/pmc; export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out/round6
stats() {  # stats <name> <python script> [args...]: rocprofv3 --kernel-trace --stats of one command -> $OUT/kernel_stats_<name>.csv
  local name=$1; shift
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tmp_$name" -o k -- python "$REPO/$1" "${@:2}" > "$OUT/rocprof_$name.log" 2>&1)
  f=$(find "$OUT/tmp_$name" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_$name.csv"
  rm -rf "$OUT/tmp_$name"; echo "== $name"; python tools/show_stats.py "$OUT/kernel_stats_$name.csv" ${STATS_ROWS:-16}
}
cl_line() {  # one line per run of tools/ab_closed_loop.py: frames/s enqueued ahead / policy in the loop
  python - "$@" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
print(sys.argv[2], " ".join(f"{'policy' if r['policy_in_loop'] else 'ahead'}={r['frames_per_s']:.0f}" for r in rows),
      "overflow", sum(r["overflow_frames"] for r in rows))
PY
}
for target in "$@"; do
case $target in
suite)
  timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "Warning\|warnings.warn\|^$" | tail -${SUITE_TAIL:-40}
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-200
  ;;
cl)
  cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.base.so; : > $OUT/closed_loop_same_box.txt
  for rep in $(seq 1 ${CL_REPS:-2}); do
    for lib in /tmp/libgsr_hip.base.so tools/variants/libgsr_hip.*.so; do
      [ -f "$lib" ] || continue
      name=$(basename $lib .so); name=${name#libgsr_hip.}
      cp $lib gsworld_amd/libgsr_hip.so
      for E in ${CL_ENVS:-1}; do
        CL_ONLY=1 timeout 300 python tools/ab_closed_loop.py 1468850 $E > $OUT/cl_${name}_E$E.txt 2> $OUT/cl_${name}_E$E.err
        cl_line $OUT/cl_${name}_E$E.txt "$name E=$E" | tee -a $OUT/closed_loop_same_box.txt
      done
    done
    cp /tmp/libgsr_hip.base.so gsworld_amd/libgsr_hip.so
    if [ -d .r5ref ]; then
      for E in ${CL_ENVS:-1}; do
        (cd .r5ref && CL_ONLY=1 timeout 300 python tools/ab_closed_loop.py 1468850 $E > $OUT/cl_r5_E$E.txt 2> $OUT/cl_r5_E$E.err)
        cl_line $OUT/cl_r5_E$E.txt "round5 E=$E" | tee -a $OUT/closed_loop_same_box.txt
      done
    fi
  done
  ;;
cl_stats)
  CL_ONLY=1,0 stats closed_loop tools/ab_closed_loop.py
  ;;
cl_vstats)  # kernel stats of the closed-loop surrogate for every variant library
  cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.base.so
  for lib in tools/variants/libgsr_hip.*.so; do
    [ -f "$lib" ] || continue
    name=$(basename $lib .so); name=${name#libgsr_hip.}
    cp $lib gsworld_amd/libgsr_hip.so
    CL_ONLY=1,0 STATS_ROWS=11 stats closed_loop_$name tools/ab_closed_loop.py
  done
  cp /tmp/libgsr_hip.base.so gsworld_amd/libgsr_hip.so
  ;;
test)  # some tests, with their output: TEST_K='expr' (pytest -k)
  timeout 1200 python -m pytest tests/ -x -q -m gpu -k "${TEST_K:-config}" 2>&1 | tail -${TEST_TAIL:-70}
  ;;
cl_waited)  # a waited-for step (policy in the loop) as one graph replay against eleven launches from the kept argument pack
  line() { python -c "
import json,sys
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
print(' '.join(('policy' if r['policy_in_loop'] else 'ahead')+'='+str(round(r['frames_per_s'])) for r in rows))"; }
  for rep in 1 2; do
    for E in ${CL_ENVS:-1 2 4}; do
      echo "E=$E graph when waited, host values through set_poses / set_cameras: $(CL_NO_FAST_STAGE=1 CL_ONLY=1 timeout 300 python tools/ab_closed_loop.py 1468850 $E 2>/dev/null | line)"
      echo "E=$E graph when waited: $(CL_ONLY=1 timeout 300 python tools/ab_closed_loop.py 1468850 $E 2>/dev/null | line)"
      echo "E=$E eager when waited: $(CL_EAGER_WAITED=1 CL_ONLY=1 timeout 300 python tools/ab_closed_loop.py 1468850 $E 2>/dev/null | line)"
    done
  done | tee $OUT/closed_loop_waited.txt
  ;;
tiles)  # how many of its 1 200 tiles the tile reuse leaves per camera and step (marker-byte trick), full-size surrogate
  timeout 300 python tools/dbg_tile_reuse.py 1468850 1 2>&1 | grep "captured=False" | tee $OUT/tile_reuse_tiles_left.txt
  ;;
cl_env)  # the closed loop under HIP runtime knobs (what does the boundary between two graph replays cost, and why)
  for e in "X=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=64" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" \
           "ROC_SYSTEM_SCOPE_SIGNAL=0" "GPU_STREAMOPS_CP_WAIT=1" "DEBUG_HIP_DYNAMIC_QUEUES=0" "ROC_ACTIVE_WAIT_TIMEOUT=0" "GPU_MAX_HW_QUEUES=1"; do
    env $e CL_ONLY=1 timeout 300 python tools/ab_closed_loop.py 1468850 1 > $OUT/cl_env.txt 2> $OUT/cl_env.err
    cl_line $OUT/cl_env.txt "$e"
  done
  ;;
cl_eager)  # the closed loop without hipGraph replay: what the graph boundary costs against eleven eager launches
  CL_EAGER=1 CL_ONLY=1 timeout 300 python tools/ab_closed_loop.py 1468850 1 > $OUT/cl_eager.txt 2> $OUT/cl_eager.err
  cl_line $OUT/cl_eager.txt "eager E=1"
  CL_EAGER=1 CL_ONLY=1,0 bash tools/gpu_trace_seq.sh cl6e render_stream tools/ab_closed_loop.py 2>&1 | tail -14
  ;;
cl_seq)  # the kernel sequence of one closed-loop step with start offsets (gaps between launches)
  for E in ${CL_SEQ_ENVS:-1}; do
    echo "-- one step, $E environment(s)"
    CL_ONLY=1,0 bash tools/gpu_trace_seq.sh cl6 render_stream tools/ab_closed_loop.py 1468850 $E 2>&1 | tail -20
  done | tee $OUT/closed_loop_step_sequence.txt
  ;;
ab_v)  # tools/ab_batch.py on the committed library and every variant: AB_VIEW=dense|sensor AB_CONFIGS=batch1,batch8,3x8
  cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.base.so
  for lib in /tmp/libgsr_hip.base.so tools/variants/libgsr_hip.*.so; do
    [ -f "$lib" ] || continue
    name=$(basename $lib .so); name=${name#libgsr_hip.}
    cp $lib gsworld_amd/libgsr_hip.so
    for view in ${AB_VIEW:-dense}; do
      echo "$name $view: $(timeout 600 python tools/ab_batch.py --view $view --steps ${AB_STEPS:-400} --configs ${AB_CONFIGS:-batch1,batch8,3x8} 2>/dev/null | python -c "
import json,sys
print(' '.join(f\"{r['config']}={r['frames_per_s']:.0f}\" for r in map(json.loads, (l for l in sys.stdin if l.startswith('{')))))")"
    done
  done
  cp /tmp/libgsr_hip.base.so gsworld_amd/libgsr_hip.so
  ;;
headline_v)  # the headline on every variant library
  cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.base.so; : > $OUT/headline_variants.txt
  for lib in /tmp/libgsr_hip.base.so tools/variants/libgsr_hip.*.so; do
    [ -f "$lib" ] || continue
    name=$(basename $lib .so); name=${name#libgsr_hip.}
    cp $lib gsworld_amd/libgsr_hip.so
    timeout 600 python bench.py --no-extras --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$name', 'value', round(d['value']), 'one frame', round(c.get('one_frame_in_flight_frames_per_s') or 0), 'one stream', round(c.get('one_stream_frames_per_s') or 0), 'p50', d['frame_roofline'].get('frame_ms_p50'), 'compositor us per 8-frame launch', round(1e3 * d['roofline']['kernel_ms'], 1), 'one frame', round(1e3 * d['roofline']['one_frame_per_launch']['kernel_ms'], 1))" | tee -a $OUT/headline_variants.txt
  done
  cp /tmp/libgsr_hip.base.so gsworld_amd/libgsr_hip.so
  ;;
headline)
  for tree in . .r5ref; do
    [ -d $tree ] || continue
    (cd $tree && timeout 600 python bench.py --no-extras --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$tree', 'value', round(d['value']), 'one frame', round(c.get('one_frame_in_flight_frames_per_s') or 0), 'one stream', round(c.get('one_stream_frames_per_s') or 0), 'p50', d['frame_roofline'].get('frame_ms_p50'))")
  done
  ;;
bench)
  echo "== bench (default invocation)"; ( time timeout 1500 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -4 $OUT/bench_default.err
  echo "== bench (the driver's flags)"; timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err
  python tools/show_bench.py $OUT/bench_default.json $OUT/bench_driver_flags.json
  ;;
stats)
  stats bench_step_launches bench.py --steps 200 --warmup 10 --no-graph --no-cpu-baseline --no-extras --batch 8 --streams 1 --blocks 1 --min-seconds 0 --only-steps
  stats bench_one_frame_per_launch bench.py --steps 100 --warmup 10 --no-graph --no-cpu-baseline --no-extras --batch 1 --streams 1 --blocks 1 --min-seconds 0
  stats bench_headline_arrangement bench.py --steps 200 --warmup 10 --no-graph --no-cpu-baseline --no-extras --blocks 1 --min-seconds 0 --only-steps
  stats dense_view tools/prof_scene.py --view dense
  stats dense_view_step_launches tools/ab_batch.py --eager --view dense --steps 200 --configs batch8
  stats moving_camera tools/prof_scene.py --view sensor --moving
  stats default_mode_frame tools/prof_scene.py --view sensor --default-mode
  ;;
dense_v)  # the dense view, one frame per launch, per kernel: the committed library and every variant present
  # (NOT for the timing-diagnostics builds -- GSR_SS_DIAG, GSR_BWD_DIAG: their frames are wrong on purpose, and a wrong rect
  #  can keep a later kernel busy for good; session of round 6: 15 GPU-minutes until the limit)
  cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.base.so
  for lib in /tmp/libgsr_hip.base.so tools/variants/libgsr_hip.*.so; do
    [ -f "$lib" ] || continue
    name=$(basename $lib .so); name=${name#libgsr_hip.}
    cp $lib gsworld_amd/libgsr_hip.so
    STATS_ROWS=10 stats dense_view_$name tools/prof_scene.py --view dense
    echo "$name: $(python tools/ab_batch.py --view dense --steps 300 --configs batch1 2>/dev/null | tail -1 | cut -c1-120)"
  done
  cp /tmp/libgsr_hip.base.so gsworld_amd/libgsr_hip.so
  ;;
train)
  stats train_step_fused tools/bench_train.py --fused --steps 30
  ;;
fuzz)  # fuzz sweeps and soaks on the committed library -> $OUT/fuzz_and_soak.txt
  {
    timeout 600 python tools/fuzz_forward_only.py 500 61 2>&1 | tail -1
    timeout 600 python tools/fuzz_batch.py 300 62 2>&1 | tail -1
    timeout 600 python tools/fuzz_parity.py 100 63 2>&1 | tail -1
    timeout 600 python tools/fuzz_fused_backward.py 20 64 2>&1 | tail -1
    timeout 600 python tools/fuzz_backward.py 30 65 2>&1 | tail -1
    timeout 900 python tools/soak_static_scene.py 9000 inference 2>&1 | tail -1
    timeout 900 python tools/soak_moving_camera.py 600 400000 1 2>&1 | tail -1
    timeout 900 python tools/soak_moving_camera.py 300 1468850 2 2>&1 | tail -1
    timeout 900 python tools/soak_tile_reuse.py 1200 400000 1 7 2>&1 | tail -1
    timeout 900 python tools/soak_tile_reuse.py 400 1468850 2 8 2>&1 | tail -1
  } | cut -c1-400 | tee $OUT/fuzz_and_soak.txt
  ;;
hostprof)  # the host's share of a policy-in-the-loop step
  timeout 600 python tools/host_step_profile.py 2>&1 | cut -c1-200 | tee $OUT/host_step_profile.txt | head -70
  ;;
train_host)  # the fused training step: device-bound or host-bound?  (tiny problem = the host's floor)
  echo "configs[4]: $(timeout 600 python tools/bench_train.py --fused --steps 60 2>&1 | tail -1 | cut -c1-200)"
  echo "tiny: $(timeout 600 python tools/bench_train.py --fused --steps 200 --num-gaussians 2000 --size 64 2>&1 | tail -1 | cut -c1-200)"
  ;;
train_seq)  # the kernel sequence of one training step with start offsets
  bash tools/gpu_trace_seq.sh tr6 render_backward tools/bench_train.py --fused --steps 12 2>&1 | tail -40 | tee $OUT/train_step_sequence.txt
  ;;
gradbound)  # the configs[4]-size gradient comparison against the oracle, several runs per library: what the atomics' order is worth
  cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.base.so
  for lib in /tmp/libgsr_hip.base.so tools/variants/libgsr_hip.gradf32.so; do
    [ -f "$lib" ] || continue
    name=$(basename $lib .so); name=${name#libgsr_hip.}
    cp $lib gsworld_amd/libgsr_hip.so
    for rep in $(seq 1 ${GRAD_REPS:-4}); do
      timeout 600 python -m pytest tests/test_backward_gpu.py -q -s -k config5_full_size_gradients 2>&1 | grep "worst normalised" | sed "s/^/$name: /" | cut -c1-330
    done
  done | tee $OUT/config5_gradient_bound.txt
  cp /tmp/libgsr_hip.base.so gsworld_amd/libgsr_hip.so
  ;;
train_v)  # the fused training step on the committed library and every variant present (TRAIN_STATS=1: + per-kernel times)
  cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.base.so
  for lib in /tmp/libgsr_hip.base.so tools/variants/libgsr_hip.*.so; do
    [ -f "$lib" ] || continue
    name=$(basename $lib .so); name=${name#libgsr_hip.}
    cp $lib gsworld_amd/libgsr_hip.so
    for rep in $(seq 1 ${TRAIN_REPS:-2}); do
      echo "$name: $(timeout 600 python tools/bench_train.py --fused --steps 100 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('ms_per_step %.4f host %.3f' % (r['ms_per_step'], r['host_issue_ms_per_step']))")"
    done
    if [ "${TRAIN_STATS:-0}" = 1 ]; then
      stats train_step_$name tools/bench_train.py --fused --steps 30 2>/dev/null | grep -E "render_backward|geometry_backward|sh_backward|band_place|preprocess" | cut -c1-110
    fi
  done
  cp /tmp/libgsr_hip.base.so gsworld_amd/libgsr_hip.so
  ;;
clpmc)  # counters of the closed-loop step (configs[2] surrogate, steps enqueued ahead) -> pmc_closed_loop.json
  CL_ONLY=1,0 PMC_CMD="tools/ab_closed_loop.py 1468850 1" bash tools/gpu_pmc_train.sh round6/pmc_cl_raw > $OUT/pmc/closed_loop_step.txt 2>&1; grep -c "grid" $OUT/pmc/closed_loop_step.txt
  python tools/pmc_train_summary.py $OUT/pmc/closed_loop_step.txt $OUT/pmc_closed_loop.json 0 "closed-loop step (configs[2] surrogate, two frames)" \
    "render_stream,preprocess_kernel,band_,ss_,stage_step,tile_starts"
  rm -rf gpurun_out/round6/pmc_cl_raw/p*/
  ;;
trainpmc)
  bash tools/gpu_pmc_train.sh round6/pmc_train_raw > $OUT/pmc/train_step.txt 2>&1; grep -c "grid" $OUT/pmc/train_step.txt
  # (+ the one memset gsr_backward still issues: 500 k Gaussians x 12 binary64 words of the gradient records)
  python tools/pmc_train_summary.py $OUT/pmc/train_step.txt $OUT/pmc_train_step.json 48000000
  rm -rf gpurun_out/round6/pmc_train_raw/p*/
  ;;
pmc)
  echo "== pmc (the step's launches: 8 frames per launch)"
  PMC_BATCH=8 bash tools/gpu_pmc.sh round6/pmc_raw 4 > $OUT/pmc/step_config2.txt 2>&1
  PMC_BATCH=8 python tools/pmc_summary.py gpurun_out/round6/pmc_raw --json $OUT/pmc_render.json | tail -1
  grep -E "render_stream|preprocess|band_place|ss_compact|ss_buckets" $OUT/pmc/step_config2.txt | cut -c1-420
  rm -rf gpurun_out/round6/pmc_raw/p*/
  ;;
knnssim)
  STATS_ROWS=8 stats knn_dist2 tools/prof_knn_ssim.py knn
  STATS_ROWS=8 stats ssim tools/prof_knn_ssim.py ssim
  ;;
rccl1)  # RCCL under gsworld_amd.distributed in a process group of one rank (all a 1-GPU box can do)
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python tools/rccl_world1.py 24 1468850 2> $OUT/rccl_world1.err | grep "^{" | tail -1 | tee $OUT/rccl_world1.json | cut -c1-900
  ;;
dist)
  echo "== two ranks on this GPU over gloo (the N > 1 record's shape; RCCL needs an 8-GPU node)"
  GSWORLD_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 16 --warmup 6 --no-extras --no-cpu-baseline > $OUT/bench_two_ranks_gloo.json 2> $OUT/bench_two_ranks_gloo.err
  tail -1 $OUT/bench_two_ranks_gloo.json | cut -c1-400
  ;;
expacc)
  cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.base.so
  for lib in /tmp/libgsr_hip.base.so tools/variants/libgsr_hip.expacc.so; do
    [ -f "$lib" ] || continue
    name=$(basename $lib .so); name=${name#libgsr_hip.}
    cp $lib gsworld_amd/libgsr_hip.so
    timeout 900 python tools/exp_parity.py --full-size --json $OUT/exp_parity_$name.json 2>&1 | tail -12
  done
  cp /tmp/libgsr_hip.base.so gsworld_amd/libgsr_hip.so
  ;;
*)
  echo "unknown target $target"
  ;;
esac
done
