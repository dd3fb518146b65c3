#!/bin/bash
# copies the judged summaries of the last `tools/gpu_round6.sh` runs from gpurun_out/round6 (scratch) into profiles/round6.
# A pmc_render.json that was collected with another render.hip than the one in the tree is REFUSED (round 5's driver line
# said "traffic: stale" because the counters were seven minutes older than the last edit of that file).
set -eu
cd "$(dirname "$0")/.."
S=gpurun_out/round6; D=profiles/round6
mkdir -p $D/pmc
for f in bench_default.json bench_driver_flags.json bench_two_ranks_gloo.json; do
  [ -f $S/$f ] && grep '^{' $S/$f | tail -1 > $D/$f || true
done
cp $S/kernel_stats_*.csv $D/ 2>/dev/null || true
cp $S/pmc/*.txt $D/pmc/ 2>/dev/null || true
for f in config5_gradient_bound.txt exp_parity_base.json exp_parity_expacc.json pmc_train_step.json pmc_closed_loop.json headline_variants.txt closed_loop_same_box.txt closed_loop_step_sequence.txt closed_loop_waited.txt rccl_world1.json tile_reuse_tiles_left.txt host_step_profile.txt train_step_sequence.txt; do [ -f $S/$f ] && cp $S/$f $D/ || true; done
if [ -f $S/pmc_render.json ]; then
  want=$(sha256sum gsworld_amd/csrc/render.hip | cut -c1-16)
  have=$(python -c "import json; print(json.load(open('$S/pmc_render.json'))['render_hip_sha16'])")
  if [ "$want" = "$have" ]; then
    cp $S/pmc_render.json $D/pmc_render.json; cp $S/pmc_render.json profiles/pmc_render.json; echo "pmc_render.json: render.hip $have (current)"
  else
    echo "REFUSED: $S/pmc_render.json was collected with render.hip $have, the tree holds $want -- run 'gpu_round6.sh pmc' again" >&2
    exit 1
  fi
fi
ls -la $D
