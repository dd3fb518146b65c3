#!/bin/bash
# copies the judged summaries of the last `tools/gpu_round5.sh` run from gpurun_out/round5 (scratch) into profiles/round5
set -eu
cd "$(dirname "$0")/.."
S=gpurun_out/round5; D=profiles/round5
mkdir -p $D/pmc
for f in bench_default.json bench_driver_flags.json bench_two_ranks_gloo.json pmc_render.json; do [ -f $S/$f ] && tail -1 $S/$f > /dev/null && cp $S/$f $D/$f; done
# (bench output files hold the one JSON line + the `time` report: keep the line only)
for f in bench_default.json bench_driver_flags.json bench_two_ranks_gloo.json; do [ -f $D/$f ] && grep '^{' $D/$f | tail -1 > $D/$f.tmp && mv $D/$f.tmp $D/$f; done
cp $S/kernel_stats_*.csv $D/ 2>/dev/null || true
cp $S/pmc/step_config2.txt $D/pmc/ 2>/dev/null || true
[ -f $D/pmc_render.json ] && cp $D/pmc_render.json profiles/pmc_render.json
[ -f gpurun_out/r5/n_stamps.txt ] && grep -v amdgpu.ids gpurun_out/r5/n_stamps.txt > $D/ss_stamps_closed_loop.txt
ls -la $D
