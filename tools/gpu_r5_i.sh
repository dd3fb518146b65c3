#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5
for cfg in "4 3" "8 2" "8 3" "6 2" "4 4"; do
  set -- $cfg
  timeout 300 python bench.py --batch $1 --streams $2 --no-extras --no-cpu-baseline > $OUT/i_bench_$1x$2.json 2> $OUT/i_bench_$1x$2.err
  python -c "
import json; d=json.loads(open('$OUT/i_bench_$1x$2.json').read().strip().splitlines()[-1]); print('batch $1 streams $2: value', round(d['value']), 'one_stream', round(d['config']['one_stream_frames_per_s']), 'roofline', round(d['roofline']['frac'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), d['config']['timed_blocks']['frames_per_s_min_median_max'])"
done
for E in 2 4; do timeout 300 python tools/ab_closed_loop.py 1468850 $E 2>/dev/null | head -2; done
