#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4; export TMPDIR=/tmp
for s in 3 4 3 4 5; do
timeout 600 python bench.py --no-extras --no-cpu-baseline --in-flight $s > gpurun_out/r4/b_$s.json 2>gpurun_out/r4/b_$s.err
python -c "
import json; d=json.loads(open('gpurun_out/r4/b_$s.json').read().strip().splitlines()[-1]); print('in-flight $s:', round(d['value']), [round(x) for x in d['config']['timed_blocks']['frames_per_s']], 'p50', round(d['frame_roofline']['frame_ms_p50'],4), 'prep', d['frame_roofline']['stage_ms'])"
done
