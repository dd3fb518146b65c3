#!/bin/bash
# closed-loop and headline numbers + ss_buckets histogram in one call
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 250 python bench.py > gpurun_out/bench_ab.json 2> gpurun_out/bench_ab.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_ab.json'))
print("headline", round(d['value']), "closed_loop", round(d['closed_loop']['frames_per_s']), "ovf", d['closed_loop']['overflow_frames'],
      "moving", round(d['moving_camera']['frames_per_s']), "dense", {k: (round(v) if isinstance(v, float) else v) for k, v in d.get('dense_view', {}).items() if 'per_s' in k or 'fps' in k})
PY
bash tools/gpu_trace_hist.sh clh ss_buckets tools/closed_loop_surrogate.py --graph
