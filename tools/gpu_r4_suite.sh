#!/bin/bash
# full GPU suite + the default bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4; export TMPDIR=/tmp
OUT=gpurun_out/r4
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_gpu.log
timeout 900 python bench.py "$@" > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -3 $OUT/bench_default.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4/bench_default.json').read().strip().splitlines()[-1])
print("value", round(d["value"]), "fps; blocks", [round(x) for x in d["config"]["timed_blocks"]["frames_per_s"]], "one", d["config"]["one_frame_in_flight_frames_per_s"], "moving", d["config"]["moving_camera_frames_per_s"])
print("roofline", {k: d["roofline"][k] for k in ("frac", "kernel_ms", "valu_issue_frac")}, "frame frac", d["frame_roofline"]["frac_of_8TBs"], "p50", d["frame_roofline"]["frame_ms_p50"])
for k in ("dense_view", "closed_loop", "moving_camera", "upstream_packing", "train_step"):
    v = d.get(k, {})
    print(k, {kk: vv for kk, vv in v.items() if kk not in ("workload",)} if k != "train_step" else {kk: (vv.get("ms_per_step") if isinstance(vv, dict) else vv) for kk, vv in v.items() if kk != "workload"})
PY
