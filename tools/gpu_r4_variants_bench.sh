#!/bin/bash
# bench.py's secondary figures for the in-tree library and every tools/variants/libgsr_hip.<name>.so
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4; export TMPDIR=/tmp
cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.base.so
for lib in /tmp/libgsr_hip.base.so tools/variants/libgsr_hip.*.so; do
  name=$(basename $lib .so); name=${name#libgsr_hip.}
  cp $lib gsworld_amd/libgsr_hip.so
  timeout 300 python bench.py --steps 200 --blocks 2 --no-cpu-baseline > gpurun_out/r4/vb_$name.json 2> gpurun_out/r4/vb_$name.err
  python - gpurun_out/r4/vb_$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
cl = d["closed_loop"]
print(f"== {sys.argv[2]}: headline {d['value']:.0f}  one {d['config']['one_frame_in_flight_frames_per_s']:.0f}  moving {d['moving_camera']['frames_per_s']:.0f}  "
      f"closed loop {cl['frames_per_s']:.0f} (3 steps in flight {cl['three_steps_in_flight']['frames_per_s']:.0f}, overflow {cl['overflow_frames']})  dense {d['dense_view']['frames_per_s']:.0f}")
PY
done
cp /tmp/libgsr_hip.base.so gsworld_amd/libgsr_hip.so
