#!/bin/bash
# kernel stats of the headline frame for the in-tree library and every tools/variants/libgsr_hip.<name>.so
# (tools/build_variants.sh), interleaved twice: tools/gpu_r4_variants_prof.sh [prof_scene args...]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4; REPO=$PWD
cp gsworld_amd/libgsr_hip.so /tmp/libgsr_hip.base.so
for round in $(seq 1 ${ROUNDS:-2}); do
for lib in /tmp/libgsr_hip.base.so tools/variants/libgsr_hip.*.so; do
  name=$(basename $lib .so); name=${name#libgsr_hip.}
  cp $lib gsworld_amd/libgsr_hip.so
  (cd /tmp && timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pv_$name -o k -- python $REPO/tools/prof_scene.py --frames 300 "$@" > $OUT/pv_$name.log 2>&1)
  f=$(find $OUT/pv_$name -name "*kernel_stats.csv" | head -1)
  echo "== $name round $round: $(python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows) / 300e3
rk = [r for r in rows if 'render_stream' in r['Name']]
print(f"render {float(rk[0]['AverageNs'])/1e3:.2f} us, all kernels {tot:.1f} us/frame")
PY
)"
  rm -rf $OUT/pv_$name
done
done
cp /tmp/libgsr_hip.base.so gsworld_amd/libgsr_hip.so
