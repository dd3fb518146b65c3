#!/bin/bash
# kernel timeline of the 3-in-flight loop (hipGraph replay): gpurun_out/r4/trace_<tag>.csv
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4; REPO=$PWD
tr() { tag=$1; shift
 (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$tag -o k -- python $REPO/tools/ab_frame.py "$@" > $OUT/tr_$tag.log 2>&1)
 f=$(find $OUT/tr_$tag -name "*kernel_trace.csv" | head -1)
 [ -n "$f" ] && python - "$f" $OUT/trace_$tag.csv <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
rows = rows[-1200:]
t0 = int(rows[0]['Start_Timestamp'])
with open(sys.argv[2], 'w') as f:
    for r in rows:
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:40]
        f.write(f"{(int(r['Start_Timestamp'])-t0)/1000:.2f},{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000:.2f},{r.get('Queue_Id','')},{r.get('Stream_Id','')},{n}\n")
PY
 rm -rf $OUT/tr_$tag; grep -v "^{" $OUT/tr_$tag.log | tail -2
}
tr good3 "render_blocks_per_cu=6" --in-flight 3 --rounds 1 --steps 100
tr bad3 "render_blocks_per_cu=6" --in-flight 1,3 --rounds 1 --steps 100
tr q4 "render_blocks_per_cu=4" --in-flight 3 --rounds 1 --steps 100
