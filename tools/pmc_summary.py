"""Aggregates rocprofv3 --pmc CSVs: per kernel, mean counter value per dispatch -- over the dispatches of the kernel's
LARGEST grid only (with PMC_BATCH = B frames per launch the run also holds the one-frame launches of the set-up: exact-mode
sizing frames; round 5's first summary averaged them in and understated the step's launch by 1.6x), first third of those
dropped (warm-up).  `--json PATH`: also writes the compositor's record (what bench.py quotes as
roofline.traffic) with the hash of the render.hip it was collected with."""
import csv
import datetime
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
json_out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
rows = []
for f in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
        rows.append((name, int(row.get("Grid_Size", 0) or 0), row["Counter_Name"], float(row["Counter_Value"])))
largest = defaultdict(int)
for name, grid, _, _ in rows:
    largest[name] = max(largest[name], grid)
agg = defaultdict(lambda: defaultdict(list))
for name, grid, counter, value in rows:
    if grid == largest[name]:
        agg[name][counter].append(value)
skip = ("at::native", "__amd_rocclr", "elementwise")
means = {}
for name in sorted(agg, key=lambda n: -sum(agg[n].get("SQ_WAVE_CYCLES", [0]))):
    if any(k in name for k in skip):
        continue
    c = agg[name]
    n = max(len(v) for v in c.values())
    means[name] = {}
    parts = []
    for k in sorted(c):
        vals = c[k][len(c[k]) // 3:]
        means[name][k] = sum(vals) / len(vals)
        parts.append(f"{k}={means[name][k]:.4g}")
    print(f"{name} (grid {largest[name]}, dispatches/pass={n}): " + " ".join(parts))
if json_out:
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = next((v for k, v in means.items() if "render_stream" in k), None)
    if r is None:
        raise SystemExit("no render_stream_kernel in the counter files")
    fetch_kb, write_kb = r.get("FETCH_SIZE", 0.0), r.get("WRITE_SIZE", 0.0)
    sha = hashlib.sha256(open(os.path.join(here, "gsworld_amd", "csrc", "render.hip"), "rb").read()).hexdigest()[:16]
    rec = {
        "kernel": "render_stream_kernel",
        "frames_per_launch": int(os.environ.get("PMC_BATCH", "1")),
        "workload": "config 2 frame (N=1468850, 640x480, right_cam)",
        "collected": datetime.date.today().isoformat(),
        "render_hip_sha16": sha,
        "FETCH_SIZE_kb_per_launch": fetch_kb, "WRITE_SIZE_kb_per_launch": write_kb,
        "hbm_bytes_per_launch": int((2.0 * fetch_kb + write_kb) * 1024),
        "counters": {k: round(v, 1) for k, v in r.items()},
        "method": "rocprofv3 --kernel-trace --pmc, one pass per counter group (tools/gpu_pmc.sh), mean over the "
                  "steady-state dispatches of the largest grid (the B-frame launches); FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for "
                  "gfx950 (128-B requests tallied at 64 B; calibrated there for 16 B/lane streaming reads -- this "
                  "kernel's reads are 16-B record gathers, so treat the read half as an upper-side estimate); "
                  "WRITE_SIZE taken as is",
        "source": os.path.relpath(root, here),
    }
    json.dump(rec, open(json_out, "w"), indent=1)
    print("wrote", json_out)
