"""Aggregates rocprofv3 --pmc CSVs: per kernel, mean counter value per dispatch (last 4 frames only)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
        agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
keys = ["render", "tile_place", "tile_count", "preprocess", "radix_scatter", "radix_hist", "rowscan", "compact", "scan_small", "tile_starts", "ssim"]
for name in sorted(agg, key=lambda n: -sum(agg[n].get("SQ_WAVE_CYCLES", [0]))):
    if not any(k in name for k in keys):
        continue
    c = agg[name]
    n = max(len(v) for v in c.values())
    parts = []
    for k in sorted(c):
        vals = c[k][len(c[k]) // 3:]
        parts.append(f"{k}={sum(vals) / len(vals):.4g}")
    print(f"{name} (dispatches/pass={n}): " + " ".join(parts))
