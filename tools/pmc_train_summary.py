#!/usr/bin/env python
"""The training step's REAL HBM traffic from the counter passes of tools/gpu_pmc_train.sh: per kernel 2 x FETCH_SIZE +
WRITE_SIZE (the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md), mean per dispatch, summed over one dispatch of
every kernel of the step (+ the memsets gsr_backward issues, which no kernel counter sees: their bytes are added as
written).  usage: pmc_train_summary.py <summary.txt> <out.json> [memset_bytes]"""
import datetime
import json
import re
import sys

path, out = sys.argv[1], sys.argv[2]
memset = int(sys.argv[3]) if len(sys.argv) > 3 else 0
what = sys.argv[4] if len(sys.argv) > 4 else "training step"  # (also used for the closed-loop step: tools/gpu_round6.sh clpmc)
only = sys.argv[5].split(",") if len(sys.argv) > 5 else None   # kernel-name substrings that belong to the step
rows = {}
for line in open(path):
    m = re.match(r"(.+?) \(grid (\d+), dispatches/pass=(\d+)\): (.*)", line)
    if not m:
        continue
    kv = dict(re.findall(r"(\w+)=([\d.e+]+)", m.group(4)))
    if "FETCH_SIZE" not in kv and "WRITE_SIZE" not in kv:
        continue
    name = m.group(1).strip()
    if only is not None and not any(o in name for o in only):
        continue
    rows[name] = {"fetch_kb": float(kv.get("FETCH_SIZE", 0)), "write_kb": float(kv.get("WRITE_SIZE", 0)),
                  "valu_wave_instructions": float(kv.get("SQ_INSTS_VALU", 0)), "dispatches_per_pass": int(m.group(3))}
    rows[name]["hbm_bytes"] = int((2.0 * rows[name]["fetch_kb"] + rows[name]["write_kb"]) * 1024)
total = sum(r["hbm_bytes"] for r in rows.values()) + memset
rec = {"what": what, "collected": datetime.date.today().isoformat(), "hbm_bytes_per_step": total, "memset_bytes_added": memset,
       "per_kernel": dict(sorted(rows.items(), key=lambda kv: -kv[1]["hbm_bytes"])),
       "method": "rocprofv3 --kernel-trace --pmc, one pass per counter group (tools/gpu_pmc_train.sh over tools/bench_train.py "
                 "--fused, or over PMC_CMD), mean per dispatch, 2 x FETCH_SIZE + WRITE_SIZE per kernel, one dispatch of every "
                 "kernel per step"}
json.dump(rec, open(out, "w"), indent=1)
print(f"{what}: {total / 1e6:.0f} MB of HBM traffic per step over {len(rows)} kernels -> {out}")
