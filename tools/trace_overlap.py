#!/usr/bin/env python
"""Busy-time analysis of a rocprofv3 kernel trace (CSV): union of the kernel intervals against the wall span, the
sum of the durations (how much the streams overlap), and the per-kernel totals.  Usage: trace_overlap.py <kernel_trace.csv>
[skip_fraction]: the first skip_fraction of the dispatches (warm-up) is ignored."""
import csv
import re
import sys
from collections import defaultdict


def _name(full):
    m = re.search(r"(\w+)(<[^(]*>)?\(", full)
    return (m.group(1) if m else full)[:48]


def main():
    path = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), _name(r["Kernel_Name"])))
    rows.sort()
    rows = rows[int(len(rows) * skip):]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    for s, e, _ in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    total = sum(e - s for s, e, _ in rows)
    per = defaultdict(lambda: [0, 0])
    for s, e, n in rows:
        per[n][0] += 1
        per[n][1] += e - s
    wall = t1 - t0
    print(f"dispatches {len(rows)}  wall {wall / 1e3:.1f} us  busy(union) {busy / 1e3:.1f} us = {busy / wall:.1%}  "
          f"sum of durations {total / 1e3:.1f} us = {total / wall:.2f} x wall")
    for n, (c, d) in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {n:48s} calls {c:6d}  avg {d / c / 1e3:8.1f} us  share of wall {d / wall:6.1%}")


if __name__ == "__main__":
    main()
