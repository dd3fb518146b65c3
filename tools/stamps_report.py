"""Report on tools/stream_stamps.py output (.npz): the shader clock is per CU, so every CU's stamps are taken relative to
its first quadrant's start (all workgroups of the resident grid start within a microsecond of each other)."""
import sys

import numpy as np

for f in sys.argv[1:]:
    d = np.load(f); s = d["stamps"]
    t0 = s[:, 0].astype(np.int64); t1 = s[:, 1].astype(np.int64); hw = s[:, 2]
    work = (s[:, 3] & 0x7FFFFFFF).astype(np.int64); queued = (s[:, 3] >> 31) != 0
    xcc = (hw >> 28).astype(np.int64)
    simd = ((hw >> 4) & 3).astype(np.int64); cu = ((hw >> 8) & 15).astype(np.int64)
    sh = ((hw >> 12) & 1).astype(np.int64); se = ((hw >> 13) & 7).astype(np.int64)
    cukey = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    for k in np.unique(cukey):
        m = cukey == k; b = t0[m].min(); t0[m] = (t0[m] - b) % (1 << 32); t1[m] = (t1[m] - b) % (1 << 32)
    span = t1.max()
    key = cukey * 4 + simd
    uniq, inv = np.unique(key, return_inverse=True)
    last = np.zeros(len(uniq), np.int64); np.maximum.at(last, inv, t1)
    cnt = np.bincount(inv); wsum = np.bincount(inv, weights=work)
    print(f"{f}: bpc {int(d['bpc'])}  CUs {len(np.unique(cukey))} SIMDs {len(uniq)}  span {span} cycles; queued units {int(queued.sum())}")
    print(f" units/SIMD {cnt.min()}/{cnt.mean():.2f}/{cnt.max()}; SIMD finish: mean {last.mean():.0f} p10 {np.percentile(last, 10):.0f} "
          f"p50 {np.percentile(last, 50):.0f} p90 {np.percentile(last, 90):.0f} p99 {np.percentile(last, 99):.0f} max {last.max()}")
    print(f" work/SIMD mean {wsum.mean():.0f} p10 {np.percentile(wsum, 10):.0f} p90 {np.percentile(wsum, 90):.0f} max {wsum.max():.0f}; "
          f"corr(work, finish) {np.corrcoef(wsum, last)[0, 1]:.3f}")
    print(f" unit life mean {np.mean(t1 - t0):.0f}; unit end mean {t1.mean():.0f}")
    edges = np.linspace(0, span, 21)
    print(" waves alive (5 % steps):", [int(((t0 <= e) & (t1 > e)).sum()) for e in edges[:-1]])
    # how fast a SIMD gets through its work (cycles per unit of work) against how much it had
    cpw = last / np.maximum(wsum, 1)
    print(f" SIMD cycles per work unit: mean {cpw.mean():.1f} p10 {np.percentile(cpw, 10):.1f} p90 {np.percentile(cpw, 90):.1f}")
    for c in sorted(set(cnt)):
        m = cnt == c
        print(f"  SIMDs with {c} units: {m.sum():4d}  finish mean {last[m].mean():.0f}  work mean {wsum[m].mean():.0f}  cycles/work {cpw[m].mean():.1f}")
    # the last wave of each SIMD: how long it ran alone
    alone = []
    for u in range(len(uniq)):
        e = np.sort(t1[inv == u])
        alone.append(e[-1] - e[-2] if len(e) > 1 else 0)
    alone = np.array(alone)
    print(f" last wave alone on its SIMD for: mean {alone.mean():.0f} p50 {np.percentile(alone, 50):.0f} p90 {np.percentile(alone, 90):.0f} cycles")
    # SIMDs of one CU
    cul = np.zeros(cukey.max() + 1, np.int64); np.maximum.at(cul, cukey, t1)
    cul = cul[cul > 0]
    print(f" CU finish: mean {cul.mean():.0f} p10 {np.percentile(cul, 10):.0f} p90 {np.percentile(cul, 90):.0f} max {cul.max()}")
    xl = [int(t1[xcc == x].max()) for x in range(8)]
    print(" XCD finish:", xl, " XCD work:", [int(work[xcc == x].sum()) for x in range(8)])
