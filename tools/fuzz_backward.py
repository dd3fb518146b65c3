#!/usr/bin/env python
"""Randomised gradient parity on the GPU: gsr_backward vs the backward oracle (tests/helpers_bwd.run_case tolerance)
over random sizes, image shapes, SH degrees, antialiasing and splat scales.  Usage: fuzz_backward.py [iterations] [seed]."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import helpers_bwd as hb  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    t0 = time.time()
    worst = 0.0
    for it in range(iters):
        n = int(rng.choice([50, 700, 4000, 12000]))
        w, h = int(rng.integers(8, 200)), int(rng.integers(8, 150))
        rep = hb.run_case(n, w, h, seed=int(rng.integers(1 << 30)), aa=bool(rng.random() < 0.3), deg=int(rng.integers(0, 4)),
                          bg=tuple(rng.random(3).tolist()), scale_boost=float(rng.uniform(-0.5, 1.5)),
                          with_invdepth=bool(rng.random() < 0.7))
        worst = max(worst, max(v["max_norm_err"] for v in rep.values()))
    print(f"fuzz backward ok: {iters} cases, seed {seed}, worst normalised error {worst:.2e}, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
