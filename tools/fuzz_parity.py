#!/usr/bin/env python
"""Randomised parity sweep on the GPU: HIP path vs oracle (tests/helpers.compare_forward: indices and preprocess floats
bit-exact, image <= 1e-4 off borderline pixels) over random sizes, image shapes, splat scales, SH degrees, options and
parameter spaces, plus bit-identity of the plain tile compositor (variant 0) with the default one on every case.
Usage: fuzz_parity.py [iterations] [seed].  Prints one summary line; exits non-zero on the first mismatch."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsworld_amd import debug as dbg, scenes  # noqa: E402
from gsworld_amd._lib import check, lib  # noqa: E402
from oracle import gs_oracle as go  # noqa: E402
from tests import helpers as hp  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    L = lib()
    t0 = time.time()
    worst_rgb, total_border, total_px = 0.0, 0, 0
    for it in range(iters):
        n = int(rng.choice([1, 7, 300, 5_000, 20_000, 60_000]))
        w, h = int(rng.integers(1, 700)), int(rng.integers(1, 500))
        raw = scenes.random_scene_camera_frame(n, seed=int(rng.integers(1 << 30)), near_fraction=float(rng.uniform(0, 0.3)))
        raw.scaling += float(rng.uniform(-1.5, 2.5))
        if rng.random() < 0.3:
            raw.scaling[:, 0] += float(rng.uniform(0, 3))  # anisotropic
        if rng.random() < 0.3:
            raw.opacity -= float(rng.uniform(0, 4))  # many near the 1/255 threshold
        if rng.random() < 0.2:
            raw.xyz[:, :2] *= float(rng.uniform(1, 5))
        cam = scenes.identity_camera(w, h, float(rng.uniform(25, 110)))
        kw = dict(sh_degree=int(rng.integers(0, 4)), antialiasing=bool(rng.random() < 0.3),
                  scale_modifier=float(rng.choice([1.0, 0.6, 1.7])), near_plane=float(rng.choice([0.05, 0.2])))
        inp, st = hp.np_inputs(raw, cam), hp.oracle_settings(cam, **kw)
        bg = rng.random(3).astype(np.float32)
        param_space = 0
        gin = inp
        if rng.random() < 0.3:  # raw parameter space: oracle runs on the canonical activations
            op, sc, ro = go.activate_params(raw.opacity.numpy().reshape(-1), raw.scaling.numpy(), raw.rotation.numpy(), 7)
            inp = dict(inp, opacities=op, scales=sc, rotations=ro)
            gin = dict(inp, opacities=raw.opacity.numpy().reshape(-1), scales=raw.scaling.numpy(),
                       rotations=raw.rotation.numpy())
            param_space = 7
        o = hp.oracle_forward(inp, st, bg)
        g = hp.gpu_forward(gin, st, bg, param_space=param_space)
        rep = hp.compare_forward(o, g, st)
        dbg.set_render_variant(0, 0)
        try:
            g0 = hp.gpu_forward(gin, st, bg, param_space=param_space)
        finally:
            dbg.set_render_variant(4, 0)
        for name in ("color", "invdepth"):
            assert np.array_equal(g[name].view(np.uint32), g0[name].view(np.uint32)), (it, name)
        if n > 0 and "views" in g:
            for name in ("final_T", "n_contrib"):
                assert np.array_equal(g["views"][name].view(np.uint32), g0["views"][name].view(np.uint32)), (it, name)
        worst_rgb = max(worst_rgb, rep.get("rgb_max_abs", 0.0))
        total_border += rep.get("borderline_pixels", 0)
        total_px += w * h
    print(f"fuzz ok: {iters} cases, seed {seed}, worst rgb {worst_rgb:.2e} off borderline pixels, "
          f"{total_border}/{total_px} borderline, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
