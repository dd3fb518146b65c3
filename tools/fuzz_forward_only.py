#!/usr/bin/env python
"""Randomised sweep on the GPU: inference frames (GsrSettings.forward_only: super-tile binning, tile-rect test in the
compositor, no backward-only writes, lean state) against default frames -- colour, inverse depth, uint8 frame and radii
must be the same BITS -- over random model sizes, image shapes (odd tile grids included), splat scales, opacities, SH
degrees, antialiasing, scale modifiers, near planes, parameter spaces and both frame paths (exact / capacity).
Usage: fuzz_forward_only.py [iterations] [seed].  Prints one summary line; exits non-zero on the first mismatch."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsworld_amd import scenes  # noqa: E402
from gsworld_amd._lib import RAW_OPACITY, RAW_ROTATIONS, RAW_SCALES  # noqa: E402
from gsworld_amd.renderer import FrameRenderer  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    t0, inst_ratio = time.time(), []
    for it in range(iters):
        n = int(rng.choice([1, 7, 300, 5_000, 20_000, 60_000, 150_000]))
        w, h = int(rng.integers(1, 900)), int(rng.integers(1, 700))
        raw = scenes.random_scene_camera_frame(n, seed=int(rng.integers(1 << 30)), near_fraction=float(rng.uniform(0, 0.3)))
        raw.scaling += float(rng.uniform(-1.5, 2.5))
        if rng.random() < 0.3:
            raw.scaling[:, 0] += float(rng.uniform(0, 3))
        if rng.random() < 0.3:
            raw.opacity -= float(rng.uniform(0, 4))
        if rng.random() < 0.2:
            raw.xyz[:, :2] *= float(rng.uniform(1, 5))
        cam = scenes.identity_camera(w, h, float(rng.uniform(25, 110))).to(dev)
        kw = dict(sh_degree=int(rng.integers(0, 4)), antialiasing=bool(rng.random() < 0.3),
                  scale_modifier=float(rng.choice([1.0, 0.6, 1.7])),
                  bg=torch.from_numpy(rng.random(3).astype(np.float32)).to(dev))
        near = float(rng.choice([0.05, 0.2]))
        r_ = raw.to(dev)
        if rng.random() < 0.4:  # raw parameters + split SH
            args = (r_.xyz, r_.opacity)
            kw.update(shs=r_.features_dc, shs_rest=r_.features_rest, scales=r_.scaling, rotations=r_.rotation,
                      param_space=RAW_OPACITY | RAW_SCALES | RAW_ROTATIONS)
        else:
            means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
            args = (means, op)
            kw.update(shs=shs, scales=sc, rotations=rot)
        full = FrameRenderer(dev, near_plane=near)
        fast = FrameRenderer(dev, near_plane=near, forward_only=True)
        f8 = [torch.zeros((h, w, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
        for k in range(2):  # frame 0 exact, frame 1 on the capacity path
            a = full.render(cam, *args, rgb8_out=f8[0], **kw)
            b = fast.render(cam, *args, rgb8_out=f8[1], **kw)
            torch.cuda.synchronize()
            ok = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(f8[0], f8[1])
            if not ok:
                print(f"MISMATCH at iteration {it} frame {k}: n={n} {w}x{h} kw={ {x: kw[x] for x in ('sh_degree', 'antialiasing', 'scale_modifier')} }")
                sys.exit(1)
        sa, sb = full.ensure_valid(lambda: None), fast.ensure_valid(lambda: None)
        assert not sa.overflow and not sb.overflow and sb.num_visible <= sa.num_visible
        if sa.num_rendered:
            inst_ratio.append(sb.num_rendered / sa.num_rendered)
    print(f"forward_only fuzz: {iters} cases x 2 frames bit-identical (seed {seed}), super-tile instances / tile instances: "
          f"median {np.median(inst_ratio):.2f} min {min(inst_ratio):.2f} max {max(inst_ratio):.2f}, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
