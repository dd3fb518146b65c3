#!/usr/bin/env python
"""Randomised sweep on the GPU: B frames through ONE gsr_forward_batch call (MultiCameraRenderer(batched=True): sets of up
to 8 frames per launch) against the same B frames through one gsr_forward call each -- colour, inverse depth, uint8 frame
and radii must be the same BITS, and V / R the same counts -- over random model sizes, B = 1 ... 19, image shapes (odd tile
grids included), cameras that differ per frame (field of view, a turn about the view axis, a shift), splat scales, SH
degrees, antialiasing, scale modifiers, inference and default frames, with and without a load-time layout, and four
consecutive steps each (exact frame, capacity path, kept splitters, blind splitters with the halved bucket count).
Usage: fuzz_batch.py [iterations] [seed].  Prints one summary line; exits non-zero on the first mismatch."""
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsworld_amd import layout as gl, scenes  # noqa: E402
from gsworld_amd.renderer import FrameRenderer, MultiCameraRenderer  # noqa: E402


def _turned(cam, angle, shift):
    """The camera turned by `angle` about its view axis and moved by `shift` in its image plane (row-vector convention of
    the 3DGS cameras: world_view_transform is the TRANSPOSED view matrix)."""
    c, s = math.cos(angle), math.sin(angle)
    T = torch.eye(4)
    T[0, 0], T[0, 1], T[1, 0], T[1, 1] = c, s, -s, c
    T[3, 0], T[3, 1] = shift
    out = type(cam)(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in vars(cam).items()})
    proj = torch.linalg.solve(cam.world_view_transform.double(), cam.full_proj_transform.double()).float()
    out.world_view_transform = cam.world_view_transform @ T
    out.full_proj_transform = out.world_view_transform @ proj
    out.camera_center = torch.linalg.inv(out.world_view_transform)[3, :3].contiguous()
    return out


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    t0, frames = time.time(), 0
    for it in range(iters):
        n = int(rng.choice([1, 300, 5_000, 20_000, 60_000, 150_000, 400_000, 1_000_000], p=[.1, .1, .15, .15, .15, .15, .12, .08]))
        B = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 11, 16, 19])) if n < 1_000_000 else int(rng.choice([2, 3, 9]))
        w, h = int(rng.integers(1, 700)), int(rng.integers(1, 500))
        raw = scenes.random_scene_camera_frame(n, seed=int(rng.integers(1 << 30)), near_fraction=float(rng.uniform(0, 0.3)))
        raw.scaling += float(rng.uniform(-1.5, 2.0))
        if rng.random() < 0.3:
            raw.opacity -= float(rng.uniform(0, 4))
        means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
        kw = dict(shs=shs, scales=sc, rotations=rot)
        if rng.random() < 0.5 and n >= 300:
            L = gl.SceneLayout.build(means, sc, rot, shs=shs, opacities=op)
            a = L.arrays
            means, op = a["means3D"], a["opacities"]
            kw = dict(shs=a["shs"], scales=a["scales"], rotations=a["rotations"], layout=L.layout)
        kw.update(sh_degree=int(rng.integers(0, 4)), antialiasing=bool(rng.random() < 0.3),
                  scale_modifier=float(rng.choice([1.0, 0.6, 1.7])),
                  bg=torch.from_numpy(rng.random(3).astype(np.float32)).to(dev))
        base = scenes.identity_camera(w, h, float(rng.uniform(25, 110)))
        cams = [_turned(base, float(rng.uniform(-0.4, 0.4)), (float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.3, 0.3)))).to(dev)
                if k and rng.random() < 0.8 else base.to(dev) for k in range(B)]
        rkw = dict(forward_only=bool("layout" in kw or rng.random() < 0.6), want_radii=True)  # (a permuted model needs inference frames)
        singles = [FrameRenderer(dev, **rkw) for _ in cams]
        mc = MultiCameraRenderer(B, dev, batched=True, **rkw)
        f1 = [torch.zeros((h, w, 3), dtype=torch.uint8, device=dev) for _ in cams]
        fb = [torch.zeros((h, w, 3), dtype=torch.uint8, device=dev) for _ in cams]
        for step in range(4):
            want = [r.render(c, means, op, rgb8_out=f, **kw) for r, c, f in zip(singles, cams, f1)]
            for r, c, f in zip(singles, cams, f1):
                r.ensure_valid(lambda r=r, c=c, f=f: r.render(c, means, op, rgb8_out=f, **kw))
            want = [tuple(t.clone() for t in x) for x in want] if step == 0 else want
            got = mc.render(cams, means, op, rgb8_out=fb, **kw)
            stats = mc.ensure_valid(lambda: mc.render(cams, means, op, rgb8_out=fb, **kw))
            torch.cuda.synchronize()
            for k in range(B):
                s1 = singles[k].stats()
                ok = (all(torch.equal(x, y) for x, y in zip(got[k], want[k])) and torch.equal(fb[k], f1[k]) and
                      (stats[k].num_visible, stats[k].num_rendered) == (s1.num_visible, s1.num_rendered) and
                      not stats[k].overflow and not s1.overflow)
                if not ok:
                    print(f"MISMATCH at iteration {it} step {step} frame {k} of {B}: n={n} {w}x{h} "
                          f"forward_only={rkw['forward_only']} layout={'layout' in kw} V {stats[k].num_visible} / {s1.num_visible} "
                          f"R {stats[k].num_rendered} / {s1.num_rendered}")
                    sys.exit(1)
            frames += B
    print(f"batch fuzz: {iters} cases, {frames} frames bit-identical to one gsr_forward call each (seed {seed}), "
          f"{time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
