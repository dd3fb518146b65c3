#!/usr/bin/env python
"""Randomised check of the fused-parameter training path (raw opacity / scale / rotation + split SH through the autograd
Function) against upstream's torch packing on the GPU: gradients of every parameter within 3e-3 of the largest entry.
Usage: fuzz_fused_backward.py [iterations] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsworld_amd import scenes  # noqa: E402
from gsworld_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda:0")
    names = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")
    worst = 0.0
    for it in range(iters):
        n = int(rng.choice([50, 2_000, 15_000]))
        W, H = int(rng.integers(17, 200)), int(rng.integers(17, 160))
        aa = bool(rng.random() < 0.4)
        deg = int(rng.integers(1, 4))
        cam = scenes.training_camera(W, H, float(rng.uniform(30, 100))).to(dev)
        raw = scenes.random_scene_camera_frame(n, seed=int(rng.integers(1 << 30))).to(dev)
        raw.scaling += float(rng.uniform(-0.5, 2.0))
        raw.rotation *= float(rng.uniform(0.2, 5.0))  # un-normalised quaternions of any length
        bg = torch.tensor(rng.random(3), dtype=torch.float32, device=dev)
        rs = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, bg, float(rng.choice([1.0, 0.7])),
                                           cam.world_view_transform, cam.full_proj_transform, deg, cam.camera_center,
                                           False, False, aa)
        rast = GaussianRasterizer(rs)
        w_img = torch.randn((3, H, W), device=dev)
        w_dep = torch.randn((1, H, W), device=dev)

        def run(fused):
            ps = [getattr(raw, k).detach().clone().requires_grad_(True) for k in names]
            xyz, dc, rest, op, sc, rot = ps
            m2d = torch.zeros_like(xyz, requires_grad=True)
            if fused:
                color, _, invd = rast(means3D=xyz, means2D=m2d, shs=dc, shs_rest=rest, opacities=op, scales=sc,
                                      rotations=rot, param_space=7)
            else:
                color, _, invd = rast(means3D=xyz, means2D=m2d, shs=torch.cat((dc, rest), dim=1),
                                      opacities=torch.sigmoid(op), scales=torch.exp(sc),
                                      rotations=torch.nn.functional.normalize(rot))
            ((color * w_img).sum() + (invd * w_dep).sum()).backward()
            return [p.grad for p in ps] + [m2d.grad]

        g0, g1 = run(False), run(True)
        for k, a, b in zip(names + ("means2D",), g0, g1):
            err = float((a - b).abs().max()) / (float(a.abs().max()) + 1e-12)
            worst = max(worst, err)
            assert err <= 3e-3 and torch.isfinite(b).all(), (it, k, err, n, W, H, aa, deg)
    print(f"fuzz fused backward ok: {iters} cases, worst normalised difference {worst:.2e}")


if __name__ == "__main__":
    main()
