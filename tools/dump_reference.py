#!/usr/bin/env python
"""Closes the "parity unpinned" gap on a machine that HAS the CUDA reference installed
(graphdeco-inria/gaussian-splatting with diff-gaussian-rasterization @ dr_aa, auxiliary.h patched 0.2f -> 0.05f as
/root/reference/README.md:33 instructs).  It renders this project's seeded synthetic inputs with the upstream
rasterizer and dumps inputs + every output in the fixture format of tests/golden/, so that

    python tools/dump_reference.py --out tests/golden/cuda_reference_config1.npz        # on the CUDA box
    python -m pytest tests/test_cuda_reference_fixture.py -m gpu                         # on the MI355X box

compares the HIP path with the true reference (RGB / inverse depth <= 1e-4, radii exact).  Not runnable in the
authoring container (no CUDA, no upstream source): it is shipped for the maintainer who can run it.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsworld_amd import scenes  # noqa: E402  (pure torch/numpy: importable anywhere)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--config", choices=["config1", "config2"], default="config1")
    args = ap.parse_args()
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # the CUDA reference

    if args.config == "config1":
        raw, cam = scenes.random_scene_camera_frame(100_000, seed=0), scenes.identity_camera(256, 256, 60.0)
    else:
        raw, cam = scenes.tabletop_scene("xarm6_align"), scenes.sensor_camera("xarm6_align")
    dev = "cuda"
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    c = cam.to(dev)
    bg = torch.zeros(3, device=dev)
    rs = GaussianRasterizationSettings(
        image_height=c.image_height, image_width=c.image_width, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=bg,
        scale_modifier=1.0, viewmatrix=c.world_view_transform, projmatrix=c.full_proj_transform, sh_degree=3,
        campos=c.camera_center, prefiltered=False, debug=False, antialiasing=False)
    color, radii, invdepth = GaussianRasterizer(rs)(means3D=means, means2D=torch.zeros_like(means), shs=shs,
                                                    opacities=op, scales=sc, rotations=rot)
    np.savez_compressed(
        args.out, config=args.config, color=color.cpu().numpy(), radii=radii.cpu().numpy(),
        invdepth=invdepth.cpu().numpy(), tanfovx=c.tanfovx, tanfovy=c.tanfovy,
        viewmatrix=c.world_view_transform.cpu().numpy(), projmatrix=c.full_proj_transform.cpu().numpy(),
        campos=c.camera_center.cpu().numpy())
    print("wrote", args.out, "visible", int((radii > 0).sum()))


if __name__ == "__main__":
    main()
