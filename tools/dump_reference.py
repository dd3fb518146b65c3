#!/usr/bin/env python
"""Closes the "parity unpinned" gap on a machine that HAS the CUDA reference installed
(graphdeco-inria/gaussian-splatting with diff-gaussian-rasterization @ dr_aa, auxiliary.h patched 0.2f -> 0.05f as
/root/reference/README.md:33 instructs).  It renders this project's seeded synthetic inputs with the upstream
rasterizer and dumps every output in the fixture format of tests/golden/, so that

    python tools/dump_reference.py --out tests/golden/cuda_reference_config1.npz        # on the CUDA box
    python -m pytest tests/test_cuda_reference_fixture.py -m gpu                         # on the MI355X box

compares the HIP path with the true reference (RGB / inverse depth <= 1e-4 off the threshold-borderline pixels,
radii exact up to a stated handful).  Not runnable in the authoring container (no CUDA, no upstream source).

What is stored besides the outputs: a sha256 of EVERY input array (so that a drift of torch's RNG between the two
machines reads as "inputs differ", not as a parity failure -- or worse, as parity) and, for config1 (23 MB) or with
--embed-inputs, the input arrays themselves, which the test then uses instead of regenerating them.
``--backend oracle`` writes the same format from this project's CPU oracle: that exercises the dump / compare
pipeline and is labelled ``source = "oracle"`` -- the fixture test refuses to count it as a CUDA pin."""
import argparse
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsworld_amd import scenes  # noqa: E402  (pure torch/numpy: importable anywhere)

INPUT_KEYS = ("means3D", "shs", "opacities", "scales", "rotations", "viewmatrix", "projmatrix", "campos")


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()


def scene_of(config: str):
    if config == "config1":
        return scenes.random_scene_camera_frame(100_000, seed=0), scenes.identity_camera(256, 256, 60.0)
    return scenes.tabletop_scene("xarm6_align"), scenes.sensor_camera("xarm6_align")


def inputs_of(raw, cam) -> dict:
    means, shs, op, sc, rot = raw.activated()
    return dict(means3D=means.numpy(), shs=shs.numpy(), opacities=op.numpy().reshape(-1), scales=sc.numpy(),
                rotations=rot.numpy(), viewmatrix=cam.world_view_transform.numpy().reshape(-1),
                projmatrix=cam.full_proj_transform.numpy().reshape(-1), campos=cam.camera_center.numpy())


def render_cuda(inp, cam):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # the CUDA reference

    dev = "cuda"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    rs = GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
        bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=t(inp["viewmatrix"]).reshape(4, 4),
        projmatrix=t(inp["projmatrix"]).reshape(4, 4), sh_degree=3, campos=t(inp["campos"]), prefiltered=False,
        debug=False, antialiasing=False)
    means = t(inp["means3D"])
    color, radii, invdepth = GaussianRasterizer(rs)(
        means3D=means, means2D=torch.zeros_like(means), shs=t(inp["shs"]), opacities=t(inp["opacities"]).reshape(-1, 1),
        scales=t(inp["scales"]), rotations=t(inp["rotations"]))
    return color.cpu().numpy(), radii.cpu().numpy(), invdepth.cpu().numpy()


def render_oracle(inp, cam):
    from tests import helpers as hp

    o = hp.oracle_forward(inp, hp.oracle_settings(cam), np.zeros(3, np.float32), border_eps=0.0, border_eps_T=0.0)
    return o["color"], o["geom"]["radii"], o["invdepth"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--config", choices=["config1", "config2"], default="config1")
    ap.add_argument("--backend", choices=["cuda", "oracle"], default="cuda")
    ap.add_argument("--embed-inputs", action="store_true", help="store the input arrays too (default for config1)")
    args = ap.parse_args()
    raw, cam = scene_of(args.config)
    inp = inputs_of(raw, cam)
    color, radii, invdepth = (render_cuda if args.backend == "cuda" else render_oracle)(inp, cam)
    out = dict(config=args.config, source="cuda" if args.backend == "cuda" else "oracle", color=color, radii=radii,
               invdepth=invdepth, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, width=cam.image_width,
               height=cam.image_height, torch_version=torch.__version__)
    for k in INPUT_KEYS:
        out[f"sha256.{k}"] = sha(inp[k])
        if args.embed_inputs or args.config == "config1" or k in ("viewmatrix", "projmatrix", "campos"):
            out[f"input.{k}"] = np.ascontiguousarray(inp[k], dtype=np.float32)
    np.savez_compressed(args.out, **out)
    print("wrote", args.out, "source", out["source"], "visible", int((radii > 0).sum()))


if __name__ == "__main__":
    main()
