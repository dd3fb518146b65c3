#!/usr/bin/env python
"""Per-frame cost of the drop-in ``gaussian_renderer.render`` exactly as GSWorld calls it
(gs_world_wrapper.py:266-267) on the config-2 scene: frozen parameters (inference fast path: SH read as stored) vs
upstream's packing (activations + ``cat(dc, rest)`` + autograd Function).  Prints one JSON line."""
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for sub in ("dropin", "gs_compat"):
    sys.path.insert(0, os.path.join(ROOT, "gsworld_amd", sub))

from arguments import PipelineParams  # noqa: E402
from gaussian_renderer import render  # noqa: E402
from scene.cameras import Camera  # noqa: E402
from scene.gaussian_model import GaussianModel  # noqa: E402

from gsworld_amd import scenes  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    raw = scenes.tabletop_scene("xarm6_align").to(dev)
    pc = GaussianModel(3)
    pc._xyz, pc._features_dc, pc._features_rest = raw.xyz, raw.features_dc.contiguous(), raw.features_rest.contiguous()
    pc._opacity, pc._scaling, pc._rotation = raw.opacity[..., None], raw.scaling, raw.rotation
    pc.active_sh_degree = 3
    ref = scenes.sensor_camera("xarm6_align")
    W2C = ref.world_view_transform.T
    cam = Camera(resolution=(640, 480), colmap_id=0, R=W2C[:3, :3].T.numpy(), T=W2C[:3, 3].numpy(), FoVx=ref.FoVx,
                 FoVy=ref.FoVy, depth_params=None, image=None, invdepthmap=None, image_name="right_cam", uid=0,
                 data_device=dev)
    pipe = PipelineParams().extract(types.SimpleNamespace())
    bg = torch.zeros(3, device=dev)

    def timed(n=100):
        for _ in range(5):
            render(cam, pc, pipe, bg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = render(cam, pc, pipe, bg)["render"]
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, out.detach().clone()

    fast_ms, a = timed()
    pipe.fused_activations = True  # opt-in: raw parameters, sigmoid / exp / normalize inside preprocess
    fused_ms, c = timed()
    pipe.fused_activations = False
    pc._xyz.requires_grad_(True)  # any trainable parameter -> upstream's path
    full_ms, b = timed()
    print(json.dumps({"metric": "drop-in gaussian_renderer.render ms/frame @640x480, 1.5M Gaussians",
                      "frozen_parameters_fast_path_ms": fast_ms, "fused_activations_ms": fused_ms,
                      "upstream_packing_autograd_path_ms": full_ms,
                      "images_bit_identical": bool(torch.equal(a, b)),
                      "fused_activations_pixels_above_1e-5": float(((c - a).abs() > 1e-5).float().mean())}))


if __name__ == "__main__":
    main()
