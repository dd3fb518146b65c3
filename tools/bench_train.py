#!/usr/bin/env python
"""BASELINE.json configs[4]: forward + backward (dL/dmeans, dscales/drot -> dcov, dSH, dopacity) on 500k Gaussians
at 800x800 with the 3DGS training loss 0.8 * L1 + 0.2 * (1 - fused_ssim) (lambda_dssim = 0.2,
/root/reference/gsworld/utils/gs_utils.py:96), through the drop-in modules exactly as upstream train.py calls them
(GaussianRasterizer autograd Function + fused_ssim).  Prints one JSON line (secondary metric; bench.py carries the
headline).  Usage: bench_train.py [--steps K] [--warmup W] [--num-gaussians N]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gsworld_amd", "dropin"))

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from fused_ssim import fused_ssim  # noqa: E402
from gsworld_amd.ssim import photometric_loss  # noqa: E402
from gsworld_amd import debug as dbg, scenes  # noqa: E402


def run(steps=50, warmup=5, num_gaussians=500_000, size=800, fused=False, device="cuda:0"):
    """One measurement of the training step; returns the JSON-able record (bench.py quotes it under `train_step`)."""
    dev = torch.device(device)
    S = size
    cam = scenes.training_camera(S, S, 60.0).to(dev)
    raw = scenes.random_scene_camera_frame(num_gaussians, seed=5).to(dev)
    tgt = scenes.random_scene_camera_frame(num_gaussians, seed=5).to(dev)
    gen = torch.Generator(device="cpu").manual_seed(6)
    tgt.xyz += (0.01 * torch.randn(tgt.xyz.shape, generator=gen)).to(dev)
    tgt.features_dc += (0.1 * torch.randn(tgt.features_dc.shape, generator=gen)).to(dev)
    bg = torch.zeros(3, device=dev)
    # A training loop draws another view every iteration: the step cycles through 8 cameras (view 0 = configs[4]'s
    # identity view, the others 3 cm off the axis looking at the same point), so that nothing the forward keeps from one
    # frame to the next on a recycled state buffer -- depth-sort splitters, placement cuts -- carries over.
    import math

    from gsworld_amd.camera import look_at_view
    cams = [cam] + [look_at_view([0.03 * math.cos(k * math.pi / 3.5), 0.03 * math.sin(k * math.pi / 3.5), 0.0],
                                 [0.0, 0.0, 3.0], [0.0, -1.0, 0.0], cam.FoVx, cam.FoVy, S, S).to(dev) for k in range(1, 8)]
    rasts = [GaussianRasterizer(GaussianRasterizationSettings(S, S, c.tanfovx, c.tanfovy, bg, 1.0, c.world_view_transform,
                                                              c.full_proj_transform, 3, c.camera_center, False, False,
                                                              False)) for c in cams]

    def render(r, grad, view=0):
        rast = rasts[view]
        params = [r.xyz, r.features_dc, r.features_rest, r.opacity, r.scaling, r.rotation]
        if grad:
            for p in params:
                p.requires_grad_(True)
                p.grad = None
        means2D = torch.zeros_like(r.xyz, requires_grad=grad)
        if fused:
            color, radii, invd = rast(means3D=r.xyz, means2D=means2D, shs=r.features_dc, shs_rest=r.features_rest,
                                      opacities=r.opacity, scales=r.scaling, rotations=r.rotation, param_space=7)
            return (color if grad else color.clamp(0, 1)), radii  # (the fused loss clamps on load)
        shs = torch.cat((r.features_dc, r.features_rest), dim=1)
        color, radii, invd = rast(means3D=r.xyz, means2D=means2D, shs=shs, opacities=torch.sigmoid(r.opacity),
                                  scales=torch.exp(r.scaling), rotations=torch.nn.functional.normalize(r.rotation))
        return color.clamp(0, 1), radii

    with torch.no_grad():
        gts = [render(tgt, False, v)[0].detach() for v in range(len(cams))]

    def step(view=0):
        img, radii = render(raw, True, view)
        gt = gts[view]
        if fused:  # clamp, L1, fused-ssim, their weights and the autograd of all of it: one node, three kernels
            loss = photometric_loss(img, gt, 0.2, clamp01=True)
        else:
            l1 = (img - gt).abs().mean()
            loss = 0.8 * l1 + 0.2 * (1.0 - fused_ssim(img[None], gt[None]))
        loss.backward()
        return loss, radii

    for k in range(warmup):
        loss, radii = step(k % len(cams))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        loss, radii = step(k % len(cams))
    host_dt = (time.perf_counter() - t0) / steps  # what the host needs to ISSUE a step (it never waits inside the loop)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    loss, radii = step(0)  # (view 0: the numbers below are configs[4]'s own view)
    # algorithmic bytes of the step: forward 48 N + 280 V + 64 R + 16 W H; backward: twice the forward's
    # instance traffic (128 R) + the gradient bytes 4 (3+3+1+3+6+48+3+4) V = 284 V + the image gradient read (16 W H)
    from gsworld_amd import _C
    with torch.no_grad():
        shs_ = torch.cat((raw.features_dc, raw.features_rest), dim=1)
        R = _C.rasterize_gaussians(bg, raw.xyz.detach(), torch.empty(0, device=dev), torch.sigmoid(raw.opacity.detach()),
                                   torch.exp(raw.scaling.detach()), torch.nn.functional.normalize(raw.rotation.detach()),
                                   1.0, torch.empty(0, device=dev), cam.world_view_transform, cam.full_proj_transform,
                                   cam.tanfovx, cam.tanfovy, S, S, shs_.detach(), 3, cam.camera_center, False, False,
                                   False)[0]
    N, V = num_gaussians, int((radii > 0).sum().item())
    b_alg = (48 * N + 280 * V + 64 * R + 16 * S * S) + (128 * R + 284 * V + 16 * S * S)
    finite = all(torch.isfinite(p.grad).all().item() for p in (raw.xyz, raw.features_dc, raw.features_rest,
                                                               raw.opacity, raw.scaling, raw.rotation))
    # the step's REAL HBM traffic, from the committed counter run of this very command (profiles/round*/pmc_train_step.json:
    # 2 x FETCH_SIZE + WRITE_SIZE per kernel + the memsets; tools/pmc_train_summary.py) -- beside the byte model's figure
    real = None
    if fused:
        import glob
        import json as _json

        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*", "pmc_train_step.json"))):
            rec_ = _json.load(open(path))
            real = {"hbm_bytes_per_step": rec_["hbm_bytes_per_step"], "real_hbm_GBs": rec_["hbm_bytes_per_step"] / dt / 1e9,
                    "frac_of_8TBs": rec_["hbm_bytes_per_step"] / dt / 1e9 / 8000.0,
                    "source": os.path.relpath(path, ROOT) + f" (collected {rec_['collected']}; counters of another run of this "
                                                            "command, this run's step time)"}
    return {
        "metric": "training iterations/sec (forward + backward, fused-ssim loss)", "value": 1.0 / dt,
        "unit": "it/s", "ms_per_step": dt * 1e3, "host_issue_ms_per_step": host_dt * 1e3, "steps": steps, "warmup": warmup, "dtype": "f32",
        "data": "synthetic",
        "roofline": {"bound": "hbm", "algorithmic_bytes_per_step": int(b_alg), "achieved": b_alg / dt / 1e9,
                     "peak": 8000.0, "unit": "GB/s", "frac": b_alg / dt / 1e9 / 8000.0,
                     "note": "whole step (forward + backward + loss) against the HBM peak; SURVEY.md 8d byte model",
                     "counters": real},
        "config": {"workload": f"{num_gaussians} Gaussians (config-1 distribution, seed 5), {S}x{S}, "
                               "loss 0.8*L1 + 0.2*(1-ssim), forward+backward, no optimizer step, another of 8 nearby views every "
                               "step as a training loop would draw them "
                               "(BASELINE.json configs[4])",
                   "parameter_packing": "fused (raw parameters, split SH; photometric_loss)" if fused else "upstream (torch)",
                   "num_visible": V, "num_rendered": int(R), "loss": float(loss.item()),
                   "grads_finite": bool(finite)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--num-gaussians", type=int, default=500_000)
    ap.add_argument("--size", type=int, default=800)
    ap.add_argument("--binning-mode", type=int, default=-1, help="A/B: 0 radix, 1 counting placement, 2 bin-then-sort")
    ap.add_argument("--fused", action="store_true",
                    help="raw parameters + split SH through the autograd Function (activations and their chain rule "
                         "inside the kernels) instead of upstream's torch packing")
    args = ap.parse_args()
    if args.binning_mode >= 0:
        dbg.set_binning_mode(args.binning_mode)
    print(json.dumps(run(args.steps, args.warmup, args.num_gaussians, args.size, args.fused)))


if __name__ == "__main__":
    main()
