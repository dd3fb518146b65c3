#!/usr/bin/env python
"""ManiSkill-free surrogate of BASELINE.json configs[2] (AlignXArmEnv-v1 random-action rollout, ep_len = 200):
1 reset + 200 steps x 2 cameras = 402 frames of the xarm6_align-like scene, with the per-step work GSWorldWrapper does
around the rasterizer (/root/reference/gsworld/mani_skill/utils/wrappers/gs_world_wrapper.py:176-198 step/reset,
:110-162 per-link transforms, :232-275 per-camera render + uint8 conversion).  SAPIEN / PhysX are not available on
a headless MI355X box, so the robot motion is a seeded random walk of 16 link poses + 2 tracked actors; what is
reproduced is the RENDER-SIDE workload, not the physics (SURVEY.md section 7 "hard parts").

Two glue variants around the same HIP rasterizer:
  --glue reference : what the wrapper does, op for op, in torch on the GPU: deepcopy of the model per step and per
                     camera-frame, 18 isin() masks + gathers + transform_gaussians per step, 36 isin() + masked
                     scatters per camera-frame, upstream render() activations (sigmoid / exp / normalize / cat).
  --glue fused     : gsworld_amd.transform.FusedPartTransform (one pass) + FrameRenderer
                     (persistent state, no host sync) + pack_rgb8.
Prints one JSON line.
"""
import argparse
import copy
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsworld_amd import scenes, transform as tf  # noqa: E402
from gsworld_amd.camera import extract_rigid_transform, look_at_view  # noqa: E402
from gsworld_amd._lib import RAW_ROTATIONS  # noqa: E402
from gsworld_amd.renderer import MultiCameraRenderer  # noqa: E402


def small_rigid(gen, k, angle=0.05, shift=0.01):
    """k random small rigid 4x4 increments."""
    w = torch.randn(k, 3, generator=gen) * angle
    th = w.norm(dim=1, keepdim=True).clamp_min(1e-8)
    a = w / th
    K = torch.zeros(k, 3, 3)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0] = -a[:, 2], a[:, 1], a[:, 2]
    K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -a[:, 0], -a[:, 1], a[:, 0]
    R = torch.eye(3) + torch.sin(th)[:, :, None] * K + (1 - torch.cos(th))[:, :, None] * (K @ K)
    M = torch.eye(4).repeat(k, 1, 1)
    M[:, :3, :3] = R
    M[:, :3, 3] = torch.randn(k, 3, generator=gen) * shift
    return M


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--glue", choices=["fused", "reference"], default="fused")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--num-gaussians", type=int, default=scenes.XARM6_ALIGN_NUM_GAUSSIANS)
    ap.add_argument("--graph", action="store_true", help="fused glue: replay one hipGraph per simulation step")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    raw = scenes.tabletop_scene("xarm6_align", n=args.num_gaussians, seed=1).to(dev)
    cams = {"right_cam": scenes.sensor_camera("xarm6_align").to(dev),
            "wrist_cam": look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, 640,
                                      480).to(dev)}
    sim2gs = torch.tensor(scenes.SIM2GS_XARM_TRANS)
    sim2gs_inv = torch.linalg.inv(sim2gs)
    # 16 robot links (labels 1..16) and 2 tracked actors (labels 17, 18), as xarm_gs_semantics / obj_gs_semantics do
    parts = {f"link{k}": k for k in range(1, 17)}
    parts.update({"005_tomato_soup_can": 17, "dtc_green_can": 18})
    K = len(parts)
    gen = torch.Generator().manual_seed(0)
    link_now = torch.eye(4).repeat(K, 1, 1)
    bg = torch.zeros(3, device=dev)
    multi = MultiCameraRenderer(len(cams), dev)
    obs = {n: torch.empty((1, 480, 640, 3), dtype=torch.uint8, device=dev) for n in cams}

    # static activations (fused path): only xyz / rotation change per step
    raw.features_dc, raw.features_rest = raw.features_dc.contiguous(), raw.features_rest.contiguous()
    opac = torch.sigmoid(raw.opacity)
    scl = torch.exp(raw.scaling)
    op = tf.FusedPartTransform(parts, raw.semantics)
    model = types.SimpleNamespace(_xyz=raw.xyz, _scaling=raw.scaling, _rotation=raw.rotation, _opacity=raw.opacity,
                                  _semantics=raw.semantics, _features_dc=raw.features_dc,
                                  _features_rest=raw.features_rest)
    t_glue = t_render = 0.0

    def part_matrices():
        # sim2gs @ link_now @ inv(link_scan = I) @ inv(sim2gs); actors also carry a uniform scale
        full = sim2gs @ link_now @ sim2gs_inv
        rigid, scale, _, _ = extract_rigid_transform(full)
        scales = torch.ones(K)
        scales[-2:] = scale[-2:] * torch.tensor([1.0, 1.0])
        return rigid, scales

    # device-resident pose buffers: what a GPU simulator hands over (ManiSkill link poses are device tensors)
    M_dev = torch.eye(4, device=dev).repeat(K, 1, 1).contiguous()
    S_dev = torch.ones(K, device=dev)
    RING = 8  # pinned staging slots: the host may run several steps ahead of the GPU
    M_pin = [torch.empty((K, 4, 4), pin_memory=True) for _ in range(RING)]
    S_pin = [torch.empty((K,), pin_memory=True) for _ in range(RING)]
    slot_free = [None] * RING
    step_no = 0
    step_graph = None

    def gpu_step():
        """Everything the GPU does per step: pose table, fused transform, quaternion normalisation, both cameras."""
        xyz, rot = op.apply(raw.xyz, raw.rotation, M_dev, S_dev)
        # the transformed quaternions keep their norm (reference semantics); preprocess normalises them on load
        multi.render(list(cams.values()), xyz, opac, rgb8_out=[obs[n][0] for n in cams], shs=raw.features_dc,
                     shs_rest=raw.features_rest, scales=scl, rotations=rot, param_space=RAW_ROTATIONS, bg=bg)

    def step_fused():
        nonlocal t_glue, t_render, step_no
        t0 = time.perf_counter()
        M, scales = part_matrices()
        s = step_no % RING
        step_no += 1
        if slot_free[s] is not None:
            slot_free[s].synchronize()  # the copy that last used this staging slot has been consumed
        M_pin[s].copy_(M)
        S_pin[s].copy_(scales)
        M_dev.copy_(M_pin[s], non_blocking=True)
        S_dev.copy_(S_pin[s], non_blocking=True)
        slot_free[s] = torch.cuda.Event()
        slot_free[s].record()
        t1 = time.perf_counter()
        if step_graph is not None:
            step_graph.replay()  # one launch per simulation step
        else:
            gpu_step()
        t2 = time.perf_counter()
        t_glue += t1 - t0
        t_render += t2 - t1

    def step_reference():
        nonlocal t_glue, t_render
        from gsworld_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

        t0 = time.perf_counter()
        M, scales = part_matrices()
        Md = M.to(dev)
        moved = {}
        splats = copy.deepcopy(model)  # gs_world_wrapper.py:112
        for k, (name, lab) in enumerate(parts.items()):
            target = torch.tensor([lab], device=dev)
            mask = torch.isin(splats._semantics.long().squeeze(-1), target.long())
            sc = None if k < 16 else scales[k].to(dev)
            moved[name] = tf.transform_gaussians(splats, torch.where(mask)[0], scale=sc, rot_mat=Md[k:k + 1, :3, :3],
                                                 translation=Md[k:k + 1, :3, 3])
        t1 = time.perf_counter()
        for name, cam in cams.items():
            gs = copy.deepcopy(model)  # :244
            for pname, lab in parts.items():
                for attr, val in zip(("_xyz", "_scaling", "_rotation", "_opacity"), moved[pname]):
                    if val.shape[0] == 1:  # the wrapper's `shape[0] == num_envs` test, num_envs = 1
                        m = torch.isin(gs._semantics.long().squeeze(-1), torch.tensor([lab], device=dev).long())
                        getattr(gs, attr)[m] = val[0]
            rs = GaussianRasterizationSettings(480, 640, cam.tanfovx, cam.tanfovy, bg, 1.0, cam.world_view_transform,
                                               cam.full_proj_transform, 3, cam.camera_center, False, False, False)
            means2D = torch.zeros_like(gs._xyz, requires_grad=True) + 0
            color, _, _ = GaussianRasterizer(rs)(
                means3D=gs._xyz, means2D=means2D, shs=torch.cat((gs._features_dc, gs._features_rest), dim=1),
                opacities=torch.sigmoid(gs._opacity), scales=torch.exp(gs._scaling),
                rotations=torch.nn.functional.normalize(gs._rotation))
            img = color.clamp(0, 1).permute(1, 2, 0).unsqueeze(0)
            obs[name] = (img * 255).clamp(0, 255).to(torch.uint8)  # :268-270
        t2 = time.perf_counter()
        t_glue += t1 - t0
        t_render += t2 - t1

    step = step_fused if args.glue == "fused" else step_reference
    step()  # reset() renders once
    torch.cuda.synchronize()
    if args.glue == "fused" and args.graph:
        # the exact-mode frames above sized every lane's binning capacity; capture the whole GPU side of a step
        step()
        multi.ensure_valid(gpu_step)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            gpu_step()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            gpu_step()
        step_graph = g
        torch.cuda.synchronize()
    t_glue = t_render = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        link_now = link_now @ small_rigid(gen, K)
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    frames = args.steps * len(cams)
    print(json.dumps({
        "metric": "closed-loop rendered frames/sec (surrogate of AlignXArmEnv-v1 rand-action rollout)",
        "value": frames / dt, "unit": "frames/s", "steps_per_s": args.steps / dt, "glue": args.glue,
        "launch": "hipGraph replay per step" if step_graph is not None else "eager",
        "config": {"workload": f"{args.num_gaussians} Gaussians, 2 cameras 640x480, {args.steps} steps, 18 moving parts",
                   "host_ms_per_step": {"transform_glue": 1e3 * t_glue / args.steps,
                                        "render_enqueue": 1e3 * t_render / args.steps},
                   "checksum": int(obs["right_cam"].sum().item())}}))


if __name__ == "__main__":
    main()
