#!/usr/bin/env python
"""ManiSkill-free surrogate of BASELINE.json configs[2] (AlignXArmEnv-v1 random-action rollout, ep_len = 200):
1 reset + 200 steps x 2 cameras = 402 frames of the xarm6_align-like scene, with the per-step work GSWorldWrapper does
around the rasterizer (/root/reference/gsworld/mani_skill/utils/wrappers/gs_world_wrapper.py:176-198 step/reset,
:110-162 per-link transforms, :232-275 per-camera render + uint8 conversion).  SAPIEN / PhysX are not available on
a headless MI355X box, so the robot motion is forward kinematics of the reference's xarm6 URDF along a seeded
random-action trajectory (15 moving links: tests/golden/xarm6_rollout.npz, gsworld_amd.closed_loop.rollout_poses) and the
2 tracked actors random-walk; what is reproduced is the RENDER-SIDE workload, not the physics.

Two glue variants around the same HIP rasterizer, same pose sequence (their frames agree within 1 LSB:
tests/test_closed_loop_gpu.py):
  --glue fused     : gsworld_amd.closed_loop.ClosedLoopRenderer (device-side pose table, one fused transform pass, all
                     frames of a step in flight, --graph: one hipGraph replay per step).
  --glue reference : what the wrapper does, op for op, in torch on the GPU (oracle/wrapper_glue_ref.py -- checker code,
                     timed here only to put a number on the glue the fused path removes).
Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsworld_amd import closed_loop as cl, scenes  # noqa: E402
from gsworld_amd.camera import look_at_view  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--glue", choices=["fused", "reference"], default="fused")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--num-gaussians", type=int, default=scenes.XARM6_ALIGN_NUM_GAUSSIANS)
    ap.add_argument("--num-envs", type=int, default=1)
    ap.add_argument("--graph", action="store_true", help="fused glue: replay one hipGraph per simulation step")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    raw = scenes.tabletop_scene("xarm6_align", n=args.num_gaussians, seed=1)
    cams = {"right_cam": scenes.sensor_camera("xarm6_align"),
            "wrist_cam": look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, 640, 480)}
    rollout = cl.xarm6_rollout()
    parts, actors = cl.xarm6_rollout_parts(rollout)
    E = args.num_envs
    poses = cl.rollout_poses(rollout, len(actors), steps=args.steps + 1, seed=0, num_envs=E)
    t_pose = t_gpu = 0.0

    if args.glue == "fused":
        loop = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, num_envs=E, device=dev)
        loop.reset(*next(poses))
        if args.graph:
            loop.capture()
        torch.cuda.synchronize()
        RING = 8  # pinned staging slots: the host may run several steps ahead of the GPU
        pin = [(torch.empty(tuple(loop.matrices.shape), pin_memory=True),
                torch.empty(tuple(loop.scales.shape), pin_memory=True)) for _ in range(RING)]
        free = [None] * RING
        t0 = time.perf_counter()
        for i in range(args.steps):
            ta = time.perf_counter()
            M, s = next(poses)
            slot = i % RING
            if free[slot] is not None:
                free[slot].synchronize()  # the copy that last used this staging slot has been consumed
            pin[slot][0].copy_(M)
            pin[slot][1].copy_(s)
            loop.set_poses(pin[slot][0], pin[slot][1])
            free[slot] = torch.cuda.Event()
            free[slot].record()
            tb = time.perf_counter()
            frames = loop.step()
            tc = time.perf_counter()
            t_pose += tb - ta
            t_gpu += tc - tb
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        overflow = any(st.overflow for st in loop.ensure_valid())
        launch = "hipGraph replay per step" if args.graph else "eager"
    else:
        from gsworld_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
        from oracle import wrapper_glue_ref as ref  # measurement of the reference-style glue only

        rawd = raw.to(dev)
        model = types.SimpleNamespace(_xyz=rawd.xyz, _scaling=rawd.scaling, _rotation=rawd.rotation,
                                      _opacity=rawd.opacity.reshape(-1, 1, 1), _semantics=rawd.semantics,
                                      _features_dc=rawd.features_dc, _features_rest=rawd.features_rest)
        cams_d = {k: v.to(dev) for k, v in cams.items()}

        def rasterize(view, means3D, shs, opacities, scales, rotations, bg):
            rs = GaussianRasterizationSettings(view.image_height, view.image_width, view.tanfovx, view.tanfovy, bg, 1.0,
                                               view.world_view_transform, view.full_proj_transform, 3,
                                               view.camera_center, False, False, False)
            return GaussianRasterizer(rs)(means3D=means3D, means2D=torch.zeros_like(means3D), shs=shs,
                                          opacities=opacities, scales=scales, rotations=rotations)[0]

        frames = ref.render_step(model, parts, cams_d, *next(poses), rasterize, actors)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            M, s = next(poses)
            frames = ref.render_step(model, parts, cams_d, M, s, rasterize, actors)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        overflow, launch = False, "eager"
    n_frames = args.steps * len(cams) * E
    print(json.dumps({
        "metric": "closed-loop rendered frames/sec (surrogate of AlignXArmEnv-v1 rand-action rollout)",
        "value": n_frames / dt, "unit": "frames/s", "steps_per_s": args.steps / dt, "glue": args.glue,
        "launch": launch,
        "config": {"workload": f"{args.num_gaussians} Gaussians, {len(cams)} cameras 640x480, {E} env(s), "
                               f"{args.steps} steps, {len(parts)} moving parts",
                   "host_ms_per_step": {"poses": 1e3 * t_pose / max(args.steps, 1),
                                        "gpu_enqueue": 1e3 * t_gpu / max(args.steps, 1)},
                   "overflow": overflow, "checksum": int(frames["right_cam"].sum().item())}}))


if __name__ == "__main__":
    main()
