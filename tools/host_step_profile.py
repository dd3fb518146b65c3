#!/usr/bin/env python
"""Where the HOST's time goes in a closed-loop step with the policy in the loop (configs[2] surrogate, 1 environment):
wall time per step, the device's kernel span from the step's events, and a cProfile of step().  Usage:
host_step_profile.py [num_gaussians]"""
import cProfile
import math
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsworld_amd import closed_loop as cl, scenes  # noqa: E402
from gsworld_amd.camera import look_at_view  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else scenes.XARM6_ALIGN_NUM_GAUSSIANS
    W, H = 640, 480
    raw = scenes.tabletop_scene("xarm6_align", n=n, seed=1)
    cams = {"right_cam": scenes.sensor_camera("xarm6_align", W, H),
            "wrist_cam": look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, W, H)}
    rollout = cl.xarm6_rollout()
    parts, actors = cl.xarm6_rollout_parts(rollout)
    ep_len = 200
    poses = list(cl.rollout_poses(rollout, len(actors), steps=ep_len + 1, seed=0, num_envs=1))
    pinned = [(M.pin_memory(), s.pin_memory()) for M, s in poses]

    def wrist_at(k):
        a = 2.0 * math.pi * k / ep_len
        v = look_at_view([0.55 - 0.10 * math.sin(a), 0.35, 0.25 + 0.05 * math.sin(2.0 * a)], [0.35, 0.05, 0.05], [0, 0, 1],
                         0.9715089, 0.7551448, W, H)
        v.world_view_transform = v.world_view_transform.pin_memory()
        v.full_proj_transform = v.full_proj_transform.pin_memory()
        v.camera_center = v.camera_center.pin_memory()
        return v

    wrists = [wrist_at(k) for k in range(ep_len + 1)]
    loop = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev, batched=True, num_envs=1)
    loop.reset(*pinned[0])
    loop.capture()
    for rep in range(2):
        torch.cuda.synchronize()
        t_sync = t_issue = 0.0
        t0 = time.perf_counter()
        for (M, s), w in zip(pinned, wrists):
            a = time.perf_counter()
            loop.step(M, s, cameras={"wrist_cam": w}, ensure=False)
            b = time.perf_counter()
            torch.cuda.current_stream(dev).synchronize()
            c = time.perf_counter()
            t_issue += b - a
            t_sync += c - b
        dt = time.perf_counter() - t0
        print(f"policy in the loop: {dt / (ep_len + 1) * 1e6:.1f} us per step = issue {t_issue / (ep_len + 1) * 1e6:.1f} "
              f"(step() returns) + wait {t_sync / (ep_len + 1) * 1e6:.1f}", flush=True)
    # the same with the graph route forced (what ensure=True takes)
    loop.eager_when_ahead = False
    for rep in range(2):
        torch.cuda.synchronize()
        t_sync = t_issue = 0.0
        t0 = time.perf_counter()
        for (M, s), w in zip(pinned, wrists):
            a = time.perf_counter()
            loop.step(M, s, cameras={"wrist_cam": w}, ensure=False)
            b = time.perf_counter()
            torch.cuda.current_stream(dev).synchronize()
            c = time.perf_counter()
            t_issue += b - a
            t_sync += c - b
        dt = time.perf_counter() - t0
        print(f"graph route: {dt / (ep_len + 1) * 1e6:.1f} us per step = issue {t_issue / (ep_len + 1) * 1e6:.1f} + wait "
              f"{t_sync / (ep_len + 1) * 1e6:.1f}", flush=True)
    pr = cProfile.Profile()
    pr.enable()
    for (M, s), w in zip(pinned, wrists):
        loop.step(M, s, cameras={"wrist_cam": w}, ensure=True)
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(28)


if __name__ == "__main__":
    main()
