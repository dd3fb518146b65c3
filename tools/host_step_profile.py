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


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "parts"):
    main()


def parts():
    """Every host-side piece of a policy-in-the-loop step timed by itself (us per call)."""
    import numpy as np

    dev = torch.device("cuda:0")
    W, H = 640, 480
    raw = scenes.tabletop_scene("xarm6_align", n=200_000, seed=1)
    cams = {"right_cam": scenes.sensor_camera("xarm6_align", W, H),
            "wrist_cam": look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, W, H)}
    rollout = cl.xarm6_rollout()
    parts_, actors = cl.xarm6_rollout_parts(rollout)
    poses = list(cl.rollout_poses(rollout, len(actors), steps=4, seed=0, num_envs=1))
    M, s = poses[1][0].pin_memory(), poses[1][1].pin_memory()
    w = look_at_view([0.5, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, W, H)
    loop = cl.ClosedLoopRenderer(raw, parts_, cams, scaled_parts=actors, device=dev, batched=True, num_envs=1)
    loop.reset(M, s)
    loop.capture()
    loop.step(M, s, cameras={"wrist_cam": w}, ensure=True)

    def t(name, fn, n=2000):
        fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        print(f"{name:34s} {(time.perf_counter() - t0) / n * 1e6:7.2f} us", flush=True)

    stream = torch.cuda.current_stream(dev)
    t("set_poses", lambda: loop.set_poses(M, s))
    t("set_cameras (one camera)", lambda: loop.set_cameras({"wrist_cam": w}))
    t("ring copy (numpy)", lambda: np.copyto(loop._ring_np[0], loop._host_np))
    t("torch.cuda.current_stream", lambda: torch.cuda.current_stream(dev))
    t("stream.synchronize (idle)", lambda: stream.synchronize())
    t("_check_overflow", lambda: loop._check_overflow(late=False))
    t("_slot_acquire + release(waited)", lambda: loop._slot_release(loop._slot_acquire(), True))
    g = loop._graphs[0]

    def replay_and_wait():
        g.replay()
        stream.synchronize()
    t("graph replay + wait (a whole step's device time inside)", replay_and_wait, 300)
    t0 = time.perf_counter()
    for _ in range(300):
        g.replay()
    t1 = time.perf_counter()
    stream.synchronize()
    print(f"{'graph.replay() alone (issue)':34s} {(t1 - t0) / 300 * 1e6:7.2f} us")
    t("step(ensure=True) whole", lambda: loop.step(M, s, cameras={"wrist_cam": w}, ensure=True), 300)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "parts":
    parts()
