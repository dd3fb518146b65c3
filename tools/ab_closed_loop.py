#!/usr/bin/env python
"""configs[2] surrogate (1 reset + 200 steps x 2 cameras, moving wrist camera, hipGraph replay per step): the frames of
a step batched per launch (gsr_forward_batch) against a stream per frame, each with steps enqueued ahead (ensure=False: a
random-action rollout) and with the policy in the loop (ensure=True: frame k is waited for before step k + 1 is issued).
One JSON line per variant."""
import json
import math
import os
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
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    W, H = 640, 480
    if os.environ.get("CL_SCENE", "") == "arm":  # (the robot's Gaussians on the URDF's links instead of scattered clusters)
        rl = cl.xarm6_rollout()
        raw = scenes.arm_tabletop_scene(rl["link_scan"], rl["labels"], n=n, seed=1)
    else:
        raw = scenes.tabletop_scene("xarm6_align", n=n, seed=1)
    cams = {"right_cam": scenes.sensor_camera("xarm6_align", W, H),
            "wrist_cam": look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, W, H)}
    rollout = cl.xarm6_rollout()
    parts, actors = cl.xarm6_rollout_parts(rollout)
    ep_len = 200
    poses = list(cl.rollout_poses(rollout, len(actors), steps=ep_len + 1, seed=0, num_envs=E))
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
    only = os.environ.get("CL_ONLY")  # "batched,ensure" as 0/1, e.g. CL_ONLY=1,0: one variant (for traces)
    for batched in (True, False):
        if only and batched != bool(int(only.split(",")[0])):
            continue
        loop = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev, batched=batched, num_envs=E,
                                     block_cache=os.environ.get("CL_NO_CACHE", "") != "1",
                                     tile_reuse=os.environ.get("CL_NO_TILE_REUSE", "") != "1")
        if os.environ.get("CL_SET_STREAMS"):  # (A/B: sets of one image size in flight at once when a step has more than 8 frames)
            loop.multi.max_set_streams = int(os.environ["CL_SET_STREAMS"])
        if os.environ.get("CL_SET_FRAMES"):
            loop.multi.set_frames = int(os.environ["CL_SET_FRAMES"])
        loop.reset(*pinned[0])
        if os.environ.get("CL_EAGER", "") != "1":  # (A/B: every step's launches issued one by one instead of a graph replay)
            loop.capture()
        if os.environ.get("CL_NO_FAST_STAGE", "") == "1":  # (A/B: the step's host values through set_poses / set_cameras)
            loop._stage_fn = None
        if os.environ.get("CL_EAGER_WAITED", "") == "1":  # (A/B: a waited-for step issues its launches one by one as well)
            loop.eager_when_waited = True
        for ensure in (False, True):
            if only and "," in only and ensure != bool(int(only.split(",")[1])):
                continue
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for (M, s), w in zip(pinned, wrists):
                loop.step(M, s, cameras={"wrist_cam": w}, ensure=ensure)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(json.dumps({"batched": batched, "policy_in_loop": ensure, "num_envs": E,
                              "frames_per_s": (ep_len + 1) * len(cams) * E / dt, "steps_per_s": (ep_len + 1) / dt,
                              "overflow_frames": loop.overflow_frames(), "recovered_steps": loop.recovered_steps}), flush=True)
        del loop


if __name__ == "__main__":
    main()
