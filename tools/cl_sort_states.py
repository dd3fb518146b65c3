#!/usr/bin/env python
"""What the depth sort of every camera did with its kept splitters, step by step, along the configs[2] surrogate
(gsworld_amd.debug.sort_state per lane after every step; synchronises: a diagnostic, not a measurement)."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsworld_amd import closed_loop as cl, debug as dbg, scenes  # noqa: E402
from gsworld_amd.camera import look_at_view  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else scenes.XARM6_ALIGN_NUM_GAUSSIANS
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
W, H = 640, 480
if os.environ.get("CL_SCENE", "") == "arm":  # (the arm-shaped surrogate: scenes.arm_tabletop_scene)
    _r = cl.xarm6_rollout()
    raw = scenes.arm_tabletop_scene(_r["link_scan"], _r["labels"], n=n, seed=1)
else:
    raw = scenes.tabletop_scene("xarm6_align", n=n, seed=1)
cams = {"right_cam": scenes.sensor_camera("xarm6_align", W, H),
        "wrist_cam": look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, W, H)}
rollout = cl.xarm6_rollout()
parts, actors = cl.xarm6_rollout_parts(rollout)
poses = list(cl.rollout_poses(rollout, len(actors), steps=steps + 1, seed=0, num_envs=1))
loop = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev, batched=True, num_envs=1)
loop.reset(*poses[0])
for k, (M, s) in enumerate(poses):
    a = 2.0 * math.pi * k / 200
    w = look_at_view([0.55 - 0.10 * math.sin(a), 0.35, 0.25 + 0.05 * math.sin(2.0 * a)], [0.35, 0.05, 0.05], [0, 0, 1],
                     0.9715089, 0.7551448, W, H)
    loop.step(M, s, cameras={"wrist_cam": w}, ensure=True)
    row = []
    for name, lane in zip(loop.names, loop.multi.lanes):
        st = dbg.sort_state(lane.geom)
        V = lane.stats().num_visible
        row.append(f"{name}: V={V} B={st['buckets']} blind={int(st['blind'])} near={int(st['near'])} fresh={int(st['fresh'])} "
                   f"bad={int(st['bad'])} trust={st['trust']}")
    print(k, " | ".join(row))
