"""Debug: how many tiles does the tile reuse leave, per camera and step? (marker byte trick of tests/test_closed_loop_gpu.py)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsworld_amd import closed_loop as cl, debug as dbg, scenes  # noqa: E402
from gsworld_amd.camera import look_at_view  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 34
rollout = cl.xarm6_rollout()
if os.environ.get("CL_SCENE", "") == "arm":
    raw = scenes.arm_tabletop_scene(rollout["link_scan"], rollout["labels"], n=n, seed=seed)
else:
    raw = scenes.tabletop_scene("xarm6_align", n=n, seed=seed)
parts, actors = cl.xarm6_rollout_parts(rollout)
cams = {"right_cam": scenes.sensor_camera("xarm6_align"),
        "wrist_cam": look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, 640, 480)}
poses = list(cl.rollout_poses(rollout, len(actors), steps=12, seed=5))
for captured in (False, True):
    lp = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev)
    lp.reset(*poses[0])
    if captured:
        lp.capture()
    for k in range(1, 12):
        for nm in cams:
            lp.frames[nm].fill_(7)
        got = lp.step(*poses[k], ensure=True)
        row = []
        for nm, lane in zip(lp.names, lp.multi.lanes):
            g = got[nm][0]
            H, W = g.shape[:2]
            t = g.reshape(H // 16, 16, W // 16, 16, 3).permute(0, 2, 1, 3, 4).reshape(H // 16, W // 16, -1)
            left = int((t == 7).all(dim=2).sum())
            st = dbg.sort_state(lane.geom)
            row.append(f"{nm}: left {left}/{t.shape[0] * t.shape[1]} kept_blocks={st['kept_blocks']} kept_tiles={st['kept_tiles']}")
        print(f"captured={captured} step {k}: " + " | ".join(row), flush=True)
