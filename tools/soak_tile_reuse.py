#!/usr/bin/env python
"""Soak of the block cache + tile reuse (csrc/preprocess.hip prep_block_cached, csrc/render.hip "tile reuse"): per step a random
subset of the parts moves (sometimes none, sometimes all), the wrist camera rests or moves, now and then the background changes;
every frame of every environment is compared, byte for byte, with the frame of a loop built with block_cache=False.  Eager for
the first half, under graph replay for the second.  usage: soak_tile_reuse.py [steps] [num_gaussians] [num_envs] [seed]"""
import json
import math
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsworld_amd import closed_loop as cl, debug as dbg, scenes  # noqa: E402
from gsworld_amd.camera import look_at_view  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 400_000
    E = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 7
    dev = torch.device("cuda:0")
    raw = scenes.tabletop_scene("xarm6_align", n=n, seed=seed)
    rollout = cl.xarm6_rollout()
    parts, actors = cl.xarm6_rollout_parts(rollout)
    cams = {"right_cam": scenes.sensor_camera("xarm6_align"),
            "wrist_cam": look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, 640, 480)}
    walk = list(cl.rollout_poses(rollout, len(actors), steps=steps, seed=seed, num_envs=E))
    rng = random.Random(seed)
    K = walk[0][0].shape[-3]

    def wrist(a):
        return look_at_view([0.55 - 0.10 * math.sin(a), 0.35, 0.25 + 0.05 * math.sin(2.0 * a)], [0.35, 0.05, 0.05], [0, 0, 1],
                            0.9715089, 0.7551448, 640, 480)

    reuse = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev, num_envs=E)
    plain = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev, num_envs=E, block_cache=False)
    M, s = walk[0][0].clone(), walk[0][1].clone()
    for lp in (reuse, plain):
        lp.reset(M, s)
    angle, kept_tiles, kept_blocks, mismatches = 0.0, 0, 0, 0
    for k in range(1, steps):
        if k == steps // 2:
            for lp in (reuse, plain):
                lp.capture()
        mode = rng.random()
        moving = [] if mode < 0.2 else (range(K) if mode > 0.85 else rng.sample(range(K), rng.randint(1, 4)))
        for p in moving:
            M[..., p, :, :], s[..., p] = walk[k][0][..., p, :, :], walk[k][1][..., p]
        if rng.random() < 0.3:
            angle += 0.05
        if rng.random() < 0.05:
            bg = torch.tensor([rng.random(), rng.random(), rng.random()], device=dev)
            for lp in (reuse, plain):
                lp.bg.copy_(bg)
        got = reuse.step(M.clone(), s.clone(), cameras={"wrist_cam": wrist(angle)}, ensure=True)
        want = plain.step(M.clone(), s.clone(), cameras={"wrist_cam": wrist(angle)}, ensure=True)
        mismatches += sum(0 if torch.equal(got[c], want[c]) else 1 for c in cams)
        for lane in reuse.multi.lanes:
            st = dbg.sort_state(lane.geom)
            kept_tiles += int(st["kept_tiles"])
            kept_blocks += int(st["kept_blocks"])
    print(json.dumps({"soak": "tile_reuse", "steps": steps - 1, "num_gaussians": n, "num_envs": E, "frames": (steps - 1) * 2 * E,
                      "frames_that_differ_from_the_loop_that_keeps_nothing": mismatches, "frames_that_kept_blocks": kept_blocks,
                      "frames_that_left_tiles": kept_tiles, "overflow_frames": reuse.overflow_frames()}))
    return 1 if mismatches else 0


if __name__ == "__main__":
    sys.exit(main())
