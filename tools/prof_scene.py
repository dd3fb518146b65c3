#!/usr/bin/env python
"""Renders a few frames of the headline scene eagerly (for rocprofv3 --kernel-trace --stats / --pmc runs).
usage: prof_scene.py [--view sensor|dense] [--frames N] [--default-mode] [--moving]"""
import argparse
import math
import time
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsworld_amd import scenes  # noqa: E402
from gsworld_amd.renderer import FrameRenderer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--view", default="sensor")
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--default-mode", action="store_true", help="forward_only = 0 (the training-capable frame)")
ap.add_argument("--moving", action="store_true", help="turn the camera a little on every frame")
ap.add_argument("--states", action="store_true", help="print what the depth sort did with its kept table, frame by frame (synchronises)")
ap.add_argument("--sh-degree", type=int, default=3, help="ablation: evaluate fewer SH bands (0: the DC term only)")
ap.add_argument("--no-layout", action="store_true", help="the model as given (no Morton order / block culling)")
args = ap.parse_args()
dev = torch.device("cuda:0")
raw = scenes.tabletop_scene("xarm6_align")
cam0 = scenes.dense_view_camera("xarm6_align") if args.view == "dense" else scenes.sensor_camera("xarm6_align")
cam = cam0.to(dev)
means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
lay = None
if not args.no_layout and not args.default_mode:  # (a permuted model needs inference frames; bench.py's headline has both)
    from gsworld_amd.layout import SceneLayout

    L_ = SceneLayout.build(means, sc, rot, shs=shs, opacities=op)
    a_ = L_.arrays
    means, shs, op, sc, rot, lay = a_["means3D"], a_["shs"], a_["opacities"], a_["scales"], a_["rotations"], L_.layout
r = FrameRenderer(dev, forward_only=not args.default_mode, want_radii=args.default_mode)
rgb8 = torch.zeros((480, 640, 3), dtype=torch.uint8, device=dev)
proj = cam0.world_view_transform.inverse() @ cam0.full_proj_transform
for k in range(args.frames):
    if args.moving:
        a = math.radians(2.0) * math.sin(2.0 * math.pi * k / 97.0)
        Rz = torch.tensor([[math.cos(a), -math.sin(a), 0, 0], [math.sin(a), math.cos(a), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
        wvt = Rz @ cam0.world_view_transform
        cam.world_view_transform.copy_(wvt)
        cam.full_proj_transform.copy_(wvt @ proj)
        cam.camera_center.copy_(wvt.inverse()[3, :3])
    r.render(cam, means, op, shs=shs, scales=sc, rotations=rot, rgb8_out=rgb8, sh_degree=args.sh_degree, layout=lay)
    if k == 1:
        r.ensure_valid(lambda: None)
        torch.cuda.synchronize()
        t_start = time.perf_counter()
    if args.states:
        from gsworld_amd import debug as dbg
        st = dbg.sort_state(r.geom)
        print(k, "B", st["buckets"], "blind", int(st["blind"]), "near", int(st["near"]), "fresh", int(st["fresh"]), "bad", int(st["bad"]),
              "trust", st["trust"])
torch.cuda.synchronize()
fps_line = f"{(args.frames - 2) / (time.perf_counter() - t_start):.0f} frames/s one at a time (eager launches)"
print("stats", r.stats())
print(fps_line)
