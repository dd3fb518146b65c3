#!/usr/bin/env python
"""A camera that creeps (millimetres, a fraction of a degree per frame), stops, jumps and creeps again over a static scene:
every inference frame against the exact-mode frame of a fresh renderer, bit for bit, whatever the depth sort did with its
kept splitter table (checked against samples / taken unchecked under the same view / taken unchecked under a view that moved
a little / drawn anew).  Usage: soak_moving_camera.py [frames] [num_gaussians] [seed]"""
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsworld_amd import debug as dbg, scenes  # noqa: E402
from gsworld_amd.camera import look_at_view  # noqa: E402
from gsworld_amd.renderer import FrameRenderer  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 400_000
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    dev = torch.device("cuda:0")
    raw = scenes.tabletop_scene("xarm6_align", n=n, seed=40 + seed)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    kw = dict(shs=shs, scales=sc, rotations=rot)
    gen = torch.Generator().manual_seed(seed)
    r = FrameRenderer(dev, forward_only=True, want_radii=False, min_capacity=1 << 26)
    exact = FrameRenderer(dev)
    eye = [0.55, 0.35, 0.25]
    phase = 0.0
    tally = dict(near=0, same=0, checked=0, fresh=0, bad=0)
    t0 = time.time()
    for k in range(frames):
        u = float(torch.rand((), generator=gen))
        if k % 97 == 96:       # a jump to another pose
            eye = [0.35 + 0.4 * float(torch.rand((), generator=gen)), 0.15 + 0.4 * float(torch.rand((), generator=gen)),
                   0.2 + 0.3 * float(torch.rand((), generator=gen))]
        elif k % 40 >= 30:     # rests
            pass
        else:                  # creeps: up to 3 mm per frame
            phase += 0.03 * u
            eye = [eye[0] + 0.003 * math.cos(phase) * u, eye[1] + 0.003 * math.sin(phase) * u, eye[2] + 0.001 * math.sin(3 * phase)]
        cam = look_at_view(eye, [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, 640, 480).to(dev)
        got = r.render(cam, means, op, **kw)[0].clone()
        st = dbg.sort_state(r.geom)
        want = exact.render(cam, means, op, exact=True, **kw)[0]
        if not torch.equal(got, want):
            print(f"MISMATCH at frame {k}: {st}")
            sys.exit(1)
        tally["near"] += int(st["near"])
        tally["same"] += int(st["blind"] and not st["near"])
        tally["fresh"] += int(st["fresh"])
        tally["checked"] += int(not st["blind"] and not st["fresh"])
        tally["bad"] += int(st["bad"])
    print(f"moving-camera soak: {frames} frames bit-identical to exact-mode frames ({n} Gaussians, seed {seed}); kept table taken "
          f"unchecked under a view that moved a little {tally['near']}, under the same view {tally['same']}, checked and kept "
          f"{tally['checked']}, drawn anew {tally['fresh']}; frames with a bucket above its bound {tally['bad']}; {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
