#!/usr/bin/env python
"""Soak test: the headline scene rendered N times on three renderer states in flight; every frame (colour, radii,
uint8 frame) must be the same bytes as the first one.  Catches rare races in the paths that keep state from frame to
frame (splitters, placement cuts, quadrant deal, cooperative quadrants).  Usage: soak_static_scene.py [frames] [inference];
SOAK_VIEW=dense in the environment: the camera above the table (V = 0.6 N) instead of right_cam."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsworld_amd import scenes  # noqa: E402
from gsworld_amd.renderer import FrameRenderer  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
    infer = len(sys.argv) > 2 and sys.argv[2] == "inference"
    dev = torch.device("cuda:0")
    raw = scenes.tabletop_scene("xarm6_align")
    cam = (scenes.dense_view_camera("xarm6_align") if os.environ.get("SOAK_VIEW") == "dense"
           else scenes.sensor_camera("xarm6_align")).to(dev)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    lay = None
    if infer:
        from gsworld_amd.layout import SceneLayout

        L = SceneLayout.build(means, sc, rot, shs=shs, opacities=op)
        a = L.arrays
        means, shs, op, sc, rot, lay = a["means3D"], a["shs"], a["opacities"], a["scales"], a["rotations"], L.layout
    mk = (lambda: FrameRenderer(dev, forward_only=True, want_radii=False)) if infer else (lambda: FrameRenderer(dev))
    lanes = [(mk(), torch.cuda.Stream(dev), torch.empty((480, 640, 3), dtype=torch.uint8, device=dev))
             for _ in range(3)]
    ref = None
    bad = 0
    for k in range(n):
        r, st, frame = lanes[k % 3]
        with torch.cuda.stream(st):
            color, radii, _ = r.render(cam, means, op, shs=shs, scales=sc, rotations=rot, rgb8_out=frame, layout=lay)
            if k < 6 or k % 97 == 0 or k >= n - 3:
                r.ensure_valid(lambda: r.render(cam, means, op, shs=shs, scales=sc, rotations=rot, rgb8_out=frame, layout=lay))
                st.synchronize()
                snap = (color.clone(), frame.clone()) if radii is None else (color.clone(), radii.clone(), frame.clone())
                if ref is None:
                    ref = snap
                elif not all(torch.equal(a, b) for a, b in zip(snap, ref)):
                    bad += 1
                    print(f"frame {k}: differs from frame 0", flush=True)
    torch.cuda.synchronize()
    print(f"soak ({'inference frames, laid-out model' if infer else 'default frames'}): {n} frames, {bad} mismatching snapshots")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
