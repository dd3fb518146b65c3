"""Renders a few config-2 frames (for rocprofv3 --pmc runs).  Usage: pmc_frames.py [variant] [frames] [blocks_per_cu]
PMC_BATCH=B in the environment: B frames per launch through gsr_forward_batch (bench.py's step), `frames` such steps."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsworld_amd import debug as dbg, scenes  # noqa: E402
from gsworld_amd._lib import check, lib  # noqa: E402
from gsworld_amd.renderer import FrameRenderer  # noqa: E402

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 4
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bpc = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda:0")
raw = scenes.tabletop_scene("xarm6_align")
cam = scenes.sensor_camera("xarm6_align").to(dev)
means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
from gsworld_amd.layout import SceneLayout  # noqa: E402

L_ = SceneLayout.build(means, sc, rot, shs=shs, opacities=op)  # the layout bench.py's headline renders with
a_ = L_.arrays
means, shs, op, sc, rot, lay = a_["means3D"], a_["shs"], a_["opacities"], a_["scales"], a_["rotations"], L_.layout
B = int(os.environ.get("PMC_BATCH", "1"))
L = lib()
dbg.set_render_variant(variant, bpc)
if B > 1:
    from gsworld_amd.renderer import MultiCameraRenderer  # noqa: E402

    mc = MultiCameraRenderer(B, dev, batched=True, forward_only=True, want_radii=False)  # bench.py's step
    for _ in range(frames):
        mc.render([cam] * B, means, op, shs=shs, scales=sc, rotations=rot, layout=lay)
    torch.cuda.synchronize()
    print("stats", mc.lanes[0].stats())
else:
    r = FrameRenderer(dev, forward_only=True, want_radii=False)  # the frame bench.py times (inference frame)
    for _ in range(frames):
        r.render(cam, means, op, shs=shs, scales=sc, rotations=rot, layout=lay)
    torch.cuda.synchronize()
    print("stats", r.stats())
