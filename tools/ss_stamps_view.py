#!/usr/bin/env python
"""Cycle stamps of the depth sort's kernels (library built with -DGSR_SS_TIMING) for one view of the headline scene rendered
as the bench renders it (inference frames of the laid-out model).  usage: ss_stamps_view.py [sensor|dense] [frames]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsworld_amd import debug as dbg, scenes  # noqa: E402
from gsworld_amd._lib import check, lib  # noqa: E402
from gsworld_amd.layout import SceneLayout  # noqa: E402
from gsworld_amd.renderer import FrameRenderer  # noqa: E402

view = sys.argv[1] if len(sys.argv) > 1 else "dense"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
W, H = 640, 480
raw = scenes.tabletop_scene("xarm6_align", seed=1)
cam = (scenes.dense_view_camera("xarm6_align", W, H) if view == "dense" else scenes.sensor_camera("xarm6_align", W, H)).to(dev)
means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
L = SceneLayout.build(means, sc, rot, shs=shs, opacities=op)
a = L.arrays
Lb = lib()
Lb.gsr_debug_ss_stamps.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
for with_layout in (True, False):
    r = FrameRenderer(dev, forward_only=True, want_radii=False)
    for _ in range(frames):
        if with_layout:
            r.render(cam, a["means3D"], a["opacities"], shs=a["shs"], scales=a["scales"], rotations=a["rotations"], layout=L.layout)
        else:
            r.render(cam, means, op, shs=shs, scales=sc, rotations=rot)
    torch.cuda.synchronize()
    out = (C.c_uint64 * 64)()
    check(Lb.gsr_debug_ss_stamps(raw.num, W, -H, C.c_void_p(r.geom.data_ptr()), out))
    v = list(out)
    d = lambda x, y: int(v[y] - v[x])  # noqa: E731
    st = r.stats()
    print(f"{view} layout={with_layout}: V {st.num_visible} R {st.num_rendered} sort {dbg.sort_state(r.geom)}")
    print(f"   prepare: counts+sums {d(0, 1)} | scan {d(1, 22)} | decision {d(22, 23)} | total to decision {d(0, 23)}")
    print(f"   compact wg64: plan {d(8, 9)} | offsets {d(9, 4)} | walk {d(4, 5)} | classify {d(5, 6)} | table {d(6, 7)} | total {d(8, 7)}")
    print(f"   partition wg64: setup {int(v[17] - v[16])} move {int(v[18] - v[17])} | buckets wg100: hdr {int(v[33] - v[32])} "
          f"sort {int(v[35] - v[33])} ties+emit {int(v[36] - v[35])} bits {v[40]} n {v[41]}")
