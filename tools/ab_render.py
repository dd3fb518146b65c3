"""Within-process interleaved A/B of library variants on the config-2 frame (stage times from HIP events)."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsworld_amd import scenes  # noqa: E402
from gsworld_amd._lib import GsrProfile, PROFILE_STAGES, check, lib  # noqa: E402
from gsworld_amd.renderer import FrameRenderer  # noqa: E402

dev = torch.device("cuda:0")
raw = scenes.tabletop_scene("xarm6_align")
cam = scenes.sensor_camera("xarm6_align").to(dev)
means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
r = FrameRenderer(dev)
L = lib()
L.gsr_debug_set_render_variant.argtypes = [C.c_int]


def frame():
    return r.render(cam, means, op, shs=shs, scales=sc, rotations=rot)


frame(); frame(); torch.cuda.synchronize()
ref = None
res = {}
for rnd in range(5):
    for variant in (0, 1):
        check(L.gsr_debug_set_render_variant(variant))
        frame(); torch.cuda.synchronize()
        check(L.gsr_profile_enable(2))
        for _ in range(50):
            color, _, _ = frame()
        torch.cuda.synchronize()
        p = GsrProfile(); check(L.gsr_profile_collect(p)); check(L.gsr_profile_enable(0))
        ms = [p.stage_ms[k] / p.frames for k in range(5)]
        res.setdefault(variant, []).append(ms)
        if ref is None:
            ref = color.clone()
        else:
            d = float((color - ref).abs().max())
            assert d < 1e-5, f"variant {variant} differs from variant 0 by {d}"
for v, rows in res.items():
    med = [sorted(x[k] for x in rows)[len(rows) // 2] for k in range(5)]
    print(f"variant {v}: " + ", ".join(f"{n}={m * 1e3:.1f}us" for n, m in zip(PROFILE_STAGES, med)) +
          f"  total={sum(med) * 1e3:.1f}us")
print(json.dumps({str(v): rows for v, rows in res.items()}))
