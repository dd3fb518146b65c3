"""Within-process interleaved A/B of compositing-kernel variants on the config-2 frame (stage times from HIP
events recorded inside libgsr_hip.so).  Usage: ab_render.py "0,0 3,6 4,6"  (variant,blocks_per_cu pairs)"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsworld_amd import debug as dbg, scenes  # noqa: E402
from gsworld_amd._lib import GsrProfile, PROFILE_STAGES, check, lib  # noqa: E402
from gsworld_amd.renderer import FrameRenderer  # noqa: E402

configs = [tuple(int(x) for x in c.split(",")) for c in (sys.argv[1] if len(sys.argv) > 1 else "0,0 3,6 4,6").split()]
dev = torch.device("cuda:0")
raw = scenes.tabletop_scene("xarm6_align")
cam = scenes.sensor_camera("xarm6_align").to(dev)
means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
r = FrameRenderer(dev)
L = lib()


def frame():
    return r.render(cam, means, op, shs=shs, scales=sc, rotations=rot)


frame(); frame(); torch.cuda.synchronize()
ref = None
res = {}
for rnd in range(5):
    for cfg in configs:
        dbg.set_render_variant(*cfg)
        frame(); torch.cuda.synchronize()
        check(L.gsr_profile_enable(2))
        for _ in range(50):
            color, _, _ = frame()
        torch.cuda.synchronize()
        p = GsrProfile(); check(L.gsr_profile_collect(p)); check(L.gsr_profile_enable(0))
        res.setdefault(cfg, []).append([p.stage_ms[k] / p.frames for k in range(5)])
        if ref is None:
            ref = color.clone()
        else:
            d = float((color - ref).abs().max())
            assert d < 1e-5, f"config {cfg} differs from the first config by {d}"
for cfg, rows in res.items():
    med = [sorted(x[k] for x in rows)[len(rows) // 2] for k in range(5)]
    print(f"variant {cfg}: " + ", ".join(f"{n}={m * 1e3:.1f}us" for n, m in zip(PROFILE_STAGES, med)) +
          f"  total={sum(med) * 1e3:.1f}us")
print(json.dumps({str(k): v for k, v in res.items()}))
