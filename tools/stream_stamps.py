"""Where and when every quadrant of the stream compositor ran (libraries built with -DGSR_STREAM_STAMPS=1; see
render.hip).  Renders the headline frame one at a time, reads the per-quadrant stamps of the LAST frame out of the image
state and writes them to an .npz; prints the kernel span, how evenly the SIMDs finish and how many waves are alive over
the kernel's life.

    python tools/stream_stamps.py out.npz [render_blocks_per_cu] [--view dense]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsworld_amd import _lib, scenes  # noqa: E402
from gsworld_amd.renderer import FrameRenderer  # noqa: E402

out_path = sys.argv[1]
bpc = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 0
view = "dense" if "--view" in sys.argv and sys.argv[sys.argv.index("--view") + 1] == "dense" else "sensor"
_lib.TUNING["render_blocks_per_cu"] = bpc
dev = torch.device("cuda:0")
W, H = 640, 480
raw = scenes.tabletop_scene("xarm6_align")
cam = (scenes.sensor_camera("xarm6_align", W, H) if view == "sensor" else scenes.dense_view_camera("xarm6_align", W, H)).to(dev)
means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
bg = torch.zeros(3, device=dev)
r = FrameRenderer(dev, forward_only=True, want_radii=False)
fr = lambda: r.render(cam, means, op, shs=shs, scales=sc, rotations=rot, bg=bg)  # noqa: E731
for _ in range(3):
    fr()
    r.ensure_valid(fr)
for _ in range(20):
    fr()
torch.cuda.synchronize()
st = r.stats()
import ctypes as C  # noqa: E402

sv = _lib.GsrStateView()
_lib.check(_lib.lib().gsr_state_view(raw.num, W, H, C.c_int64(0), None, None, C.c_void_p(r.image.data_ptr()), C.byref(sv)))
T = ((W + 15) // 16) * ((H + 15) // 16)
off = int(sv.final_T) - r.image.data_ptr()
s = r.image[off:off + 64 * T].view(torch.int32).cpu().numpy().view(np.uint32).reshape(4 * T, 4)
np.savez_compressed(out_path, stamps=s, bpc=bpc)
t0, t1, hw, work = s[:, 0].astype(np.int64), s[:, 1].astype(np.int64), s[:, 2], s[:, 3] & 0x7FFFFFFF
queued = (s[:, 3] >> 31) != 0
base = t0.min()
t0 -= base; t1 -= base
t0 %= 1 << 32; t1 %= 1 << 32
span = t1.max()
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = hw >> 28
key = ((xcc.astype(np.int64) * 8 + se) * 2 + sh) * 64 + cu * 4 + simd
uniq, inv = np.unique(key, return_inverse=True)
last = np.zeros(len(uniq), np.int64); np.maximum.at(last, inv, t1)
cnt = np.bincount(inv); wsum = np.bincount(inv, weights=work)
print(f"bpc={bpc} units={len(s)} queued={int(queued.sum())} span={span} ticks; SIMDs seen {len(uniq)}; units per SIMD "
      f"min/mean/max {cnt.min()}/{cnt.mean():.2f}/{cnt.max()}")
print(f" SIMD finish time: mean {last.mean():.0f} p10 {np.percentile(last, 10):.0f} p50 {np.percentile(last, 50):.0f} "
      f"p90 {np.percentile(last, 90):.0f} max {last.max()}  (ticks; span {span})")
print(f" work per SIMD: mean {wsum.mean():.0f} p10 {np.percentile(wsum, 10):.0f} p90 {np.percentile(wsum, 90):.0f} max {wsum.max():.0f};"
      f" corr(work, finish) {np.corrcoef(wsum, last)[0, 1]:.3f}")
print(f" unit life: mean {np.mean(t1 - t0):.0f} ticks; unit end: mean {t1.mean():.0f}; start: max {t0.max()}")
# waves alive over time, chip-wide
edges = np.linspace(0, span, 11)
alive = [(int(((t0 <= e) & (t1 > e)).sum())) for e in edges[:-1]]
print(" waves alive at 0,10,..90 % of the span:", alive)
# rate of a unit: work per tick against how many units share its SIMD on average over its life
rate = work / np.maximum(t1 - t0, 1)
print(f" work per tick of a unit: mean {rate.mean():.3f} p10 {np.percentile(rate, 10):.3f} p90 {np.percentile(rate, 90):.3f}")
