#!/usr/bin/env python
"""The two sibling ops alone, for `rocprofv3 --kernel-trace --stats` (neither
had profile evidence): `knn` = gsr_knn_dist2 on the 1.47 M means of the headline scene, 20 calls; `ssim` = the fused_ssim
drop-in forward + backward on a (1,3,800,800) image pair, 50 steps (configs[4]'s loss)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gsworld_amd", "dropin"))
from gsworld_amd import scenes  # noqa: E402

dev = torch.device("cuda:0")
what = sys.argv[1] if len(sys.argv) > 1 else "knn"
if what == "knn":
    from gsworld_amd.knn import distCUDA2

    raw = scenes.tabletop_scene("xarm6_align")
    pts = raw.xyz.to(dev).contiguous()
    out = torch.empty((raw.num,), dtype=torch.float32, device=dev)
    for _ in range(20):
        distCUDA2(pts, out=out)
    torch.cuda.synchronize()
    print("knn ok", float(out.mean()))
else:
    from fused_ssim import fused_ssim

    g = torch.Generator(device="cpu").manual_seed(0)
    a = torch.rand((1, 3, 800, 800), generator=g).to(dev).requires_grad_(True)
    b = torch.rand((1, 3, 800, 800), generator=g).to(dev)
    for _ in range(50):
        a.grad = None
        s = fused_ssim(a, b)
        s.backward()
    torch.cuda.synchronize()
    print("ssim ok", float(s))
