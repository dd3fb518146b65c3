#!/usr/bin/env python
"""Cycle stamps of the depth sort's kernels (library built with -DGSR_SS_TIMING: tools/build_variants.sh depthsort.hip
sstiming:"-DGSR_SS_TIMING") for the two frames of a closed-loop step (static right_cam over a moving arm, moving
wrist_cam), workgroup 64 of each kernel, after `steps` steps of the FK rollout."""
import ctypes as C
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsworld_amd import closed_loop as cl, debug as dbg, scenes  # noqa: E402
from gsworld_amd._lib import check, lib  # noqa: E402
from gsworld_amd.camera import look_at_view  # noqa: E402

dev = torch.device("cuda:0")
W, H = 640, 480
raw = scenes.tabletop_scene("xarm6_align", seed=1)
cams = {"right_cam": scenes.sensor_camera("xarm6_align", W, H),
        "wrist_cam": look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, W, H)}
rollout = cl.xarm6_rollout()
parts, actors = cl.xarm6_rollout_parts(rollout)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
poses = list(cl.rollout_poses(rollout, len(actors), steps=steps + 1, seed=0))
loop = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev)
loop.reset(*poses[0])
L = lib()
L.gsr_debug_ss_stamps.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
for k, (M, s) in enumerate(poses[1:]):
    a = 2.0 * math.pi * k / 200
    w = look_at_view([0.55 - 0.10 * math.sin(a), 0.35, 0.25 + 0.05 * math.sin(2.0 * a)], [0.35, 0.05, 0.05], [0, 0, 1],
                     0.9715089, 0.7551448, W, H)
    loop.step(M, s, cameras={"wrist_cam": w}, ensure=True)
    if k % 10 == 9 or k == steps - 1:
        for name, lane in zip(loop.names, loop.multi.lanes):
            out = (C.c_uint64 * 64)()
            check(L.gsr_debug_ss_stamps(raw.num, W, -H, C.c_void_p(lane.geom.data_ptr()), out))
            v = list(out)
            st = lane.stats()
            d = lambda a_, b_: int(v[b_] - v[a_])  # noqa: E731
            print(f"step {k} {name}: V {st.num_visible} R {st.num_rendered} sort {dbg.sort_state(lane.geom)}")
            print(f"   prepare (1024 threads): counts+sums {d(0, 1)} | scan {d(1, 22)} | table + blind decision {d(22, 23)} | "
                  f"ranges + sample search {d(23, 24)} | sample keys {d(24, 2)} | validate / new splitters {d(2, 3)} | total {d(0, 3)}")
            # (slots 10-13 -- the slowest compaction workgroup, kept by an atomicMax -- start from whatever the allocator
            #  left in a fresh state buffer: a "maximum" beyond a second of cycles was never written by a workgroup)
            slow = (f"slowest wg {v[10]} clk (wg {v[11]}, {v[12]} blocks, {v[13]} records)"
                    if 0 < v[10] < 2_000_000_000 and v[11] < 65536 else "slowest wg: not recorded")
            print(f"   compact wg64: plan {d(8, 9)} | offsets {d(9, 4)} | walk {d(4, 5)} | classify {d(5, 6)} | table {d(6, 7)} | total {d(8, 7)}; "
                  f"{slow}")
            if 0 < v[48] < 2_000_000_000 and v[52] > 0:
                print(f"   buckets: slowest wg {v[48]} clk ({v[49]} records, bucket {v[50]}), mean {v[51] / v[52]:.0f} clk over {v[52]} workgroups; longest tie fix-up {v[53]} clk (bucket of {v[54]})")
            print(f"   partition wg64: setup {int(v[17] - v[16])} move {int(v[18] - v[17])} | buckets wg100: hdr {int(v[33] - v[32])} "
                  f"sort {int(v[35] - v[33])} emit {int(v[36] - v[35])} (quantiles {int(v[37] - v[35])}, gather + first scan {int(v[38] - v[37])}) "
                  f"bits {v[40]} n {v[41]}")
