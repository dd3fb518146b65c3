#!/usr/bin/env python
"""Frames per launch (gsr_forward_batch) against frames per stream, on the headline frame and the dense view.

    python tools/ab_batch.py [--view sensor|dense] [--steps 400] [--configs streams4,batch2,batch4,batch8,2x4,2x8]

One JSON line per configuration: frames/s over `--steps` frames after a warm-up, hipGraph replay.
  streamsS : S lanes, one frame per lane and stream (round 4's headline arrangement)
  batchB   : B frames per gsr_forward_batch call, ONE stream
  GxB      : G such batches alternating on G streams (a batch's sort chain under another's compositor)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--view", default="sensor")
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--configs", default="streams1,streams4,batch1,batch2,batch4,batch8,2x4,2x8,3x4")
    ap.add_argument("--eager", action="store_true", help="no graph capture (for rocprofv3 kernel names per launch)")
    args = ap.parse_args()
    import torch

    from gsworld_amd import scenes
    from gsworld_amd.layout import SceneLayout
    from gsworld_amd.renderer import FrameRenderer, MultiCameraRenderer

    dev = torch.device("cuda", 0)
    name = "xarm6_align"
    raw = scenes.tabletop_scene(name, n=args.n or scenes.XARM6_ALIGN_NUM_GAUSSIANS, seed=1)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    L = SceneLayout.build(means, sc, rot, shs=shs, opacities=op)
    a = L.arrays
    W, H = 640, 480
    cam = (scenes.sensor_camera(name, W, H) if args.view == "sensor" else scenes.dense_view_camera(name, W, H)).to(dev)
    bg = torch.zeros(3, device=dev)
    kw = dict(shs=a["shs"], scales=a["scales"], rotations=a["rotations"], bg=bg, layout=L.layout)

    def graphed(fn, stream):
        stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(stream):
            fn()
        torch.cuda.synchronize()
        if args.eager:
            class E:
                def replay(self_):
                    fn()
            return E()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            fn()
        return g

    def timeit(enqueue, frames_per_call, n_calls, warm):
        for k in range(warm):
            enqueue(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n_calls):
            enqueue(warm + k)
        torch.cuda.synchronize()
        return n_calls * frames_per_call / (time.perf_counter() - t0)

    ref = None
    for cfg in args.configs.split(","):
        if cfg.startswith("streams"):
            S = int(cfg[7:])
            streams = [torch.cuda.Stream(dev) for _ in range(S)]
            rs = [FrameRenderer(dev, forward_only=True, want_radii=False) for _ in range(S)]
            outs = [torch.zeros((H, W, 3), dtype=torch.uint8, device=dev) for _ in range(S)]
            fns = [(lambda l=l: rs[l].render(cam, a["means3D"], a["opacities"], rgb8_out=outs[l], **kw)) for l in range(S)]
            gs = []
            for l in range(S):
                for _ in range(2):
                    fns[l]()
                    rs[l].ensure_valid(fns[l])
                gs.append(graphed(fns[l], streams[l]))

            def enq(k):
                with torch.cuda.stream(streams[k % S]):
                    gs[k % S].replay()

            fps = timeit(enq, 1, args.steps, 8 * S)
            ovf = any(r.stats().overflow for r in rs)
            rec = {"config": cfg, "frames_per_s": fps, "overflow": ovf}
            if ref is None:
                ref = outs[0].clone()
            rec["same_frame"] = bool(torch.equal(outs[0], ref))
        else:
            G, B = (int(x) for x in cfg.split("x")) if "x" in cfg else (1, int(cfg[5:]))
            streams = [torch.cuda.Stream(dev) for _ in range(G)]
            mcs = [MultiCameraRenderer(B, dev, batched=True, forward_only=True, want_radii=False) for _ in range(G)]
            outs = [[torch.zeros((H, W, 3), dtype=torch.uint8, device=dev) for _ in range(B)] for _ in range(G)]
            fns = [(lambda g=g: mcs[g].render([cam] * B, a["means3D"], a["opacities"], rgb8_out=outs[g], **kw))
                   for g in range(G)]
            gs = []
            for g in range(G):
                for _ in range(2):
                    fns[g]()
                    mcs[g].ensure_valid(fns[g])
                gs.append(graphed(fns[g], streams[g]))

            def enq(k):
                with torch.cuda.stream(streams[k % G]):
                    gs[k % G].replay()

            fps = timeit(enq, B, max(args.steps // B, 8), 8 * G)
            ovf = any(l.stats().overflow for m in mcs for l in m.lanes)
            rec = {"config": cfg, "frames_per_s": fps, "overflow": ovf}
            if ref is None:
                ref = outs[0][0].clone()
            rec["same_frame"] = bool(all(torch.equal(o, ref) for og in outs for o in og))
        rec["view"] = args.view
        print(json.dumps(rec), flush=True)
        del gs


if __name__ == "__main__":
    main()
