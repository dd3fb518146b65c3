#!/usr/bin/env python
"""RCCL under gsworld_amd.distributed on a ONE-GPU box: a process group of one rank over the "nccl" backend (RCCL on ROCm), and
the frame gather of BASELINE.json configs[3] forced to issue its collective anyway (FrameGather(force_collective=True)) --
communicator set-up bound to the device, the collective on the side stream, its ordering against frames rendered under hipGraph
replay on other streams, double-buffered slots, HIP-event timing.  What a one-rank group cannot show: xGMI traffic, a straggling
root, eight communicators.  Prints one JSON line.  usage: rccl_world1.py [batches] [num_gaussians]"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsworld_amd import distributed as gd, scenes  # noqa: E402
from gsworld_amd.renderer import MultiCameraRenderer  # noqa: E402


def main():
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 400_000
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {"tool": "rccl_world1", **gd.rccl_info(), "num_gaussians": n}
    W, H, B, G = 640, 480, 4, 2
    raw = scenes.tabletop_scene("xarm6_align", n=n, seed=1)
    cam = scenes.sensor_camera("xarm6_align", W, H).to(dev)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    bg = torch.zeros(3, device=dev)
    kw = dict(shs=shs, scales=sc, rotations=rot, bg=bg)
    for collective in ("gather", "all_gather"):
        fg = gd.FrameGather(H, W, batch=B * G, device=dev, world=1, buffers=2, collective=collective, timing=True,
                            force_collective=True)
        streams = [torch.cuda.Stream(dev) for _ in range(G)]
        mcs = [MultiCameraRenderer(B, dev, batched=True, forward_only=True, want_radii=False) for _ in range(G)]
        # one reference frame (every slot renders the same view: any slot that differs was read too early or overwritten)
        ref = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
        mcs[0].render([cam] * B, means, op, rgb8_out=[ref] + [torch.empty_like(ref) for _ in range(B - 1)], **kw)
        mcs[0].ensure_valid(lambda: mcs[0].render([cam] * B, means, op, rgb8_out=[ref] + [torch.empty_like(ref) for _ in range(B - 1)], **kw))
        torch.cuda.synchronize()
        # a hipGraph per (stream, half): the step renders its B slots
        graphs = {}
        for g in range(G):
            for half in range(2):
                outs = [fg.slot(half * fg.batch + g * B + j) for j in range(B)]
                with torch.cuda.stream(streams[g]):
                    mcs[g].render([cam] * B, means, op, rgb8_out=outs, **kw)
                    mcs[g].ensure_valid(lambda g=g, outs=outs: mcs[g].render([cam] * B, means, op, rgb8_out=outs, **kw))
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=streams[g]):
                    mcs[g].render([cam] * B, means, op, rgb8_out=outs, **kw)
                graphs[(g, half)] = gr
        torch.cuda.synchronize()
        wrong = 0
        t0 = time.perf_counter()
        cur = torch.cuda.current_stream(dev)
        for b in range(batches):
            half = b % 2
            for g in range(G):
                i0 = b * fg.batch + g * B
                fg.wait_reusable(i0, streams[g])  # (the collective that last read these slots)
                with torch.cuda.stream(streams[g]):
                    graphs[(g, half)].replay()
            for g in range(G):
                cur.wait_stream(streams[g])
            for i in range(b * fg.batch, (b + 1) * fg.batch):
                fg.step_done(i)
            got = fg.wait_gathered()
            wrong += sum(0 if torch.equal(got[j], ref) else 1 for j in range(got.shape[0]))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[collective] = {"batches": batches, "frames_per_batch": fg.batch, "frames_gathered": batches * fg.batch,
                           "frames_that_differ_from_the_reference_frame": wrong, "gather_ms": fg.gather_time_ms(),
                           "frames_per_s_with_a_check_per_batch": batches * fg.batch / dt,
                           "gathered_shape": list(fg.gathered.shape)}
    dist.destroy_process_group()
    out["ok"] = all(out[c]["frames_that_differ_from_the_reference_frame"] == 0 for c in ("gather", "all_gather"))
    print(json.dumps(out))
    return 0 if out["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
