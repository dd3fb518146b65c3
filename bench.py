#!/usr/bin/env python
"""bench.py -- rendered frames/s of the forward 3DGS rasterizer at BASELINE.json's headline configuration.

One "step" = one pass of the hot path over one frame: project -> bin -> depth/tile sort -> SH colour -> composite,
plus GSWorld's uint8 frame conversion, on a synthetic xarm6_align-like scene (1,468,850 Gaussians, 640x480,
sensor camera `right_cam`; BASELINE.json configs[1]).  Inputs are resident in HBM before the timed
region.  N > 1: one independent scene per GPU (configs[3]); finished uint8 frames are gathered with RCCL.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); 6290 GB/s measured copy


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--num-gaussians", type=int, default=None, help="override N (default: 1,468,850)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--gather-every", type=int, default=16, help="frames per RCCL gather batch (N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=5)
    ap.add_argument("--breakdown", action="store_true", help="print a per-stage event timing table to stderr")
    ap.add_argument("--no-graph", action="store_true", help="do not capture the frame in a hipGraph")
    ap.add_argument("--render-bpc", type=int, default=0, help="persistent compositing workgroups per CU (0 = library default)")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="independent frames in flight, each on its own HIP stream with its own renderer state "
                         "(GSWorld renders 2 cameras per step; 1 = strictly one frame at a time)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    from gsworld_amd import scenes
    from gsworld_amd._lib import GsrProfile, PROFILE_STAGES, check, lib
    from gsworld_amd.renderer import FrameRenderer

    from gsworld_amd import distributed as gd

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (torch.cuda.is_available() is False)")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("GSWORLD_DIST_BACKEND", "") == "gloo":
        local_rank = 0  # test hook: all ranks on the one GPU of a test box (see gsworld_amd/distributed.py)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rank, world, _ = gd.init_from_env(dev)  # RCCL ("nccl" backend on ROCm) when WORLD_SIZE > 1

    # ---- scene: one per rank (weak scaling over independent scenes) ---------------------------------------------
    name, seed = gd.scene_for_rank(rank, scenes.SCENE_NAMES)
    n = args.num_gaussians or scenes.XARM6_ALIGN_NUM_GAUSSIANS
    raw = scenes.tabletop_scene(name, n=n, seed=seed)
    cam_cpu = scenes.sensor_camera(name, args.width, args.height)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    cam = cam_cpu.to(dev)
    bg = torch.zeros(3, device=dev)  # gs_world_wrapper.py:234-235
    W, H = args.width, args.height

    S = max(1, args.in_flight)
    # compositing workgroups per CU: the library default (6: every tile quadrant resident at once) is the optimum
    # both for one frame and for 3 frames in flight (tools/sweep_bench.sh); the flag is for sweeps
    bpc = args.render_bpc
    if bpc:
        from gsworld_amd import debug as dbg

        dbg.set_render_variant(4, bpc)
    K_g = max(1, args.gather_every)
    K_g = (K_g + S - 1) // S * S  # a frame slot belongs to exactly one lane: lane = slot % S
    fg = gd.FrameGather(H, W, batch=K_g, device=dev, world=world, buffers=2 if world > 1 else 1)
    n_slots = fg.num_slots
    rs_ = [FrameRenderer(dev) for _ in range(S)]
    lanes = [torch.cuda.Stream(dev) for _ in range(S)] if S > 1 else [torch.cuda.current_stream(dev)]
    r = rs_[0]

    def frame(slot, lane=None):
        rr = rs_[slot % S if lane is None else lane]
        # the uint8 HWC frame GSWorld consumes is written by the compositor itself (GsrOutputs.out_rgb8)
        rr.render(cam, means, op, shs=shs, scales=sc, rotations=rot, bg=bg, rgb8_out=fg.frames[slot])

    # exact-mode frame sizes the binning capacity from the real R; then check the no-sync path is valid
    for l in range(S):
        for _ in range(2):
            frame(l, l)
            rs_[l].ensure_valid(lambda l=l: frame(l, l))
    torch.cuda.synchronize()

    # ---- hipGraph capture of one frame per slot (launch-bound inner loop) ----------------------------------------
    graph = None
    if not args.no_graph:
        try:
            graphs = []
            for slot in range(n_slots):
                st = lanes[slot % S] if S > 1 else torch.cuda.Stream(dev)
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    frame(slot)  # warm the capture stream
                torch.cuda.current_stream().wait_stream(st)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    frame(slot)
                graphs.append(g)
            graph = graphs
        except Exception as ex:  # noqa: BLE001
            print(f"[bench] hipGraph capture failed ({type(ex).__name__}: {ex}); running eager", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

    def step(i):
        slot = i % n_slots
        if S > 1:
            lane = lanes[slot % S]
            if world > 1 and (i % K_g) < S:
                fg.wait_reusable(i, lane)  # first frame of a batch on this lane: its half was gathered a batch ago
            with torch.cuda.stream(lane):
                if graph is not None:
                    graph[slot].replay()
                else:
                    frame(slot)
            if world > 1 and (i % K_g) == K_g - 1:
                cur = torch.cuda.current_stream()
                for st in lanes:
                    cur.wait_stream(st)  # every lane has written its slots of this batch
                fg.step_done(i)          # RCCL all_gather on a side stream; the lanes go on with the other half
        else:
            if world > 1 and (i % K_g) == 0:
                fg.wait_reusable(i)
            if graph is not None:
                graph[slot].replay()
            else:
                frame(slot)
            fg.step_done(i)  # RCCL all_gather of the batch of finished frames on a side stream (N > 1)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()

    # ---- timed region: exactly K steps; HIP events around the compositing kernel on the launch stream --------
    # (events are recorded inside libgsr_hip.so on the stream the kernels run on; not available under graph replay)
    profile_mode = 1 if graph is None else 0
    check(lib().gsr_profile_enable(profile_mode))
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    prof = GsrProfile()
    check(lib().gsr_profile_collect(prof))
    check(lib().gsr_profile_enable(0))

    # per-kernel time of the dominant kernel: eager re-run of K frames with events (same stream, same inputs)
    check(lib().gsr_profile_enable(2))
    for i in range(min(args.steps, 200)):
        frame(i % n_slots)
    torch.cuda.synchronize()
    prof2 = GsrProfile()
    check(lib().gsr_profile_collect(prof2))
    check(lib().gsr_profile_enable(0))
    stage_ms = [prof2.stage_ms[k] / max(prof2.frames, 1) for k in range(len(PROFILE_STAGES))]
    if prof.frames > 0:
        stage_ms[-1] = prof.stage_ms[len(PROFILE_STAGES) - 1] / prof.frames  # measured inside the timed region

    # per-frame latency distribution (hipEvent per frame, median and p95), strictly one frame at a time
    n_lat = min(args.steps, 200)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_lat)]
    for i, (e0, e1) in enumerate(ev):
        e0.record()
        if graph is not None and S == 1:
            graph[i % n_slots].replay()
        else:
            frame(i % n_slots)
        e1.record()
    torch.cuda.synchronize()
    lat = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    frame_ms_p50, frame_ms_p95 = lat[len(lat) // 2], lat[min(len(lat) - 1, int(0.95 * len(lat)))]

    stats = r.ensure_valid(lambda: frame(0))
    for l in range(1, S):
        if rs_[l].stats().overflow:
            raise SystemExit("binning capacity overflowed on a lane during the timed region: result invalid")
    if stats.overflow:
        raise SystemExit("binning capacity overflowed during the timed region: result invalid")

    t_max = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    elapsed = float(t_max.item())
    fps = world * args.steps / elapsed

    if rank == 0:
        b_alg = stats.algorithmic_bytes(W, H)
        render_ms = stage_ms[-1]
        render_bytes = 40 * stats.num_rendered + 16 * W * H  # 40 B per composited instance + outputs
        ach = render_bytes / (render_ms * 1e-3) / 1e9 if render_ms > 0 else 0.0
        traffic = valu_frac = None
        pmc = os.path.join(ROOT, "profiles", "pmc_render.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                traffic = rec.get("hbm_bytes_per_launch")
                # what actually bounds the compositor: VALU issue.  SQ_INSTS_VALU wave-instructions (committed PMC
                # run of the same frame) x 4 cycles on 256 CUs x 4 SIMDs at 2.4 GHz, over the kernel time measured now
                insts = rec.get("counters", {}).get("SQ_INSTS_VALU")
                if insts and render_ms > 0:
                    valu_frac = insts * 4.0 / (256 * 4 * 2.4e9 * render_ms * 1e-3)
            except Exception:  # noqa: BLE001
                traffic = valu_frac = None
        out = {
            "metric": "rendered frames/sec @640x480, 1.5M Gaussians",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{name}-like synthetic scene, {n} Gaussians, {W}x{H} right_cam, forward-only "
                            "(BASELINE.json configs[1]; one scene per GPU for N>1 = configs[3])",
                "num_gaussians": n, "num_visible": stats.num_visible, "num_rendered": stats.num_rendered,
                "sh_degree": 3, "launch": "hipGraph replay" if graph is not None else "eager",
                "frames_in_flight": S,
                "frame_gather": f"RCCL all_gather of uint8 frames every {K_g} frames" if world > 1 else "none",
            },
            "roofline": {
                "bound": "hbm", "kernel": "render_stream_kernel", "achieved": ach, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": render_bytes, "kernel_ms": render_ms,
                "valu_issue_frac": valu_frac,
                "note": "HIP events around the kernel on its launch stream, one frame in flight (events cannot be "
                        "recorded inside a replayed hipGraph).  The compositor is VALU/exp-bound, not HBM-bound: "
                        "saturated pixels stop reading their tile list early, so real traffic is far below the "
                        "algorithmic 40 B x num_rendered (DESIGN.md); whole-frame figures below",
            },
            "frame_roofline": {
                "algorithmic_bytes_per_frame": b_alg, "achieved_GBs": b_alg * fps / world / 1e9,
                "frac_of_8TBs": b_alg * fps / world / 1e9 / HBM_PEAK_GBS,
                "frac_of_6.3TBs": b_alg * fps / world / 1e9 / 6290.0,
                "stage_ms": dict(zip(PROFILE_STAGES, stage_ms)),
                "single_frame_latency_ms": sum(stage_ms),
                "frame_ms_p50": frame_ms_p50, "frame_ms_p95": frame_ms_p95,  # one frame at a time, HIP events
            },
        }
        if args.breakdown:
            print("[bench] stage ms:", dict(zip(PROFILE_STAGES, stage_ms)), file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(raw, cam_cpu, args.cpu_frames)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(raw, cam, frames):
    """The oracle (CPU restatement of the reference algorithm, OpenMP) on the SAME frame, host cores of this box.
    Bounded sample: `frames` whole frames (about 2-3 s each on 8 cores)."""
    import numpy as np

    from oracle import gs_oracle as go

    means, shs, op, sc, rot = (t.numpy() for t in raw.activated())
    st = go.Settings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy)
    cores = os.cpu_count() or 1
    go.set_threads(cores)
    args = (st, np.zeros(3, np.float32), means, shs, None, op.reshape(-1), sc, rot, None,
            cam.world_view_transform.numpy().reshape(-1), cam.full_proj_transform.numpy().reshape(-1),
            cam.camera_center.numpy())
    go.forward(*args)  # warm-up
    ts = []
    for _ in range(frames):
        t0 = time.perf_counter()
        go.forward(*args)
        ts.append(time.perf_counter() - t0)
    med = sorted(ts)[len(ts) // 2]
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": 1.0 / med, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"median of {frames} whole frames of the same scene/camera after 1 warm-up; "
                      f"oracle/gs_oracle.c with OpenMP (preprocess, render) + single-thread radix sort; CPU: {model}"}


if __name__ == "__main__":
    main()
