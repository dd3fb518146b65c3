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
    ap.add_argument("--collective", choices=["gather", "all_gather"], default="gather",
                    help="N > 1: gather-to-rank-0 (default, what north_star asks for) or all_gather of the uint8 frames")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary measurements (dense view, upstream packing, closed loop, torch-CPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=5)
    ap.add_argument("--breakdown", action="store_true", help="print a per-stage event timing table to stderr")
    ap.add_argument("--no-graph", action="store_true", help="do not capture the step in a hipGraph")
    ap.add_argument("--render-bpc", type=int, default=0, help="persistent compositing workgroups per CU (0 = library default)")
    ap.add_argument("--no-layout", action="store_true",
                    help="render the model in the order it was given, without block culling (gsworld_amd/layout.py)")
    ap.add_argument("--blocks", type=int, default=25,
                    help="timed blocks of --steps steps each (at least this many, and until --min-seconds are timed); "
                         "value = their median")
    ap.add_argument("--min-seconds", type=float, default=0.3)
    ap.add_argument("--batch", type=int, default=8,
                    help="frames per step: ONE gsr_forward_batch call whose launches span them (include/gsr.h; 1..8)")
    ap.add_argument("--streams", type=int, default=3,
                    help="steps in flight: consecutive steps alternate over this many HIP streams, each with its own renderer "
                         "states (1 = every launch on one stream)")
    ap.add_argument("--only-steps", action="store_true",
                    help="profiling aid: nothing but the step's own launches after the set-up (no one-frame-per-launch stage "
                         "table, latency loop or one-frame figure), so that a rocprofv3 --stats average of this command is "
                         "the average of the step's launches")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="A/B of rounds 2-4: N frames in flight, ONE frame per step and stream (= --batch 1 --streams N)")
    a = ap.parse_args()
    if a.in_flight > 0:
        a.batch, a.streams = 1, a.in_flight
    if not 1 <= a.batch <= 8 or a.streams < 1:
        ap.error("--batch must be 1..8, --streams >= 1")
    return a


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    from gsworld_amd import scenes
    from gsworld_amd._lib import GsrProfile, PROFILE_STAGES, check, lib
    from gsworld_amd.renderer import FrameRenderer, MultiCameraRenderer

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
    # Load-time layout (gsworld_amd/layout.py): a copy of the model in Morton order per size class + bounds of every block
    # of 256 Gaussians, built ONCE per scene -- a GSWorld scene is loaded once and rendered for whole episodes.  Frames
    # then skip the blocks no tile can see (71 % of them from right_cam); same bits out (checked below against a frame
    # of the model as given).  Not part of the timed region, like the scene load itself; its cost is in the line.
    lay, layout_s = None, None
    m_means, m_shs, m_op, m_sc, m_rot = means, shs, op, sc, rot
    if not args.no_layout:
        from gsworld_amd.layout import SceneLayout

        torch.cuda.synchronize()
        t_l = time.perf_counter()
        L = SceneLayout.build(means, sc, rot, shs=shs, opacities=op)
        torch.cuda.synchronize()
        layout_s = time.perf_counter() - t_l
        a = L.arrays
        m_means, m_shs, m_op, m_sc, m_rot, lay = a["means3D"], a["shs"], a["opacities"], a["scales"], a["rotations"], L.layout
    cam = cam_cpu.to(dev)
    bg = torch.zeros(3, device=dev)  # gs_world_wrapper.py:234-235
    W, H = args.width, args.height

    # ---- the step: B frames through ONE gsr_forward_batch call (one set of launches whose grids span the frames),
    # consecutive steps on G streams in turn.  Fixed by flags -- nothing is chosen by a trial inside the run.
    B, G = args.batch, args.streams
    group_streams = [torch.cuda.Stream(dev) for _ in range(G)] if G > 1 else [torch.cuda.current_stream(dev)]
    # compositing workgroups per CU: the library default (6: every tile quadrant resident at once); the flag is for sweeps
    if args.render_bpc:
        from gsworld_amd import debug as dbg

        dbg.set_render_variant(4, args.render_bpc)
    K_g = max(1, args.gather_every)
    K_g = (K_g + B * G - 1) // (B * G) * (B * G)  # a gather batch is a whole number of steps of every stream
    fg = gd.FrameGather(H, W, batch=K_g, device=dev, world=world, buffers=2 if world > 1 else 1,
                        collective=args.collective, timing=world > 1)
    n_slots = fg.num_slots
    n_groups = n_slots // B  # distinct steps (output slots) before the frame buffers are reused
    # inference frames (GsrSettings.forward_only): GSWorld's loop keeps ["render"] only (gs_world_wrapper.py:266-270) --
    # nothing a backward would read is written, instances are binned per 2 x 1 super-tile; the image is bit-identical
    # (tests/test_renderer_gpu.py, and checked against a default frame right below)
    mcs = [MultiCameraRenderer(B, dev, batched=True, forward_only=True, want_radii=False) for _ in range(G)]
    r1 = FrameRenderer(dev, forward_only=True, want_radii=False)   # one frame, one gsr_forward call (latency, stage table)
    one_rgb8 = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
    # N, V, R of SURVEY.md 8d's byte model are those of the reference's own per-tile pipeline: one default frame gives them
    ref_r = FrameRenderer(dev)
    ref_rgb8 = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
    ref_r.render(cam, means, op, shs=shs, scales=sc, rotations=rot, bg=bg, rgb8_out=ref_rgb8, exact=True)
    true_stats = ref_r.stats()
    del ref_r
    kw = dict(shs=m_shs, scales=m_sc, rotations=m_rot, bg=bg, layout=lay)

    def group(j):
        """Step j: the B frames of output slots j B .. j B + B - 1 on renderer set j mod G."""
        base = (j % n_groups) * B
        mcs[j % G].render([cam] * B, m_means, m_op, rgb8_out=[fg.frames[base + b] for b in range(B)], **kw)

    def frame1():
        r1.render(cam, m_means, m_op, rgb8_out=one_rgb8, **kw)

    # exact-mode frames size the binning capacities from the real R; then check the no-sync path is valid
    for g in range(G):
        for _ in range(2):
            group(g)
            mcs[g].ensure_valid(lambda g=g: group(g))
    for _ in range(2):
        frame1()
        r1.ensure_valid(frame1)
    torch.cuda.synchronize()
    for b in range(B):
        if not torch.equal(fg.frames[b], ref_rgb8):
            raise SystemExit("forward_only frame differs from the default frame: result invalid")
    if not torch.equal(one_rgb8, ref_rgb8):
        raise SystemExit("forward_only frame differs from the default frame: result invalid")

    # ---- hipGraph capture of every distinct step (launch-bound inner loop) ---------------------------------------
    graph = None
    if not args.no_graph:
        try:
            graphs = []
            for j in range(n_groups):
                st = group_streams[j % G] if G > 1 else torch.cuda.Stream(dev)
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    group(j)  # warm the capture stream
                torch.cuda.current_stream().wait_stream(st)
                g_ = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_, stream=st):
                    group(j)
                graphs.append(g_)
            graph = graphs
        except Exception as ex:  # noqa: BLE001
            print(f"[bench] hipGraph capture failed ({type(ex).__name__}: {ex}); running eager", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

    steps_per_gather = K_g // B

    def step(i):
        j = i % n_groups
        st = group_streams[i % G]
        if world > 1 and (i % steps_per_gather) < G:
            fg.wait_reusable(i * B, st if G > 1 else None)  # first step of a gather batch on this stream
        if G > 1:
            with torch.cuda.stream(st):
                graph[j].replay() if graph is not None else group(j)
        else:
            graph[j].replay() if graph is not None else group(j)
        if world > 1 and (i % steps_per_gather) == steps_per_gather - 1:
            if G > 1:
                cur = torch.cuda.current_stream()
                for s_ in group_streams:
                    cur.wait_stream(s_)  # every stream has written its slots of this gather batch
            fg.step_done(i * B + B - 1)  # RCCL gather on a side stream; the renderers go on with the other half
        elif world == 1:
            pass

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()

    # ---- timed region: exactly K steps between barriers, block after block: at ~0.3 ms a step one block of the driver's
    # K is a few milliseconds, so at least --blocks of them and --min-seconds are timed; the line carries the median
    # block, the spread and the total-time figure
    block_s, done = [], 0
    while len(block_s) < max(1, args.blocks) or (sum(block_s) < args.min_seconds and len(block_s) < 400):
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(done + i)
        barrier()
        block_s.append(time.perf_counter() - t0)
        done += args.steps
    elapsed = sorted(block_s)[len(block_s) // 2]
    # Every frame slot is checked AGAIN behind the timed region, byte for byte against the default exact-mode frame: the
    # routes a static camera earns only after several balanced frames (kept splitters taken blind, buckets of 1024 records,
    # the previous frame's placement cuts, cooperative quadrants dealt by last frame's costs) are engaged in the timed
    # steps, not in the frames checked before them (VERDICT round 5, weak #1d).  A mismatch fails the run.
    verified_after = 0
    for b in range(n_slots):
        if not torch.equal(fg.frames[b], ref_rgb8):
            raise SystemExit(f"frame slot {b} differs from the default frame AFTER the timed region: result invalid")
        verified_after += 1

    # ---- the dominant kernel's launch duration, HIP events on its launch stream: (a) the step's own launch (B frames
    # per launch), eager on one stream; (b) one frame per launch.  Every stage of both (events cannot be recorded inside
    # a replayed hipGraph)
    def staged(fn, reps):
        check(lib().gsr_profile_enable(2))
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        pr = GsrProfile()
        check(lib().gsr_profile_collect(pr))
        check(lib().gsr_profile_enable(0))
        return [pr.stage_ms[k] / max(pr.frames, 1) for k in range(len(PROFILE_STAGES))]

    reps = min(args.steps, 200)
    for _ in range(5):
        group(0)
    torch.cuda.synchronize()
    stage_ms_step = staged(lambda: group(0), max(reps // B, 20))
    stage_ms = staged(frame1, reps) if not args.only_steps else [0.0] * len(PROFILE_STAGES)

    # per-frame latency distribution (hipEvent per frame, median and p95), strictly one frame at a time
    g1 = None
    if not args.no_graph and not args.only_steps:
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            frame1()
        torch.cuda.current_stream().wait_stream(side)
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, stream=side):
            frame1()
    n_lat = min(max(args.steps, 100), 200) if not args.only_steps else 0
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_lat)]
    for e0, e1 in ev:
        e0.record()
        g1.replay() if g1 is not None else frame1()
        e1.record()
    torch.cuda.synchronize()
    lat = sorted(e0.elapsed_time(e1) for e0, e1 in ev) or [0.0]
    frame_ms_p50, frame_ms_p95 = lat[len(lat) // 2], lat[min(len(lat) - 1, int(0.95 * len(lat)))]

    for g in range(G):
        if any(s_.overflow for s_ in mcs[g].ensure_valid(lambda: None)):
            raise SystemExit("binning capacity overflowed during the timed region: result invalid")
    stats = r1.ensure_valid(frame1) if not args.only_steps else r1.stats()
    if stats.overflow:
        raise SystemExit("binning capacity overflowed during the timed region: result invalid")

    own_elapsed = elapsed
    t_max = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    per_rank = [own_elapsed]
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        every = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(every, torch.tensor([own_elapsed], dtype=torch.float64, device=dev))
        per_rank = [float(t.item()) for t in every]
    elapsed = float(t_max.item())
    fps = world * B * args.steps / elapsed

    # the same frame in other arrangements, for the line's config: strictly one frame at a time (the figure a caller
    # sees who needs frame k before it can ask for frame k + 1), and the step's B frames per launch on ONE stream
    one_fps = one_stream_fps = None
    if rank == 0 and world == 1 and graph is not None:
        one_fps = _time_frames(torch, g1.replay, min(max(args.steps, 100), 300))
        if G > 1:
            k_ = [0]

            def enq():
                graph[(k_[0] * G) % n_groups].replay()  # (the steps captured on stream 0)
                k_[0] += 1

            with torch.cuda.stream(group_streams[0]):
                one_stream_fps = B * _time_frames(torch, enq, min(max(args.steps, 50), 100))
        else:
            one_stream_fps = fps
    extras = {}
    if rank == 0 and world == 1 and not args.no_extras:
        extras = secondary_measurements(args, dev, raw, name, (means, shs, op, sc, rot),
                                        (m_means, m_shs, m_op, m_sc, m_rot, lay), bg, group_streams if G > 1 else None)

    if rank == 0:
        binned = stats.num_rendered  # instances actually placed (super-tile lists)
        stats = true_stats           # the byte model counts the reference's per-tile instances
        b_alg = stats.algorithmic_bytes(W, H)
        render_ms_step = stage_ms_step[-1]          # one launch of the step: B frames
        render_ms = stage_ms[-1]                    # one launch of one frame
        frame_render_bytes = 40 * stats.num_rendered + 16 * W * H  # 40 B per composited instance + outputs
        render_bytes = B * frame_render_bytes
        ach = render_bytes / (render_ms_step * 1e-3) / 1e9 if render_ms_step > 0 else 0.0
        ach1 = frame_render_bytes / (render_ms * 1e-3) / 1e9 if render_ms > 0 else 0.0
        traffic = valu_frac = None
        traffic_source = "none"
        pmc = os.path.join(ROOT, "profiles", "pmc_render.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                traffic = rec.get("hbm_bytes_per_launch")
                # NOT measured in this run (PMC counters need a rocprofv3 session): a committed counter run of the same
                # launch.  The record carries the hash of render.hip it was taken with and the frames per launch; a
                # kernel edited since then is flagged instead of silently quoted.
                import hashlib
                sha = hashlib.sha256(open(os.path.join(ROOT, "gsworld_amd", "csrc", "render.hip"), "rb").read()).hexdigest()[:16]
                stale = rec.get("render_hip_sha16") != sha
                per_launch = int(rec.get("frames_per_launch", 1))
                if traffic is not None and per_launch != B:
                    traffic = traffic * B / per_launch
                traffic_source = (f"committed rocprofv3 --pmc run {os.path.relpath(pmc, ROOT)} "
                                  f"({rec.get('collected', 'undated')}, {per_launch} frame(s) per launch"
                                  f"{'' if per_launch == B else f', scaled to {B}'}; render.hip "
                                  f"{'CHANGED since then: stale' if stale else 'unchanged since then'})")
                # what actually bounds the compositor: VALU issue.  SQ_INSTS_VALU wave-instructions per frame (committed
                # PMC run) x 4 cycles on 256 CUs x 4 SIMDs at the 2.13 GHz the stamps measured under this kernel
                insts = rec.get("counters", {}).get("SQ_INSTS_VALU")
                if insts and render_ms_step > 0:
                    valu_frac = (insts / per_launch) * B * 4.0 / (256 * 4 * 2.13e9 * render_ms_step * 1e-3)
            except Exception:  # noqa: BLE001
                traffic = valu_frac = None
        # the same kernel's average in the committed rocprofv3 --kernel-trace --stats run of this command (profiles/)
        rocprof_ms = None
        try:
            import csv
            import glob

            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*", "kernel_stats_bench_step_launches.csv"))):
                for row in csv.DictReader(open(path)):
                    if "render_stream_kernel<true>" in row["Name"]:
                        rocprof_ms = {"file": os.path.relpath(path, ROOT), "avg_ms": float(row["AverageNs"]) * 1e-6}
        except Exception:  # noqa: BLE001
            rocprof_ms = None
        total_fps = world * B * args.steps * len(block_s) / sum(block_s)
        out = {
            "metric": "rendered frames/sec @640x480, 1.5M Gaussians",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{name}-like synthetic scene, {n} Gaussians, {W}x{H} right_cam, forward-only "
                            "(BASELINE.json configs[1]; one scene per GPU for N>1 = configs[3])",
                "num_gaussians": n, "num_visible": stats.num_visible, "num_rendered": stats.num_rendered,
                "binned_instances": binned,
                "step": f"one pass of the hot path over a batch of {B} frames: ONE gsr_forward_batch call = 10 kernel launches "
                        f"whose grids span the {B} frames (include/gsr.h); value = {B} x steps / time",
                "frames_per_step": B, "steps_in_flight": G,
                "frames_in_flight": B * G,
                "frame_mode": "forward_only (inference: super-tile binning, no backward-only writes; image bit-identical "
                              "to the default frame, checked in this run before AND after the timed region)",
                "frame_slots_verified_after_timed_region": verified_after,
                "sh_degree": 3, "launch": "hipGraph replay" if graph is not None else "eager",
                "comparability": "rounds 1-3: value = frames / total time, one frame per step, 1-3 frames in flight on "
                                 "streams; round 4: median of 5 blocks, 3 or 4 stream lanes picked by a trial in the run; "
                                 "round 5: fixed arrangement (flags --batch / --streams), median of >= 25 blocks, "
                                 "total-time figure beside it",
                # the same frame: strictly one at a time / the step's launches on one stream / camera turned every frame
                "one_frame_in_flight_frames_per_s": one_fps,
                "one_stream_frames_per_s": one_stream_fps,
                "moving_camera_frames_per_s": extras.get("moving_camera", {}).get("frames_per_s"),
                "model_layout": ("Morton order per size class + block bounds built once per scene "
                                 f"(gsworld_amd/layout.py, {layout_s:.3f} s on this box, outside the timed region): "
                                 "preprocess skips the blocks of 256 Gaussians no tile can see; image, radii and tie "
                                 "order unchanged") if lay is not None else "model as given (--no-layout)",
                "timed_blocks": {"blocks": len(block_s), "steps_each": args.steps, "seconds": sum(block_s),
                                 "frames_per_s_min_median_max": [world * B * args.steps / max(block_s), fps,
                                                                 world * B * args.steps / min(block_s)],
                                 "frames_per_s_total_time": total_fps, "value_is": "median block"},
                "frame_gather": (f"RCCL {args.collective} of uint8 frames every {K_g} frames "
                                 f"(backend {dist.get_backend()}, {dist.get_world_size()} ranks)") if world > 1 else "none",
                "world_size": world,
                # N > 1: which collective library ran, how long a gather of one batch took on its side stream (HIP
                # events) and whether any render stream depended on a collective younger than two batches
                "collective_library": gd.rccl_info() if world > 1 else None,
                "gather_ms_per_batch": fg.gather_time_ms() if world > 1 else None,
                "render_waited_on_batches_back": sorted({(i // K_g) - b for i, b, _ in fg.waits}) if world > 1 else None,
                # every rank's own rate over the same K steps: sum ~ value when no rank is a straggler, and rank 0's
                # figure is directly comparable with the N = 1 run
                "per_rank_frames_per_s": [B * args.steps / t for t in per_rank],
            },
            "roofline": {
                # what binds this kernel is VALU issue (valu_issue_frac below); achieved / peak / frac are the HBM figures
                # the contract asks for (algorithmic bytes per launch over the launch's duration, against 8 TB/s)
                "bound": "valu", "kernel": "render_stream_kernel", "achieved": ach, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "frames_per_launch": B, "algorithmic_bytes_per_launch": render_bytes, "kernel_ms": render_ms_step,
                "kernel_ms_rocprof_committed": rocprof_ms,
                "one_frame_per_launch": {"kernel_ms": render_ms, "achieved": ach1, "frac": ach1 / HBM_PEAK_GBS,
                                         "algorithmic_bytes_per_launch": frame_render_bytes},
                "valu_issue_frac_at_measured_2.13GHz": valu_frac,
                "note": "HIP events around the kernel on its launch stream, the step's launch alone on the chip (events "
                        "cannot be recorded inside a replayed hipGraph).  The compositor is VALU/exp-bound, not "
                        "HBM-bound: saturated pixels stop reading their tile list early, so real traffic is far below "
                        "the algorithmic 40 B x num_rendered (DESIGN.md); whole-frame figures below",
            },
            "frame_roofline": {
                "algorithmic_bytes_per_frame": b_alg, "achieved_GBs": b_alg * fps / world / 1e9,
                "frac_of_8TBs": b_alg * fps / world / 1e9 / HBM_PEAK_GBS,
                "frac_of_6.3TBs": b_alg * fps / world / 1e9 / 6290.0,
                "one_stream_frac_of_8TBs": (b_alg * one_stream_fps / 1e9 / HBM_PEAK_GBS) if one_stream_fps else None,
                "stage_ms_one_frame_per_launch": dict(zip(PROFILE_STAGES, stage_ms)),
                "stage_ms_step_launches": dict(zip(PROFILE_STAGES, stage_ms_step)),
                "single_frame_latency_ms": sum(stage_ms),
                "frame_ms_p50": frame_ms_p50, "frame_ms_p95": frame_ms_p95,  # one frame at a time, HIP events
            },
        }
        if args.breakdown:
            print("[bench] stage ms (one frame per launch):", dict(zip(PROFILE_STAGES, stage_ms)), file=sys.stderr)
            print(f"[bench] stage ms ({B} frames per launch):", dict(zip(PROFILE_STAGES, stage_ms_step)), file=sys.stderr)
        out.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(raw, cam_cpu, args.cpu_frames)
            if not args.no_extras:
                out["cpu_baseline_config0"] = cpu_baseline_config0()
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _time_frames(torch, enqueue, steps, warmup=10):
    for _ in range(warmup):
        enqueue()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        enqueue()
    torch.cuda.synchronize()
    return steps / (time.perf_counter() - t0)


def _arrangement_fps(torch, dev, cams, model_kw, means, op, B, streams, steps, H, W, poses=None):
    """frames/s of `cams` (one ViewParams per stream; each rendered B times per step) in the headline's arrangement: B frames
    per gsr_forward_batch call, consecutive steps on `streams` in turn, hipGraph replay.  -> (frames/s, overflow)."""
    from gsworld_amd.renderer import MultiCameraRenderer

    G = len(streams)
    mcs = [MultiCameraRenderer(B, dev, batched=True, forward_only=True, want_radii=False) for _ in range(G)]
    outs = [[torch.zeros((H, W, 3), dtype=torch.uint8, device=dev) for _ in range(B)] for _ in range(G)]
    fns = [(lambda g=g: mcs[g].render([cams[g]] * B, means, op, rgb8_out=outs[g], **model_kw)) for g in range(G)]
    graphs = []
    for g in range(G):
        for _ in range(2):
            fns[g]()
            mcs[g].ensure_valid(fns[g])
        streams[g].wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(streams[g]):
            fns[g]()
        torch.cuda.synchronize(dev)
        gl = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gl, stream=streams[g]):
            fns[g]()
        graphs.append(gl)
    torch.cuda.synchronize(dev)
    k = [0]

    def enqueue():
        g = k[0] % G
        with torch.cuda.stream(streams[g]):
            if poses is not None:  # a camera that moves every step: three small H2D copies in front of the replay
                wvt, full, center = poses[k[0] % len(poses)]
                cams[g].world_view_transform.copy_(wvt, non_blocking=True)
                cams[g].full_proj_transform.copy_(full, non_blocking=True)
                cams[g].camera_center.copy_(center, non_blocking=True)
            graphs[g].replay()
        k[0] += 1

    f = B * _time_frames(torch, enqueue, max(steps // B, 8 * G), warmup=4 * G)
    ovf = any(x.overflow for m in mcs for x in m.ensure_valid(lambda: None))
    return f, ovf


def secondary_measurements(args, dev, raw, name, model, laid, bg, group_streams=None):
    """SURVEY.md 8d's second numbers, N = 1 only, outside the headline's timed region, each a few hundred frames:
    * dense_view / visibility_sweep: the same scene and N from cameras that see V = 0.6 N / 0.3 N of it (8d's worked
      example assumes 0.6; right_cam sees 0.12 N), one frame at a time and in the headline's arrangement;
    * upstream_packing: the rasterizer figure INCLUDING what upstream render() does per frame before it
      (sigmoid / exp / normalize over the model + cat(dc, rest) -> (N,16,3)), and the same frame with those four passes
      fused into preprocess (raw parameters + split SH, GsrInputs.param_space / shs_rest);
    * closed_loop: BASELINE.json configs[2] surrogate -- 1 reset + 200 steps x 2 cameras through
      gsworld_amd.closed_loop (pose upload, fused transform, both frames in one batched call, one hipGraph replay per
      step), steps enqueued ahead and with the policy in the loop."""
    import torch

    from gsworld_amd import closed_loop as cl, scenes
    from gsworld_amd._lib import RAW_OPACITY, RAW_ROTATIONS, RAW_SCALES
    from gsworld_amd.camera import look_at_view
    from gsworld_amd.renderer import FrameRenderer

    out = {}
    means, shs, op, sc, rot = model                        # the model as given (what a drop-in call hands over)
    l_means, l_shs, l_op, l_sc, l_rot, lay = laid          # the headline's layout (== model with --no-layout)
    W, H = args.width, args.height
    B = args.batch
    steps = min(max(args.steps, 100), 200)
    rgb8 = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
    streams = list(group_streams) if group_streams else [torch.cuda.Stream(dev)]
    l_kw = dict(shs=l_shs, scales=l_sc, rotations=l_rot, bg=bg, layout=lay)

    def graphed(fn):
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            fn()
        return g

    # ---- visibility sweep: V / N = 0.12 (right_cam), ~0.3, 0.6 (SURVEY.md 8d's worked example) ------------------------
    sweep = []
    for label, cam_v in (("right_cam (the headline's view)", scenes.sensor_camera(name, W, H)),
                         ("0.95 m above the table centre, looking down", scenes.dense_view_camera(name, W, H, height_m=0.95)),
                         ("1.5 m above the table centre, looking down (dense_view)", scenes.dense_view_camera(name, W, H))):
        cam_v = cam_v.to(dev)
        rd = FrameRenderer(dev, forward_only=True, want_radii=False)
        fr = lambda: rd.render(cam_v, l_means, l_op, rgb8_out=rgb8, **l_kw)  # noqa: E731
        for _ in range(2):
            fr()
            rd.ensure_valid(fr)
        g = graphed(fr)
        f = _time_frames(torch, g.replay, steps)
        if rd.ensure_valid(fr).overflow:
            raise SystemExit("visibility sweep: capacity overflow")
        # the same view the way the headline is measured.  One frame at a time a dense compositor ends with a few quadrants
        # alone on the chip (a wave is a latency chain, profiles/round4/NOTES_compositor_scheduling.md); the workgroups of
        # the other frames of the launch, and of the steps on the other streams, fill that tail
        f_arr, ovf = _arrangement_fps(torch, dev, [cam_v] * len(streams), l_kw, l_means, l_op, B, streams, steps, H, W)
        if ovf:
            raise SystemExit("visibility sweep (arrangement): capacity overflow")
        rt = FrameRenderer(dev)  # N, V, R of the byte model: the reference's per-tile pipeline (one default frame)
        rt.render(cam_v, means, op, shs=shs, scales=sc, rotations=rot, bg=bg, exact=True)
        st = rt.stats()
        del rt, rd, g
        b_alg = st.algorithmic_bytes(W, H)
        sweep.append({"camera": label, "visible_fraction": st.num_visible / st.num_gaussians,
                      "num_visible": st.num_visible, "num_rendered": st.num_rendered,
                      "algorithmic_bytes_per_frame": b_alg,
                      "one_frame_at_a_time": {"frames_per_s": f, "frac_of_8TBs": b_alg * f / 1e9 / HBM_PEAK_GBS},
                      "headline_arrangement": {"frames_per_s": f_arr, "frac_of_8TBs": b_alg * f_arr / 1e9 / HBM_PEAK_GBS,
                                               "frames_per_step": B, "steps_in_flight": len(streams)}})
    out["visibility_sweep"] = sweep
    d = sweep[-1]
    out["dense_view"] = {
        "frames_per_s": d["one_frame_at_a_time"]["frames_per_s"], "frames_in_flight": 1,
        "frames_per_s_headline_arrangement": d["headline_arrangement"]["frames_per_s"],
        "frames_per_step": B, "steps_in_flight": len(streams),
        "num_visible": d["num_visible"], "num_rendered": d["num_rendered"],
        "visible_fraction": d["visible_fraction"], "algorithmic_bytes_per_frame": d["algorithmic_bytes_per_frame"],
        "frac_of_8TBs": d["one_frame_at_a_time"]["frac_of_8TBs"],
        "frac_of_8TBs_headline_arrangement": d["headline_arrangement"]["frac_of_8TBs"],
        "workload": f"same {raw.num} Gaussians, camera 1.5 m above the table centre looking down "
                    "(gsworld_amd.scenes.dense_view_camera), hipGraph replay"}
    # ---- what upstream render() adds in front of the rasterizer -----------------------------------------------
    rawd = raw.to(dev)
    cam = scenes.sensor_camera(name, W, H).to(dev)
    rp = FrameRenderer(dev, forward_only=True, want_radii=False)

    def packed():
        shs_ = torch.cat((rawd.features_dc, rawd.features_rest), dim=1)
        rp.render(cam, rawd.xyz, torch.sigmoid(rawd.opacity), shs=shs_, scales=torch.exp(rawd.scaling),
                  rotations=torch.nn.functional.normalize(rawd.rotation), bg=bg, rgb8_out=rgb8)

    def fused():
        rp.render(cam, rawd.xyz, rawd.opacity, shs=rawd.features_dc, shs_rest=rawd.features_rest, scales=rawd.scaling,
                  rotations=rawd.rotation, bg=bg, rgb8_out=rgb8,
                  param_space=RAW_OPACITY | RAW_SCALES | RAW_ROTATIONS)

    def plain():
        rp.render(cam, means, op, shs=shs, scales=sc, rotations=rot, bg=bg, rgb8_out=rgb8)

    res = {}
    for label, fn in (("pre_activated_inputs", plain), ("torch_sigmoid_exp_normalize_cat_per_frame", packed),
                      ("activations_and_split_sh_fused_into_preprocess", fused)):
        for _ in range(2):
            fn()
            rp.ensure_valid(fn)
        g = graphed(fn)
        res[label] = _time_frames(torch, g.replay, steps)
        del g
    out["upstream_packing"] = {"frames_per_s": res, "frames_in_flight": 1,
                               "workload": "headline scene and camera, the model AS GIVEN (no load-time layout: a drop-in "
                                           "call gets fresh tensors), one frame at a time, hipGraph replay"}
    del rp
    # ---- closed loop (configs[2] surrogate) -------------------------------------------------------------------
    cams = {"right_cam": scenes.sensor_camera(name, W, H),
            "wrist_cam": look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, W, H)}
    # robot-link poses: forward kinematics of the reference's xarm6 URDF along a seeded random-action rollout
    # (tests/golden/xarm6_rollout.npz, tools/make_xarm6_rollout.py); the two tracked objects random-walk
    rollout = cl.xarm6_rollout()
    parts, actors = cl.xarm6_rollout_parts(rollout)
    loop = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev)
    ep_len = 200
    poses = list(cl.rollout_poses(rollout, len(actors), steps=ep_len + 1, seed=0))
    pinned = [(M.pin_memory(), s.pin_memory()) for M, s in poses]

    # the wrist camera rides on the arm: the wrapper recomputes its cameras on every render (gs_world_wrapper.py:238),
    # so the surrogate moves it too (10 cm sweep over the episode) -- its frames then take the sampling path of the depth
    # sort every step, the fixed right_cam keeps its splitters
    def wrist_at(k):
        import math as _m

        a = 2.0 * _m.pi * k / ep_len
        v = look_at_view([0.55 - 0.10 * _m.sin(a), 0.35, 0.25 + 0.05 * _m.sin(2.0 * a)], [0.35, 0.05, 0.05], [0, 0, 1],
                         0.9715089, 0.7551448, W, H)
        v.world_view_transform = v.world_view_transform.pin_memory()
        v.full_proj_transform = v.full_proj_transform.pin_memory()
        v.camera_center = v.camera_center.pin_memory()
        return v

    wrists = [wrist_at(k) for k in range(ep_len + 1)]
    loop.reset(*pinned[0])
    loop.capture()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop.step(*pinned[0], cameras={"wrist_cam": wrists[0]})  # the episode's reset() frame pair
    for (M, s), w in zip(pinned[1:], wrists[1:]):
        loop.step(M, s, cameras={"wrist_cam": w})
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    overflow = any(x.overflow for x in loop.ensure_valid())
    # the same rollout with the POLICY IN THE LOOP: step k + 1 is issued only after the frames of step k have arrived
    # (step(ensure=True): a host wait per step; the frames stay on the device, as gs_world_wrapper.py:268-270 leaves them) --
    # what every closed loop other than a random-action rollout does
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for (M, s), w in zip(pinned, wrists):
        loop.step(M, s, cameras={"wrist_cam": w}, ensure=True)
    dt_policy = time.perf_counter() - t0
    # ---- the same two measurements with every block recomputed every frame (block_cache=False): what the frames cost when
    # nothing is kept from the step before (csrc/preprocess.hip prep_block_cached keeps, under the fixed right_cam, the blocks of
    # Gaussians that belong to no moving part -- their camera and pose are the previous frame's bit for bit)
    def loop_without(**kw):
        try:
            lp0 = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev, **kw)
            lp0.reset(*pinned[0])
            lp0.capture()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for (M, s), w in zip(pinned, wrists):
                lp0.step(M, s, cameras={"wrist_cam": w})
            torch.cuda.synchronize()
            dt0 = time.perf_counter() - t0
            t0 = time.perf_counter()
            for (M, s), w in zip(pinned, wrists):
                lp0.step(M, s, cameras={"wrist_cam": w}, ensure=True)
            dt0p = time.perf_counter() - t0
            same = all(torch.equal(a_, b_) for a_, b_ in zip(lp0.frames.values(), loop.frames.values()))
            del lp0
            return {"frames_per_s": (ep_len + 1) * len(cams) / dt0, "policy_in_loop_frames_per_s": (ep_len + 1) * len(cams) / dt0p,
                    "last_frames_identical_to_the_cached_loop_s": bool(same)}
        except Exception as ex:  # noqa: BLE001
            return {"error": f"{type(ex).__name__}: {ex}"}

    no_cache = loop_without(block_cache=False)
    # ... and with the block cache but every tile composited every frame (tile_reuse=False: csrc/render.hip leaves, under the
    # fixed right_cam, the tiles no recomputed Gaussian touches as the previous step's frame holds them)
    no_tile_reuse = loop_without(tile_reuse=False)
    # ---- SURVEY 8d's byte model for THIS loop: N, V, R of the reference's per-tile pipeline for both cameras at mid-episode
    # (step 100's poses and wrist camera; one default exact-mode frame per camera over the loop's own model and pose table)
    cl_alg = None
    try:
        from gsworld_amd._lib import RAW_ROTATIONS, RAW_SCALES

        k_mid = ep_len // 2
        loop.step(*pinned[k_mid], cameras={"wrist_cam": wrists[k_mid]}, ensure=True)
        per_cam = {}
        for nm, view in zip(loop.names, loop.cameras):
            fr_ = FrameRenderer(dev)
            fr_.render(view, loop.xyz, loop.opacity, shs=loop.features_dc, shs_rest=loop.features_rest, scales=loop.scaling,
                       rotations=loop.rotation, param_space=RAW_SCALES | RAW_ROTATIONS, bg=loop.bg,
                       parts=loop.op.parts_of_table(loop._table), exact=True)
            st_ = fr_.stats()
            per_cam[nm] = {"num_visible": st_.num_visible, "num_rendered": st_.num_rendered,
                           "algorithmic_bytes": st_.algorithmic_bytes(W, H)}
            del fr_
        b_step = sum(v["algorithmic_bytes"] for v in per_cam.values())
        cl_alg = {"per_camera_at_step": k_mid, "per_camera": per_cam, "algorithmic_bytes_per_step": b_step,
                  "frac_of_8TBs": b_step * ((ep_len + 1) / dt) / 8e12,
                  "policy_in_loop_frac_of_8TBs": b_step * ((ep_len + 1) / dt_policy) / 8e12,
                  "note": "B_alg = 48 N + 280 V + 64 R + 16 W H per frame (SURVEY 8d) with the reference's per-tile R, summed "
                          "over the step's two frames, x steps per second / 8 TB/s"}
        # the step's REAL HBM traffic, from the committed counter run of the same surrogate (tools/gpu_round6.sh clpmc:
        # 2 x FETCH_SIZE + WRITE_SIZE per kernel, one dispatch of each of the step's eleven kernels) x this run's steps per second
        import glob as _glob

        for path in sorted(_glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "round*",
                                                   "pmc_closed_loop.json"))):
            rec_ = json.load(open(path))
            cl_alg["counters"] = {"hbm_bytes_per_step": rec_["hbm_bytes_per_step"],
                                  "real_hbm_GBs": rec_["hbm_bytes_per_step"] * ((ep_len + 1) / dt) / 1e9,
                                  "real_over_algorithmic": rec_["hbm_bytes_per_step"] / b_step,
                                  "source": os.path.relpath(path, os.path.dirname(os.path.abspath(__file__))),
                                  "collected": rec_.get("collected")}
    except Exception as ex:  # noqa: BLE001
        cl_alg = {"error": f"{type(ex).__name__}: {ex}"}
    # ---- the same rollout with E environments per step (the wrapper's `for i in range(self.num_envs)`, gs_world_wrapper.py:
    # 241-242): E x 2 frames per gsr_forward_batch call, environment e playing the trajectory 17 e steps ahead
    env_sweep = {}
    for E in (2, 4, 8):
        try:
            lp = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev, num_envs=E)
            pe = [(M.pin_memory(), s_.pin_memory()) for M, s_ in cl.rollout_poses(rollout, len(actors), steps=ep_len + 1, seed=0, num_envs=E)]
            lp.reset(*pe[0])
            lp.capture()
            rec = {}
            for label, ensure in (("frames_per_s", False), ("policy_in_loop_frames_per_s", True)):
                torch.cuda.synchronize()
                t0e = time.perf_counter()
                for (M, s_), w in zip(pe, wrists):
                    lp.step(M, s_, cameras={"wrist_cam": w}, ensure=ensure)
                torch.cuda.synchronize()
                rec[label] = (ep_len + 1) * len(cams) * E / (time.perf_counter() - t0e)
            rec["frames_per_launch"], rec["overflow_frames"] = min(E * len(cams), lp.multi.set_frames), lp.overflow_frames()
            rec["sets_per_step"] = -(-E * len(cams) // lp.multi.set_frames)  # (each on a stream of its own, up to three)
            env_sweep[f"num_envs_{E}"] = rec
            del lp, pe
        except Exception as ex:  # noqa: BLE001
            env_sweep[f"num_envs_{E}"] = {"error": f"{type(ex).__name__}: {ex}"}
    # ---- a second surrogate in which the robot LOOKS like a robot (scenes.arm_tabletop_scene: the robot's Gaussians on the
    # links of the reference's URDF at the scan pose, 8 % of the model, two objects on the table) under the SAME rollout:
    # tabletop_scene scatters its part clusters over a 0.8 m cube that fills right_cam's view, and the rollout swings them
    # through the whole frame -- hardly a tile stays as it was.  Not a BASELINE configuration; beside configs[2]'s surrogate.
    arm_shaped = None
    try:
        raw_arm = scenes.arm_tabletop_scene(rollout["link_scan"], rollout["labels"], n=raw.num, seed=1)
        arm_shaped = {}
        for E in (1, 4, 8):
            pe = pinned if E == 1 else [(M.pin_memory(), s_.pin_memory())
                                        for M, s_ in cl.rollout_poses(rollout, len(actors), steps=ep_len + 1, seed=0, num_envs=E)]
            rec = {}
            for label, kw in (("kept", {}), ("every_tile_composited", {"tile_reuse": False}), ("nothing_kept", {"block_cache": False})):
                lp = cl.ClosedLoopRenderer(raw_arm, parts, cams, scaled_parts=actors, device=dev, num_envs=E, **kw)
                lp.reset(*pe[0])
                lp.capture()
                r_ = {}
                for key, ensure in (("frames_per_s", False), ("policy_in_loop_frames_per_s", True)):
                    torch.cuda.synchronize()
                    t0a = time.perf_counter()
                    for (M, s_), w in zip(pe, wrists):
                        lp.step(M, s_, cameras={"wrist_cam": w}, ensure=ensure)
                    torch.cuda.synchronize()
                    r_[key] = (ep_len + 1) * len(cams) * E / (time.perf_counter() - t0a)
                r_["overflow_frames"] = lp.overflow_frames()
                if label == "kept":
                    last = {n_: f.clone() for n_, f in lp.frames.items()}
                else:
                    r_["last_frames_identical_to_the_loop_that_keeps_s"] = all(torch.equal(last[n_], f) for n_, f in lp.frames.items())
                rec[label] = r_
                del lp
            arm_shaped[f"num_envs_{E}"] = rec
        arm_shaped["what"] = ("scenes.arm_tabletop_scene under the configs[2] rollout: ClosedLoopRenderer as shipped (blocks nothing "
                              "moved in keep their records, tiles nothing touched keep their pixels), with tile_reuse=False, with "
                              "block_cache=False; frames bit-identical between the three")
        del raw_arm
    except Exception as ex:  # noqa: BLE001
        arm_shaped = {"error": f"{type(ex).__name__}: {ex}"}
    # ---- the same rollout with consecutive steps in flight (PipelinedClosedLoop, depth 3: six frames instead of two).
    # configs[2] is a RANDOM-ACTION rollout: gsworld_rand_action_tabletop.py:107-133 never looks at its observations, so
    # step k + 1 may be enqueued while step k renders; a policy that needs frame k first gets the figure above.
    pipe_fps = pipe_ovf = None
    try:
        pipe = cl.PipelinedClosedLoop(raw, parts, cams, depth=3, scaled_parts=actors, device=dev)
        pipe.reset(*pinned[0])
        pipe.capture()
        torch.cuda.synchronize()
        t0p = time.perf_counter()
        pipe.step(*pinned[0], cameras={"wrist_cam": wrists[0]}, wait=False)
        for (M, s_), w in zip(pinned[1:], wrists[1:]):
            pipe.step(M, s_, cameras={"wrist_cam": w}, wait=False)
        torch.cuda.synchronize()
        pipe_fps = (ep_len + 1) * len(cams) / (time.perf_counter() - t0p)
        pipe_ovf = pipe.overflow_frames()
        del pipe
    except Exception as ex:  # noqa: BLE001
        pipe_fps = f"{type(ex).__name__}: {ex}"
    # the same steps through the wrapper's OWN glue (baseline leg, like cpu_baseline: oracle/wrapper_glue_ref.py restates
    # gs_world_wrapper.py:110-162, 232-275 op for op in torch -- deep copies, isin masks, masked write-backs, upstream
    # render()'s activations and SH concat) around this package's drop-in rasterizer in exact mode: what the loop costs
    # when only `diff_gaussian_rasterization` is swapped and the wrapper is left as it is
    ref_glue = None
    try:
        import types

        from gsworld_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
        from oracle import wrapper_glue_ref as wg

        rawd2 = raw.to(dev)
        model = types.SimpleNamespace(_xyz=rawd2.xyz, _scaling=rawd2.scaling, _rotation=rawd2.rotation,
                                      _opacity=rawd2.opacity.reshape(-1, 1, 1), _semantics=rawd2.semantics,
                                      _features_dc=rawd2.features_dc, _features_rest=rawd2.features_rest)
        cams_d = {k: v.to(dev) for k, v in cams.items()}

        def rasterize(view, means3D, shs_, opacities, scales, rotations, bg_):
            rs = GaussianRasterizationSettings(view.image_height, view.image_width, view.tanfovx, view.tanfovy, bg_, 1.0,
                                               view.world_view_transform, view.full_proj_transform, 3,
                                               view.camera_center, False, False, False)
            return GaussianRasterizer(rs)(means3D=means3D, means2D=torch.zeros_like(means3D), shs=shs_,
                                          opacities=opacities, scales=scales, rotations=rotations)[0]

        n_ref = 12
        wg.render_step(model, parts, cams_d, *poses[0], rasterize, actors)
        torch.cuda.synchronize()
        t0r = time.perf_counter()
        for M, s_ in poses[1:1 + n_ref]:
            wg.render_step(model, parts, cams_d, M, s_, rasterize, actors)
        torch.cuda.synchronize()
        ref_glue = n_ref * len(cams) / (time.perf_counter() - t0r)
        del model, rawd2
    except Exception as ex:  # noqa: BLE001
        ref_glue = f"{type(ex).__name__}: {ex}"
    out["closed_loop"] = {
        "reference_glue_frames_per_s": ref_glue,
        "three_steps_in_flight": {"frames_per_s": pipe_fps, "overflow_frames": pipe_ovf,
                                "what": "PipelinedClosedLoop(depth=3): steps k + 1, k + 2 enqueued while step k renders (legitimate "
                                        "for a random-action / scripted rollout, whose actions do not depend on the frames)"},
        "frames_per_s": (ep_len + 1) * len(cams) / dt, "steps_per_s": (ep_len + 1) / dt,
        "policy_in_loop_frames_per_s": (ep_len + 1) * len(cams) / dt_policy,
        "policy_in_loop_steps_per_s": (ep_len + 1) / dt_policy,
        "environments_per_step": env_sweep,
        "arm_shaped": arm_shaped,
        "block_cache": {"on": True, "what": "inference frames of ClosedLoopRenderer keep a block of 256 Gaussians as the previous "
                        "frame on the state computed it when settings, camera and the block's pose row are that frame's bit for "
                        "bit (GSR_MODEL_VERSION; the headline and every other figure of this line never pass a model version: "
                        "nothing is kept there)", "without": no_cache},
        "tile_reuse": {"on": True, "what": "... and their compositor leaves a 16 x 16 tile of the loop's own uint8 frame as the "
                       "previous step wrote it when camera and background are that step's and no recomputed Gaussian touches "
                       "the tile now or touched it then (GSR_FRAME_KEPT)", "without": no_tile_reuse},
        "roofline": cl_alg,
        "frames_per_launch": len(cams),
        "frames": (ep_len + 1) * len(cams), "overflow": overflow,
        # counted by the frames themselves on the device: 0 = every one of the 402 frames fitted its binning capacity
        "overflow_frames": loop.overflow_frames(),
        "workload": f"BASELINE.json configs[2] surrogate: 1 reset + {ep_len} steps x {len(cams)} cameras {W}x{H}, "
                    f"{raw.num} Gaussians, {len(parts)} moving parts (robot links: FK of the reference's xarm6 URDF along a seeded "
                    "random-action rollout, kinematic PD stand-in instead of PhysX; objects: seeded random walk), per step: "
                    "pose + wrist-camera upload (the wrist camera moves every step), device-side pose table, rigid transform inside "
                    "preprocess, both frames through one gsr_forward_batch call, one hipGraph replay that stages the step's host values itself; frames_per_s: steps "
                    "enqueued without waiting (a random-action rollout), policy_in_loop_*: every step waited for"}
    del loop
    # ---- the headline scene under a camera that MOVES every step (no kept splitters / cuts / static-camera reuse) ----
    base_cam = scenes.sensor_camera(name, W, H)

    def orbit(k):  # the sensor pose turned by up to +-2 degrees about the world z axis through the table centre
        import math as _m

        a = _m.radians(2.0) * _m.sin(2.0 * _m.pi * k / 97.0)
        Rz = torch.tensor([[_m.cos(a), -_m.sin(a), 0, 0], [_m.sin(a), _m.cos(a), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
        wvt = (Rz @ base_cam.world_view_transform)  # world_view_transform is W2C^T: points are row vectors
        proj = base_cam.world_view_transform.inverse() @ base_cam.full_proj_transform
        return wvt.contiguous().pin_memory(), (wvt @ proj).contiguous().pin_memory(), \
            wvt.inverse()[3, :3].contiguous().pin_memory()

    poses_mv = [orbit(k) for k in range(97)]
    mv_cams = [scenes.sensor_camera(name, W, H).to(dev) for _ in streams]
    f_mv, ovf = _arrangement_fps(torch, dev, mv_cams, l_kw, l_means, l_op, B, streams, steps, H, W, poses=poses_mv)
    out["moving_camera"] = {
        "frames_per_s": f_mv, "frames_per_step": B, "steps_in_flight": len(streams), "overflow": ovf,
        "workload": "headline scene, the sensor camera turned by a different angle (+-2 degrees about the world z axis) "
                    "on EVERY step: three small H2D copies per step, no static-camera reuse in the depth sort or the "
                    "placement; the headline's arrangement, hipGraph replay"}
    # ---- simple_knn distCUDA2 at the headline model size ---------------------------------------
    try:
        import numpy as np
        from scipy.spatial import cKDTree

        from gsworld_amd.knn import distCUDA2

        pts = raw.xyz.to(dev)
        for _ in range(2):
            d2 = distCUDA2(pts)
        torch.cuda.synchronize()
        d2 = torch.empty((raw.num,), dtype=torch.float32, device=dev)
        distCUDA2(pts, out=d2)  # (the workspace is allocated here, once, outside the timed calls)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            distCUDA2(pts, out=d2)
        e1.record()
        torch.cuda.synchronize()
        knn_ms = e0.elapsed_time(e1) / 5
        xyz = raw.xyz.numpy().astype(np.float64)
        sub = np.random.default_rng(0).choice(raw.num, 20_000, replace=False)
        dd, _ = cKDTree(xyz).query(xyz[sub], k=4)
        want = (dd[:, 1:] ** 2).mean(1)
        got = d2.cpu().numpy()[sub].astype(np.float64)
        out["knn_dist2"] = {
            "ms": knn_ms, "points": raw.num, "points_per_s": raw.num / (knn_ms * 1e-3),
            "max_rel_err_vs_ckdtree_20k_subsample": float(np.max(np.abs(got - want) / np.maximum(want, 1e-12))),
            "bound": "distance evaluations (VALU), not bytes: the search kernel tests every point against the points of "
                     "the boxes its running third-nearest distance still reaches -- 16 B of compulsory traffic per point "
                     "say nothing about it; profiles/round6/kernel_stats_knn_dist2.csv holds the per-kernel times "
                     "(knn_search_kernel dominant)",
            "workload": "gsr_knn_dist2 (simple_knn distCUDA2) on the headline scene's 1.47 M means, workspace allocated "
                        "once outside the timed calls; exact 3-NN, checked against scipy.spatial.cKDTree on a 20 k subsample"}
    except Exception as ex:  # noqa: BLE001
        out["knn_dist2"] = {"error": f"{type(ex).__name__}: {ex}"}
    # ---- BASELINE.json configs[4]: the training step ---------------------------------------------------------------------
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_train

        for label, fused in (("upstream_packing", False), ("fused_packing", True)):
            rec = bench_train.run(steps=30, warmup=5, fused=fused, device=str(dev))
            out.setdefault("train_step", {})[label] = {
                "ms_per_step": rec["ms_per_step"], "it_per_s": rec["value"],
                "algorithmic_bytes_per_step": rec["roofline"]["algorithmic_bytes_per_step"],
                "frac_of_8TBs": rec["roofline"]["frac"],
                "real_hbm_GBs": (rec["roofline"].get("counters") or {}).get("real_hbm_GBs"),
                "real_hbm_source": (rec["roofline"].get("counters") or {}).get("source"),
                "num_visible": rec["config"]["num_visible"],
                "num_rendered": rec["config"]["num_rendered"], "grads_finite": rec["config"]["grads_finite"],
                "variant": rec["config"]["parameter_packing"], "loss": rec["config"]["loss"]}
        out["train_step"]["workload"] = rec["config"]["workload"]
    except Exception as ex:  # noqa: BLE001
        out["train_step"] = {"error": f"{type(ex).__name__}: {ex}"}
    out["parity"] = parity_record(dev)
    return out


def parity_record(dev):
    """The checker's verdict on THIS build, in the line: the 8 scenes of BASELINE.json configs[3] at FULL size (1,468,850
    Gaussians each) through the drop-in rasterizer against the CPU oracle (oracle/gs_oracle.c; checker use, like
    cpu_baseline) -- per scene the worst pixel INCLUDING the ones the oracle flags as borderline (an alpha >= 1/255 or
    T >= 1e-4 decision within an exp() ulp of its threshold), the worst off them, how many are flagged.  north_star's
    bar is 1e-4."""
    try:
        import numpy as np

        from gsworld_amd import scenes
        from tests import helpers as hp

        worst_all, worst_off, border, per, per_off = 0.0, 0.0, 0, {}, {}
        for i, n in enumerate(scenes.SCENE_NAMES):
            raw, cam = scenes.tabletop_scene(n, seed=1 + i), scenes.sensor_camera(n)
            inp, st = hp.np_inputs(raw, cam), hp.oracle_settings(cam)
            bg = np.zeros(3, np.float32)
            o = hp.oracle_forward(inp, st, bg)
            g = hp.gpu_forward(inp, st, bg, device=str(dev))
            d = np.abs(g["color"] - o["color"]).max(0)
            b = o["borderline"] != 0
            per[n], per_off[n] = float(d.max()), float(d[~b].max())
            worst_all, worst_off, border = max(worst_all, per[n]), max(worst_off, per_off[n]), border + int(b.sum())
            del raw, inp, o, g
        # configs[4]: all eight gradients at full size (500 k Gaussians, 800 x 800) against the backward oracle (binary64
        # sums) -- the worst element's normalised error; the bar is 2e-3 (tests/test_backward_gpu.py)
        grads = None
        try:
            from tests import helpers_bwd as hb

            raw5 = scenes.random_scene_camera_frame(500_000, seed=5, near_fraction=0.0)
            rep = hb.run_case(500_000, 800, 800, seed=5, scale_boost=0.0, raw=raw5, **hb.SUITE_TOLERANCES)
            grads = {"worst_normalised_error": max(v["max_norm_err"] for v in rep.values()),
                     "per_output": {k: v["max_norm_err"] for k, v in rep.items()},
                     "fraction_within_2e-3": min(v["frac_within"] for v in rep.values()),
                     "workload": "configs[4]: 500 k Gaussians, 800 x 800, all eight gradients against oracle/gs_oracle.c"}
        except Exception as ex:  # noqa: BLE001
            grads = {"error": f"{type(ex).__name__}: {ex}"}
        return {"worst_pixel_all_scenes": worst_all, "worst_pixel_off_borderline": worst_off,
                "config5_gradients": grads,
                "borderline_pixels": border, "pixels": 8 * 640 * 480, "per_scene_worst": per,
                "per_scene_worst_off_borderline": per_off,
                "against": "oracle/gs_oracle.c (CPU restatement; UNPINNED against the CUDA reference, DESIGN.md section 2)",
                "workload": "8 scenes of configs[3] at full size (1,468,850 Gaussians each), 640x480 sensor camera, "
                            "default frames"}
    except Exception as ex:  # noqa: BLE001
        return {"error": f"{type(ex).__name__}: {ex}"}


def cpu_baseline_config0():
    """BASELINE.json configs[0]: 100 k random Gaussians, one 256x256 camera -- north_star's "PyTorch-CPU fallback render
    timed on the host cores": oracle/torch_cpu_render.py (vectorised PyTorch, CPU) and oracle/gs_oracle.c beside it.
    torch threads are capped at 32: with every host thread of a 256-thread box the op-by-op torch render spends its time
    in thread-pool hand-offs and does not finish in minutes."""
    import numpy as np
    import torch

    from gsworld_amd import scenes
    from oracle import gs_oracle as go
    from oracle import torch_cpu_render as tcr

    raw = scenes.random_scene_camera_frame(100_000, seed=0)
    cam = scenes.identity_camera(256, 256, 60.0)
    means, shs, op, sc, rot = raw.activated()
    cores = os.cpu_count() or 1
    threads = min(32, cores)
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        ts = []
        for k in range(3):
            t0 = time.perf_counter()
            tcr.render(means, shs, op.reshape(-1), sc, rot, cam.world_view_transform, cam.full_proj_transform,
                       cam.camera_center, torch.zeros(3), cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy)
            ts.append(time.perf_counter() - t0)
    finally:
        torch.set_num_threads(old)
    t_torch = sorted(ts[1:])[0]
    st = go.Settings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy)
    go.set_threads(threads)  # (a 256x256 frame has 256 tiles: more threads than that only add hand-off time)
    a = (st, np.zeros(3, np.float32), means.numpy(), shs.numpy(), None, op.numpy().reshape(-1), sc.numpy(), rot.numpy(),
         None, cam.world_view_transform.numpy().reshape(-1), cam.full_proj_transform.numpy().reshape(-1),
         cam.camera_center.numpy())
    go.forward(*a)
    tc = []
    for _ in range(5):
        t0 = time.perf_counter()
        go.forward(*a)
        tc.append(time.perf_counter() - t0)
    go.set_threads(cores)
    return {"workload": "BASELINE.json configs[0]: 100000 random Gaussians, 256x256",
            "pytorch_cpu": {"value": 1.0 / t_torch, "unit": "frames/s", "cores": threads, "kind": "port",
                            "sample": "best of 2 frames after 1 warm-up, oracle/torch_cpu_render.py"},
            "c_openmp": {"value": 1.0 / sorted(tc)[len(tc) // 2], "unit": "frames/s", "cores": threads, "kind": "port",
                         "sample": "median of 5 frames after 1 warm-up, oracle/gs_oracle.c"}}


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
