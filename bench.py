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
    ap.add_argument("--no-graph", action="store_true", help="do not capture the frame in a hipGraph")
    ap.add_argument("--render-bpc", type=int, default=0, help="persistent compositing workgroups per CU (0 = library default)")
    ap.add_argument("--no-layout", action="store_true",
                    help="render the model in the order it was given, without block culling (gsworld_amd/layout.py)")
    ap.add_argument("--blocks", type=int, default=5, help="timed blocks of --steps frames each; value = their median")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="independent frames in flight, each on its own HIP stream with its own renderer state "
                         "(GSWorld renders 2 cameras per step; 1 = strictly one frame at a time).  0 (default): 3 or 4, "
                         "whichever a short trial on this box finds faster (see pick_lanes)")
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

    trial = None
    if args.in_flight > 0:
        S = args.in_flight
        lane_streams = [torch.cuda.Stream(dev) for _ in range(S)] if S > 1 else None
    else:
        S, lane_streams, trial = pick_lanes(torch, dev, cam, (m_means, m_shs, m_op, m_sc, m_rot, lay), bg, W, H)
        if world > 1:
            # every rank must gather batches of the same size: rank 0's choice holds for all (each rank keeps the
            # streams of both candidates)
            choice = [S]
            dist.broadcast_object_list(choice, src=0)
            S, lane_streams = int(choice[0]), trial["streams"][int(choice[0])]
            trial["chosen"] = S
        trial.pop("streams", None)
        args.in_flight = S  # (the moving-camera extra runs with as many lanes)
    # compositing workgroups per CU: the library default (6: every tile quadrant resident at once) is the optimum
    # both for one frame and for 3 frames in flight (tools/sweep_bench.sh); the flag is for sweeps
    bpc = args.render_bpc
    if bpc:
        from gsworld_amd import debug as dbg

        dbg.set_render_variant(4, bpc)
    K_g = max(1, args.gather_every)
    K_g = (K_g + S - 1) // S * S  # a frame slot belongs to exactly one lane: lane = slot % S
    fg = gd.FrameGather(H, W, batch=K_g, device=dev, world=world, buffers=2 if world > 1 else 1,
                        collective=args.collective, timing=world > 1)
    n_slots = fg.num_slots
    # inference frames (GsrSettings.forward_only): GSWorld's loop keeps ["render"] only (gs_world_wrapper.py:266-270) --
    # nothing a backward would read is written, instances are binned per 2 x 1 super-tile; the image is bit-identical
    # (tests/test_renderer_gpu.py, and checked against a default frame right below)
    rs_ = [FrameRenderer(dev, forward_only=True, want_radii=False) for _ in range(S)]
    lanes = lane_streams if S > 1 else [torch.cuda.current_stream(dev)]
    r = rs_[0]
    # N, V, R of SURVEY.md 8d's byte model are those of the reference's own per-tile pipeline: one default frame gives them
    ref_r = FrameRenderer(dev)
    ref_rgb8 = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
    ref_r.render(cam, means, op, shs=shs, scales=sc, rotations=rot, bg=bg, rgb8_out=ref_rgb8, exact=True)
    true_stats = ref_r.stats()
    del ref_r

    def frame(slot, lane=None):
        rr = rs_[slot % S if lane is None else lane]
        # the uint8 HWC frame GSWorld consumes is written by the compositor itself (GsrOutputs.out_rgb8)
        rr.render(cam, m_means, m_op, shs=m_shs, scales=m_sc, rotations=m_rot, bg=bg, rgb8_out=fg.frames[slot], layout=lay)

    # exact-mode frame sizes the binning capacity from the real R; then check the no-sync path is valid
    for l in range(S):
        for _ in range(2):
            frame(l, l)
            rs_[l].ensure_valid(lambda l=l: frame(l, l))
    torch.cuda.synchronize()
    if not torch.equal(fg.frames[0], ref_rgb8):
        raise SystemExit("forward_only frame differs from the default frame: result invalid")

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
    # exactly K steps between barriers, `--blocks` times over: at ~0.1 ms a frame one block of the driver's K is a few
    # milliseconds, so the line carries the median block and the spread
    block_s = []
    for b in range(max(1, args.blocks)):
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(b * args.steps + i)
        barrier()
        block_s.append(time.perf_counter() - t0)
    elapsed = sorted(block_s)[len(block_s) // 2]
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

    own_elapsed = elapsed
    t_max = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    per_rank = [own_elapsed]
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        every = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(every, torch.tensor([own_elapsed], dtype=torch.float64, device=dev))
        per_rank = [float(t.item()) for t in every]
    elapsed = float(t_max.item())
    fps = world * args.steps / elapsed

    extras = {}
    if rank == 0 and world == 1 and not args.no_extras:
        extras = secondary_measurements(args, dev, raw, name, (means, shs, op, sc, rot),
                                        (m_means, m_shs, m_op, m_sc, m_rot, lay), bg,
                                        lane_streams if S > 1 else None)
    # strictly one frame at a time (hipGraph replay of one lane), for the line's config: the figure a caller sees who
    # needs frame k before it can ask for frame k + 1
    one_fps = None
    if rank == 0 and world == 1 and graph is not None:
        one_fps = _time_frames(torch, graph[0].replay, min(args.steps, 300))

    if rank == 0:
        binned = stats.num_rendered  # instances actually placed (super-tile lists)
        stats = true_stats           # the byte model counts the reference's per-tile instances
        b_alg = stats.algorithmic_bytes(W, H)
        render_ms = stage_ms[-1]
        render_bytes = 40 * stats.num_rendered + 16 * W * H  # 40 B per composited instance + outputs
        ach = render_bytes / (render_ms * 1e-3) / 1e9 if render_ms > 0 else 0.0
        traffic = valu_frac = valu_frac_213 = None
        traffic_source = "none"
        pmc = os.path.join(ROOT, "profiles", "pmc_render.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                traffic = rec.get("hbm_bytes_per_launch")
                # NOT measured in this run (PMC counters need a rocprofv3 session): a committed counter run of the same
                # frame.  The record carries the hash of render.hip it was taken with; a kernel edited since then is
                # flagged instead of silently quoted.
                import hashlib
                sha = hashlib.sha256(open(os.path.join(ROOT, "gsworld_amd", "csrc", "render.hip"), "rb").read()).hexdigest()[:16]
                stale = rec.get("render_hip_sha16") != sha
                traffic_source = (f"committed rocprofv3 --pmc run {os.path.relpath(pmc, ROOT)} "
                                  f"({rec.get('collected', 'undated')}; render.hip "
                                  f"{'CHANGED since then: stale' if stale else 'unchanged since then'})")
                # what actually bounds the compositor: VALU issue.  SQ_INSTS_VALU wave-instructions (committed PMC
                # run of the same frame) x 4 cycles on 256 CUs x 4 SIMDs at 2.4 GHz, over the kernel time measured now
                insts = rec.get("counters", {}).get("SQ_INSTS_VALU")
                if insts and render_ms > 0:
                    valu_frac = insts * 4.0 / (256 * 4 * 2.4e9 * render_ms * 1e-3)
                    # (per-quadrant s_memtime stamps put the shader clock under this kernel at 2.13 GHz, not the nominal
                    #  2.4: profiles/round4/stream_stamps_all_static.txt)
                    valu_frac_213 = insts * 4.0 / (256 * 4 * 2.13e9 * render_ms * 1e-3)
            except Exception:  # noqa: BLE001
                traffic = valu_frac = valu_frac_213 = None
        # the same kernel's average in the committed rocprofv3 --kernel-trace --stats run of this command (profiles/)
        rocprof_ms = None
        try:
            import csv
            import glob

            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*", "kernel_stats_bench_one_frame_in_flight.csv"))):
                for row in csv.DictReader(open(path)):
                    if "render_stream_kernel<true>" in row["Name"]:
                        rocprof_ms = {"file": os.path.relpath(path, ROOT), "avg_ms": float(row["AverageNs"]) * 1e-6}
        except Exception:  # noqa: BLE001
            rocprof_ms = None
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
                "frame_mode": "forward_only (inference: super-tile binning, no backward-only writes; image bit-identical "
                              "to the default frame, checked in this run)",
                "sh_degree": 3, "launch": "hipGraph replay" if graph is not None else "eager",
                "frames_in_flight": S,
                "frames_in_flight_trial": trial,
                # the same frame, strictly one at a time / under a camera that turns on every frame (extras below)
                "one_frame_in_flight_frames_per_s": one_fps,
                "moving_camera_frames_per_s": extras.get("moving_camera", {}).get("frames_per_s"),
                "model_layout": ("Morton order per size class + block bounds built once per scene "
                                 f"(gsworld_amd/layout.py, {layout_s:.3f} s on this box, outside the timed region): "
                                 "preprocess skips the blocks of 256 Gaussians no tile can see; image, radii and tie "
                                 "order unchanged") if lay is not None else "model as given (--no-layout)",
                "timed_blocks": {"blocks": len(block_s), "steps_each": args.steps,
                                 "frames_per_s": [world * args.steps / t for t in block_s],
                                 "value_is": "median block"},
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
                "per_rank_frames_per_s": [args.steps / t for t in per_rank],
            },
            "roofline": {
                # what binds this kernel is VALU issue (valu_issue_frac below); achieved / peak / frac are the HBM figures
                # the contract asks for (algorithmic bytes per launch over the kernel time, against 8 TB/s)
                "bound": "valu", "kernel": "render_stream_kernel", "achieved": ach, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": render_bytes, "kernel_ms": render_ms,
                "kernel_ms_rocprof_committed": rocprof_ms,
                "valu_issue_frac": valu_frac, "valu_issue_frac_at_measured_2.13GHz": valu_frac_213,
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
        out.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(raw, cam_cpu, args.cpu_frames)
            if not args.no_extras:
                out["cpu_baseline_config0"] = cpu_baseline_config0()
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pick_lanes(torch, dev, cam, model, bg, W, H):
    """How many frames to keep in flight, decided by a short trial on THIS box: 3 lanes on the first three streams the
    process uses, or 4 lanes on the next four.  Round 4 measured (tools/ab_frame.py --pre-streams, profiles/round4/
    NOTES_compositor_scheduling.md): the HIP runtime treats the first three user streams of a process and all later ones as
    two classes -- a group of lanes that lies inside one class overlaps its frames 2.3x, a group that straddles the two
    1.5x (7.8-9.5 k instead of 11.2-12 k frames/s) -- and four lanes of the later class beat three of the first by ~4 %
    on most boxes.  Instead of relying on either observation the bench measures both candidates (150 frames each, same
    frame, hipGraph replay) and takes the faster.  -> (lanes, streams, record of the trial)."""
    from gsworld_amd.renderer import FrameRenderer

    means, shs, op, sc, rot, lay = model
    rec, best = {"streams": {}}, None
    for S in (3, 4):
        streams = [torch.cuda.Stream(dev) for _ in range(S)]
        rs = [FrameRenderer(dev, forward_only=True, want_radii=False) for _ in range(S)]
        outs = [torch.zeros((H, W, 3), dtype=torch.uint8, device=dev) for _ in range(S)]
        fns = [(lambda l=l: rs[l].render(cam, means, op, shs=shs, scales=sc, rotations=rot, bg=bg, rgb8_out=outs[l],
                                         layout=lay)) for l in range(S)]
        graphs = []
        for l in range(S):
            for _ in range(2):
                fns[l]()
                rs[l].ensure_valid(fns[l])
            streams[l].wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(streams[l]):
                fns[l]()
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=streams[l]):
                fns[l]()
            graphs.append(g)
        torch.cuda.synchronize(dev)

        def run(n):
            for i in range(n):
                with torch.cuda.stream(streams[i % S]):
                    graphs[i % S].replay()
            torch.cuda.synchronize(dev)

        run(40)
        t0 = time.perf_counter()
        run(150)
        fps = 150 / (time.perf_counter() - t0)
        rec[f"{S}_lanes_frames_per_s"] = fps
        rec["streams"][S] = streams
        if best is None or fps > best[0]:
            best = (fps, S, streams)
        del graphs, rs, outs
    rec["chosen"] = best[1]
    return best[1], best[2], rec


def _time_frames(torch, enqueue, steps, warmup=10):
    for _ in range(warmup):
        enqueue()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        enqueue()
    torch.cuda.synchronize()
    return steps / (time.perf_counter() - t0)


def secondary_measurements(args, dev, raw, name, model, laid, bg, lane_streams=None):
    """SURVEY.md 8d's second numbers, N = 1 only, outside the headline's timed region, each a few hundred frames:
    * dense_view: the same scene and N from a camera that sees V = 0.6 N of it (8d's worked example; right_cam sees 0.12 N);
    * upstream_packing: the rasterizer figure INCLUDING what upstream render() does per frame before it
      (sigmoid / exp / normalize over the model + cat(dc, rest) -> (N,16,3)), and the same frame with those four passes
      fused into preprocess (raw parameters + split SH, GsrInputs.param_space / shs_rest);
    * closed_loop: BASELINE.json configs[2] surrogate -- 1 reset + 200 steps x 2 cameras through
      gsworld_amd.closed_loop (pose upload, fused transform, both frames, one hipGraph replay per step)."""
    import torch

    from gsworld_amd import closed_loop as cl, scenes
    from gsworld_amd._lib import RAW_OPACITY, RAW_ROTATIONS, RAW_SCALES
    from gsworld_amd.camera import look_at_view
    from gsworld_amd.renderer import FrameRenderer

    out = {}
    means, shs, op, sc, rot = model                        # the model as given (what a drop-in call hands over)
    l_means, l_shs, l_op, l_sc, l_rot, lay = laid          # the headline's layout (== model with --no-layout)
    W, H = args.width, args.height
    steps = min(args.steps, 200)
    rgb8 = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)

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

    # ---- dense view ---------------------------------------------------------------------------------------------
    cam_d = scenes.dense_view_camera(name, W, H).to(dev)
    rd = FrameRenderer(dev, forward_only=True, want_radii=False)
    fr = lambda: rd.render(cam_d, l_means, l_op, shs=l_shs, scales=l_sc, rotations=l_rot, bg=bg, rgb8_out=rgb8,  # noqa: E731
                           layout=lay)
    for _ in range(2):
        fr()
        st = rd.ensure_valid(fr)
    g = graphed(fr)
    f = _time_frames(torch, g.replay, steps)
    if rd.ensure_valid(fr).overflow:
        raise SystemExit("dense view: capacity overflow")
    # the same view the way the headline is measured: one frame per lane in flight on the headline's streams.  One frame
    # at a time the dense compositor ends with a few quadrants alone on the chip (a wave is a latency chain: 220 cycles
    # per unit of work alone against 76 when five share a SIMD, profiles/round4/NOTES_compositor_scheduling.md); with
    # other frames in flight that tail is filled
    f_lanes = None
    if lane_streams:
        S = len(lane_streams)
        rds = [FrameRenderer(dev, forward_only=True, want_radii=False) for _ in range(S)]
        outs = [torch.zeros((H, W, 3), dtype=torch.uint8, device=dev) for _ in range(S)]
        fns = [(lambda l=l: rds[l].render(cam_d, l_means, l_op, shs=l_shs, scales=l_sc, rotations=l_rot, bg=bg,
                                          rgb8_out=outs[l], layout=lay)) for l in range(S)]
        graphs = []
        for l in range(S):
            for _ in range(2):
                fns[l]()
                rds[l].ensure_valid(fns[l])
            lane_streams[l].wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(lane_streams[l]):
                fns[l]()
            torch.cuda.synchronize(dev)
            gl = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gl, stream=lane_streams[l]):
                fns[l]()
            graphs.append(gl)
        torch.cuda.synchronize(dev)
        k = [0]

        def enqueue_lane():
            with torch.cuda.stream(lane_streams[k[0] % S]):
                graphs[k[0] % S].replay()
            k[0] += 1

        f_lanes = _time_frames(torch, enqueue_lane, steps, warmup=4 * S)
        if any(r_.ensure_valid(fn_).overflow for r_, fn_ in zip(rds, fns)):
            raise SystemExit("dense view (lanes): capacity overflow")
        del graphs, rds, outs
    rt = FrameRenderer(dev)  # N, V, R of the byte model: the reference's per-tile pipeline (one default frame)
    rt.render(cam_d, means, op, shs=shs, scales=sc, rotations=rot, bg=bg, exact=True)
    st = rt.stats()
    del rt
    b_alg = st.algorithmic_bytes(W, H)
    out["dense_view"] = {
        "frames_per_s": f, "frames_in_flight": 1,
        "frames_per_s_headline_lanes": f_lanes, "headline_lanes": len(lane_streams) if lane_streams else None,
        "num_visible": st.num_visible, "num_rendered": st.num_rendered,
        "visible_fraction": st.num_visible / st.num_gaussians, "algorithmic_bytes_per_frame": b_alg,
        "frac_of_8TBs": b_alg * f / 1e9 / HBM_PEAK_GBS,
        "workload": f"same {st.num_gaussians} Gaussians, camera 1.5 m above the table centre looking down "
                    "(gsworld_amd.scenes.dense_view_camera), one frame at a time, hipGraph replay"}
    del rd, g
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
        "frames": (ep_len + 1) * len(cams), "overflow": overflow,
        # counted by the frames themselves on the device: 0 = every one of the 402 frames fitted its binning capacity
        "overflow_frames": loop.overflow_frames(),
        "workload": f"BASELINE.json configs[2] surrogate: 1 reset + {ep_len} steps x {len(cams)} cameras {W}x{H}, "
                    f"{raw.num} Gaussians, {len(parts)} moving parts (robot links: FK of the reference's xarm6 URDF along a seeded "
                    "random-action rollout, kinematic PD stand-in instead of PhysX; objects: seeded random walk), per step: "
                    "pose + wrist-camera upload (the wrist camera moves every step), device-side pose table, rigid transform inside "
                    "preprocess, both frames, one hipGraph replay"}
    del loop
    # ---- the headline scene under a camera that MOVES every frame (no kept splitters / cuts / static-camera reuse) ----
    S_mv = max(1, args.in_flight)
    mv_r = [FrameRenderer(dev, forward_only=True, want_radii=False) for _ in range(S_mv)]
    # (on the headline's own lane streams: a fresh group of streams may straddle the runtime's two stream classes --
    #  pick_lanes -- and overlap its frames badly: 8.4 k instead of 10.7 k frames/s was measured that way)
    mv_st = list(lane_streams) if lane_streams and len(lane_streams) == S_mv else [torch.cuda.Stream(dev) for _ in range(S_mv)]
    mv_cam = [scenes.sensor_camera(name, W, H).to(dev) for _ in range(S_mv)]
    mv_out = [torch.zeros((H, W, 3), dtype=torch.uint8, device=dev) for _ in range(S_mv)]
    base_cam = scenes.sensor_camera(name, W, H)

    def orbit(k):  # the sensor pose turned by up to +-2 degrees about the world z axis through the table centre
        import math as _m

        a = _m.radians(2.0) * _m.sin(2.0 * _m.pi * k / 97.0)
        Rz = torch.tensor([[_m.cos(a), -_m.sin(a), 0, 0], [_m.sin(a), _m.cos(a), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
        wvt = (Rz @ base_cam.world_view_transform)  # world_view_transform is W2C^T: points are row vectors
        proj = base_cam.world_view_transform.inverse() @ base_cam.full_proj_transform
        return wvt.contiguous().pin_memory(), (wvt @ proj).contiguous().pin_memory(), \
            wvt.inverse()[3, :3].contiguous().pin_memory()

    poses_mv = [orbit(k) for k in range(steps + 16)]
    mv_fn = [(lambda l=l: mv_r[l].render(mv_cam[l], l_means, l_op, shs=l_shs, scales=l_sc, rotations=l_rot, bg=bg,
                                        rgb8_out=mv_out[l], layout=lay)) for l in range(S_mv)]
    mv_g = []
    for l in range(S_mv):
        for _ in range(2):
            mv_fn[l]()
            mv_r[l].ensure_valid(mv_fn[l])
        with torch.cuda.stream(mv_st[l]):
            mv_fn[l]()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=mv_st[l]):
            mv_fn[l]()
        mv_g.append(g)

    def mv_step(k):
        l = k % S_mv
        with torch.cuda.stream(mv_st[l]):
            wvt, full, center = poses_mv[k]
            mv_cam[l].world_view_transform.copy_(wvt, non_blocking=True)
            mv_cam[l].full_proj_transform.copy_(full, non_blocking=True)
            mv_cam[l].camera_center.copy_(center, non_blocking=True)
            mv_g[l].replay()

    for k in range(10):
        mv_step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        mv_step(10 + k)
    torch.cuda.synchronize()
    f_mv = steps / (time.perf_counter() - t0)
    ovf = any(x.stats().overflow for x in mv_r)
    out["moving_camera"] = {
        "frames_per_s": f_mv, "frames_in_flight": S_mv, "overflow": ovf,
        "workload": "headline scene, the sensor camera turned by a different angle (+-2 degrees about the world z axis) "
                    "on EVERY frame: three small H2D copies per frame, no static-camera reuse in the depth sort or the "
                    "placement; hipGraph replay"}
    del mv_r, mv_g
    # ---- simple_knn distCUDA2 at the headline model size ---------------------------------------
    try:
        import numpy as np
        from scipy.spatial import cKDTree

        from gsworld_amd.knn import distCUDA2

        pts = raw.xyz.to(dev)
        for _ in range(2):
            d2 = distCUDA2(pts)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            d2 = distCUDA2(pts)
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
            "algorithmic_GBs": 16.0 * raw.num / (knn_ms * 1e-3) / 1e9,  # 12 B read + 4 B written per point
            "max_rel_err_vs_ckdtree_20k_subsample": float(np.max(np.abs(got - want) / np.maximum(want, 1e-12))),
            "workload": "gsr_knn_dist2 (simple_knn distCUDA2) on the headline scene's 1.47 M means, incl. workspace "
                        "allocation; exact 3-NN, checked against scipy.spatial.cKDTree on a 20 k subsample"}
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
                "frac_of_8TBs": rec["roofline"]["frac"], "num_visible": rec["config"]["num_visible"],
                "num_rendered": rec["config"]["num_rendered"], "grads_finite": rec["config"]["grads_finite"],
                "variant": rec["config"]["parameter_packing"], "loss": rec["config"]["loss"]}
        out["train_step"]["workload"] = rec["config"]["workload"]
    except Exception as ex:  # noqa: BLE001
        out["train_step"] = {"error": f"{type(ex).__name__}: {ex}"}
    out["parity"] = parity_record(dev)
    return out


def parity_record(dev):
    """The checker's verdict on THIS build, in the line: the 8 scenes of BASELINE.json configs[3] at 200 k Gaussians
    through the drop-in rasterizer against the CPU oracle (oracle/gs_oracle.c; checker use, like cpu_baseline) -- the
    worst pixel INCLUDING the ones the oracle flags as borderline (an alpha >= 1/255 or T >= 1e-4 decision within an
    exp() ulp of its threshold), the worst off them, how many are flagged.  north_star's bar is 1e-4."""
    try:
        import numpy as np

        from gsworld_amd import scenes
        from tests import helpers as hp

        worst_all, worst_off, border, per = 0.0, 0.0, 0, {}
        for i, n in enumerate(scenes.SCENE_NAMES):
            raw, cam = scenes.tabletop_scene(n, n=200_000, seed=1 + i), scenes.sensor_camera(n)
            inp, st = hp.np_inputs(raw, cam), hp.oracle_settings(cam)
            bg = np.zeros(3, np.float32)
            o = hp.oracle_forward(inp, st, bg)
            g = hp.gpu_forward(inp, st, bg, device=str(dev))
            d = np.abs(g["color"] - o["color"]).max(0)
            b = o["borderline"] != 0
            per[n] = float(d.max())
            worst_all, worst_off, border = max(worst_all, float(d.max())), max(worst_off, float(d[~b].max())), border + int(b.sum())
        return {"worst_pixel_all_scenes": worst_all, "worst_pixel_off_borderline": worst_off,
                "borderline_pixels": border, "pixels": 8 * 640 * 480, "per_scene_worst": per,
                "against": "oracle/gs_oracle.c (CPU restatement; UNPINNED against the CUDA reference, DESIGN.md section 2)",
                "workload": "8 scenes of configs[3], 200 k Gaussians each, 640x480 sensor camera, default frames"}
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
