"""gsr_forward_batch (include/gsr.h): the frames of one step -- GSWorld's double loop over cameras and environments,
gs_world_wrapper.py:238-267 -- through ONE set of launches whose grids span the frames.  Every frame must come out as a
gsr_forward call with the same arguments leaves it: images, uint8 frames, radii AND the opaque state, bit for bit."""
import pytest
import torch

from gsworld_amd import layout as gl, scenes
from gsworld_amd.camera import look_at_view

pytestmark = pytest.mark.gpu


def _cams(dev, n, W=640, H=480):
    base = [scenes.sensor_camera("xarm6_align", W, H), scenes.dense_view_camera("xarm6_align", W, H),
            look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, W, H),
            look_at_view([0.4, 0.0, 0.3], [2.0, 1.5, 0.2], [0, 0, 1], 0.9715089, 0.7551448, W, H)]
    out = []
    for k in range(n):
        if k < len(base):
            out.append(base[k])
        else:  # a ring of wrist-like views around the table
            import math

            a = 2.0 * math.pi * k / n
            out.append(look_at_view([0.35 + 0.5 * math.cos(a), 0.05 + 0.5 * math.sin(a), 0.3 + 0.02 * k],
                                    [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, W, H))
    return [c.to(dev) for c in out]


@pytest.mark.parametrize("B,forward_only,with_layout", [(2, False, False), (3, True, False), (4, True, True),
                                                         (8, True, True), (11, True, True)])
def test_batched_frames_equal_frames_rendered_one_by_one(cuda_device, B, forward_only, with_layout):
    """B cameras through MultiCameraRenderer(batched=True) -- one gsr_forward_batch call per step, sets of up to 8 frames
    per launch -- against the same cameras through one FrameRenderer each, one gsr_forward call per frame."""
    from gsworld_amd.renderer import FrameRenderer, MultiCameraRenderer

    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=250_000, seed=31)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    kw = dict(shs=shs, scales=sc, rotations=rot)
    if with_layout:
        L = gl.SceneLayout.build(means, sc, rot, shs=shs, opacities=op)
        a = L.arrays
        means, op = a["means3D"], a["opacities"]
        kw = dict(shs=a["shs"], scales=a["scales"], rotations=a["rotations"], layout=L.layout)
    bg = torch.tensor([0.2, 0.05, 0.4], device=dev)
    cams = _cams(dev, B)
    rkw = dict(forward_only=forward_only, want_radii=True)
    singles = [FrameRenderer(dev, **rkw) for _ in cams]
    want = []
    for r, cam in zip(singles, cams):
        f8 = torch.zeros((480, 640, 3), dtype=torch.uint8, device=dev)
        for _ in range(3):  # exact frame, then two on the no-sync capacity path (kept splitters, kept cuts)
            color, radii, invd = r.render(cam, means, op, bg=bg, rgb8_out=f8, **kw)
        assert not r.ensure_valid(lambda: None).overflow
        want.append((color.clone(), radii.clone(), invd.clone(), f8))
    mc = MultiCameraRenderer(B, dev, batched=True, **rkw)
    frames = [torch.zeros((480, 640, 3), dtype=torch.uint8, device=dev) for _ in cams]
    for _ in range(3):
        outs = mc.render(cams, means, op, rgb8_out=frames, bg=bg, **kw)
    stats = mc.ensure_valid(lambda: None)
    torch.cuda.synchronize()
    assert all(not s.overflow and s.num_rendered > 0 for s in stats)
    for k, ((color, radii, invd), frame, (wc, wr, wi, wf)) in enumerate(zip(outs, frames, want)):
        assert torch.equal(color, wc), f"colour of frame {k}"
        assert torch.equal(invd, wi), f"inverse depth of frame {k}"
        assert torch.equal(radii, wr), f"radii of frame {k}"
        assert torch.equal(frame, wf), f"uint8 frame {k}"
        s1 = singles[k].stats()
        assert (stats[k].num_visible, stats[k].num_rendered) == (s1.num_visible, s1.num_rendered)
    # the opaque state a default frame leaves for a backward: point list, ranges, per-pixel state -- through the views
    if not forward_only:
        from gsworld_amd import debug as dbg

        for k in range(B):
            st = stats[k]
            va, vb = (dbg.state_view(raw.num, 640, 480, st.num_rendered, st.num_visible, r.geom, r.binning, r.image,
                                     r_capacity=r.r_capacity) for r in (singles[k], mc.lanes[k]))
            assert singles[k].r_capacity == mc.lanes[k].r_capacity
            for name in ("point_list", "ranges", "final_T", "n_contrib", "depth_order", "tiles_touched"):
                assert torch.equal(va[name], vb[name]), f"state array {name} of frame {k}"
            vis = want[k][1] > 0  # (records of culled Gaussians are never written: whatever the buffer held)
            assert torch.equal(va["splat"][vis], vb["splat"][vis]) and torch.equal(va["cov3D"][vis], vb["cov3D"][vis])


def test_batch_and_stream_paths_agree_over_a_moving_rollout(cuda_device):
    """The closed loop (two cameras, moving parts, wrist camera that moves every step, hipGraph replay) with its frames
    batched per launch against the same loop with a stream per frame: every step's uint8 frames equal."""
    from gsworld_amd import closed_loop as cl

    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=200_000, seed=5)
    cams = {"right_cam": scenes.sensor_camera("xarm6_align"),
            "wrist_cam": look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, 640, 480)}
    rollout = cl.xarm6_rollout()
    parts, actors = cl.xarm6_rollout_parts(rollout)
    poses = list(cl.rollout_poses(rollout, len(actors), steps=25, seed=3))
    loops = [cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev, batched=b, bound_capacity=False)
             for b in (True, False)]
    assert loops[0].multi.batched and not loops[1].multi.batched
    for lp in loops:
        lp.reset(*poses[0])
        lp.capture()
    for k, (M, s) in enumerate(poses[1:]):
        import math

        w = look_at_view([0.55 - 0.1 * math.sin(0.3 * k), 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089,
                         0.7551448, 640, 480)
        got = [{n: f.clone() for n, f in lp.step(M, s, cameras={"wrist_cam": w}, ensure=True).items()} for lp in loops]
        for n in got[0]:
            assert torch.equal(got[0][n], got[1][n]), f"step {k}, camera {n}"
    assert all(lp.overflow_frames() == 0 for lp in loops)


def test_environments_batch_into_one_launch_set(cuda_device):
    """num_envs x cameras = 6 frames of one step (per-environment pose tables inside preprocess) in one batched call,
    against the stream-per-frame path."""
    from gsworld_amd import closed_loop as cl

    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=120_000, seed=8)
    cams = {"right_cam": scenes.sensor_camera("xarm6_align"),
            "wrist_cam": look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, 640, 480)}
    rollout = cl.xarm6_rollout()
    parts, actors = cl.xarm6_rollout_parts(rollout)
    E = 3
    poses = list(cl.rollout_poses(rollout, len(actors), steps=6, seed=1, num_envs=E))
    loops = [cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev, num_envs=E, batched=b)
             for b in (True, False)]
    for lp in loops:
        lp.reset(*poses[0])
    for k, (M, s) in enumerate(poses[1:]):
        got = [{n: f.clone() for n, f in lp.step(M, s, ensure=True).items()} for lp in loops]
        for n in got[0]:
            assert got[0][n].shape[0] == E and torch.equal(got[0][n], got[1][n]), f"step {k}, camera {n}"
            assert not torch.equal(got[0][n][0], got[0][n][1])  # the environments really differ


def test_frames_that_cannot_share_launches_still_render(cuda_device):
    """A batch call with frames of two image sizes, an exact-mode frame (capacity 0) and an empty model in the middle:
    runs of compatible frames share launches, the rest go one by one -- every frame equal to its own gsr_forward call."""
    from gsworld_amd import _C
    from gsworld_amd.renderer import FrameRenderer

    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=80_000, seed=12)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    views = [scenes.sensor_camera("xarm6_align", 640, 480), scenes.dense_view_camera("xarm6_align", 640, 480),
             scenes.sensor_camera("xarm6_align", 320, 240), scenes.dense_view_camera("xarm6_align", 320, 240),
             scenes.sensor_camera("xarm6_align", 640, 480)]
    views = [v.to(dev) for v in views]
    want, lanes = [], []
    for v in views:
        r = FrameRenderer(dev, forward_only=True, want_radii=False)
        for _ in range(2):
            c, _, d = r.render(v, means, op, shs=shs, scales=sc, rotations=rot)
        want.append((c.clone(), d.clone()))
        lanes.append(FrameRenderer(dev, forward_only=True, want_radii=False))
    # size the batch lanes (exact frame each), then one batched call over all five; lane 4 is forced back to exact mode
    for r, v in zip(lanes, views):
        r.render(v, means, op, shs=shs, scales=sc, rotations=rot)
    lanes[4].r_capacity = 0
    calls = []
    for r, v in zip(lanes, views):
        call, c, _, d = r._prepare(v, means, op, shs=shs, scales=sc, rotations=rot)
        calls.append((call, c, d))
    assert calls[4][0]["r_capacity"] == 0 and all(c[0]["r_capacity"] > 0 for c in calls[:4])
    _C.forward_batch_raw([c[0] for c in calls], device=dev)
    torch.cuda.synchronize()
    for k, ((_, c, d), (wc, wd)) in enumerate(zip(calls, want)):
        assert torch.equal(c, wc) and torch.equal(d, wd), f"frame {k}"


def test_ctypes_and_compiled_bindings_issue_the_same_batch(cuda_device):
    """gsworld_amd._C.forward_batch_raw through the compiled binding and through ctypes (the same C ABI)."""
    import importlib
    import os

    from gsworld_amd import _C
    from gsworld_amd.renderer import FrameRenderer

    if _C._ext is None or not hasattr(_C._ext, "forward_batch"):
        pytest.skip("compiled binding not built")
    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=60_000, seed=14)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    views = _cams(dev, 3)
    results = []
    for use_ctypes in (False, True):
        ext = _C._ext
        if use_ctypes:
            _C._ext = None
        try:
            lanes = [FrameRenderer(dev, forward_only=True, want_radii=False) for _ in views]
            for r, v in zip(lanes, views):
                r.render(v, means, op, shs=shs, scales=sc, rotations=rot)
            calls = [r._prepare(v, means, op, shs=shs, scales=sc, rotations=rot) for r, v in zip(lanes, views)]
            _C.forward_batch_raw([c[0] for c in calls], device=dev)
            torch.cuda.synchronize()
            results.append([(c[1].clone(), c[3].clone()) for c in calls])
        finally:
            _C._ext = ext
    for (c0, d0), (c1, d1) in zip(*results):
        assert torch.equal(c0, c1) and torch.equal(d0, d1)
    del importlib, os


def test_the_timed_arrangement_meets_the_oracle_after_many_replays(cuda_device):
    """What bench.py's `value` times, checked against the ORACLE directly: BASELINE configs[1] at
    full size in its load-time layout, inference frames, EIGHT frames per gsr_forward_batch call, three such steps in flight
    on three streams, each step a replayed hipGraph -- and only after 33 replays, when every route a resting camera earns
    is engaged (kept splitters taken blind, buckets of 1024 records, the previous frame's placement cuts, cooperative
    quadrants dealt by the last frame's costs).  Every one of the 24 frames: colour and inverse depth within 1e-4 of the
    oracle's on every pixel, and the uint8 frame equal to GSWorld's conversion of the colour image."""
    import numpy as np

    from gsworld_amd import debug as dbg
    from gsworld_amd.layout import SceneLayout
    from gsworld_amd.renderer import MultiCameraRenderer
    from tests import helpers as hp

    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align")
    cam_cpu = scenes.sensor_camera("xarm6_align")
    inp, st = hp.np_inputs(raw, cam_cpu), hp.oracle_settings(cam_cpu)
    o = hp.oracle_forward(inp, st, np.zeros(3, np.float32))
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    L = SceneLayout.build(means, sc, rot, shs=shs, opacities=op)
    a = L.arrays
    cam = cam_cpu.to(dev)
    B, G = 8, 3
    mcs = [MultiCameraRenderer(B, dev, batched=True, forward_only=True, want_radii=False) for _ in range(G)]
    rgb8 = [[torch.zeros((480, 640, 3), dtype=torch.uint8, device=dev) for _ in range(B)] for _ in range(G)]
    outs = [None] * G

    def step(g):
        outs[g] = mcs[g].render([cam] * B, a["means3D"], a["opacities"], rgb8_out=rgb8[g], shs=a["shs"], scales=a["scales"],
                                rotations=a["rotations"], bg=torch.zeros(3, device=dev), layout=L.layout)

    for g in range(G):
        for _ in range(2):
            step(g)
            mcs[g].ensure_valid(lambda g=g: step(g))
    streams = [torch.cuda.Stream(dev) for _ in range(G)]
    graphs = []
    for g in range(G):
        streams[g].wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(streams[g]):
            step(g)
        torch.cuda.current_stream().wait_stream(streams[g])
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=streams[g]):
            step(g)
        graphs.append(gr)
    for k in range(33):
        with torch.cuda.stream(streams[k % G]):
            graphs[k % G].replay()
    torch.cuda.synchronize()
    border = o["borderline"] > 0
    worst = 0.0
    for g in range(G):
        assert not any(s.overflow or s.truncated for s in mcs[g].ensure_valid(lambda: None))
        assert dbg.sort_state(mcs[g].lanes[0].geom)["blind"], "a resting camera takes its kept splitters unchecked by now"
        for b in range(B):
            color, _, invd = outs[g][b]
            dc = np.abs(color.cpu().numpy() - o["color"])
            dd = np.abs(invd.cpu().numpy() - o["invdepth"])
            worst = max(worst, float(dc.max()))
            assert float(dc.max()) <= 1e-4, f"stream {g} frame {b}: colour off by {float(dc.max()):.3e}"
            assert float(dd.max()) <= 1e-4 * max(1.0, float(np.abs(o["invdepth"]).max())), f"stream {g} frame {b}: inverse depth"
            assert float(dc[:, ~border].max()) <= 1e-5
            want8 = (color.permute(1, 2, 0) * 255).clamp(0, 255).to(torch.uint8)
            assert torch.equal(rgb8[g][b], want8), f"stream {g} frame {b}: uint8 frame"
    print(f"timed arrangement vs oracle: worst pixel of 24 frames {worst:.3e}")


def test_cameras_of_two_image_sizes_overlap_as_two_sets_and_stay_bit_identical(cuda_device):
    """A rig with a wrist camera of another resolution than the sensor cameras: the batched
    MultiCameraRenderer sends the frames of each image size through gsr_forward_batch as one set, each set on a stream of
    its own, forked from and joined to the caller's stream -- also under hipGraph capture.  Every frame equals the frame of
    a FrameRenderer that renders it alone."""
    from gsworld_amd.renderer import FrameRenderer, MultiCameraRenderer

    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=150_000, seed=12)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    big = _cams(dev, 3)
    small = _cams(dev, 2, W=320, H=240)
    cams = [big[0], small[0], big[1], small[1], big[2]]
    kw = dict(shs=shs, scales=sc, rotations=rot, bg=torch.tensor([0.2, 0.1, 0.0], device=dev))
    want = []
    for cam in cams:
        r = FrameRenderer(dev, forward_only=True, want_radii=False)
        for _ in range(2):
            color, _, invd = r.render(cam, means, op, **kw)
        assert not r.ensure_valid(lambda: None).overflow
        want.append((color.clone(), invd.clone()))
    mc = MultiCameraRenderer(len(cams), dev, batched=True, forward_only=True, want_radii=False)
    for _ in range(3):
        outs = mc.render(cams, means, op, **kw)
        mc.ensure_valid(lambda: mc.render(cams, means, op, **kw))
    assert len(mc._set_streams) >= 2
    torch.cuda.synchronize()
    for (color, _, invd), (wc, wi) in zip(outs, want):
        assert torch.equal(color, wc) and torch.equal(invd, wi)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        mc.render(cams, means, op, **kw)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        outs = mc.render(cams, means, op, **kw)
    for color, _, invd in outs:
        color.zero_()
    g.replay()
    torch.cuda.synchronize()
    for (color, _, invd), (wc, wi) in zip(outs, want):
        assert torch.equal(color, wc) and torch.equal(invd, wi)
