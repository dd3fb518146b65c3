"""Closed-loop renderers: FrameRenderer (persistent state, no host sync) and MultiCameraRenderer (all cameras of a
step concurrently) must reproduce the drop-in rasterizer bit for bit."""
import pytest
import torch

from gsworld_amd import debug as dbg, scenes
from gsworld_amd.camera import look_at_view

pytestmark = pytest.mark.gpu


def _two_cameras(dev):
    right = scenes.sensor_camera("xarm6_align").to(dev)
    wrist = look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, 640, 480).to(dev)
    return [right, wrist]


def test_multi_camera_matches_sequential_frames(cuda_device):
    from gsworld_amd.renderer import FrameRenderer, MultiCameraRenderer

    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=300_000, seed=2)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    cams = _two_cameras(dev)
    bg = torch.tensor([0.1, 0.0, 0.2], device=dev)
    want = []
    for cam in cams:
        r = FrameRenderer(dev)
        for _ in range(2):  # second frame runs on the no-sync capacity path
            color, radii, invd = r.render(cam, means, op, shs=shs, scales=sc, rotations=rot, bg=bg)
        assert not r.ensure_valid(lambda: None).overflow
        want.append((color.clone(), radii.clone(), invd.clone(), r.pack_rgb8(color).clone()))
    mc = MultiCameraRenderer(len(cams), dev)
    frames = [torch.empty((480, 640, 3), dtype=torch.uint8, device=dev) for _ in cams]
    for _ in range(3):
        outs = mc.render(cams, means, op, rgb8_out=frames, shs=shs, scales=sc, rotations=rot, bg=bg)
    stats = mc.ensure_valid(lambda: mc.render(cams, means, op, rgb8_out=frames, shs=shs, scales=sc, rotations=rot,
                                              bg=bg))
    torch.cuda.synchronize()
    assert all(not s.overflow and s.num_rendered > 0 for s in stats)
    assert stats[0].num_rendered != stats[1].num_rendered  # two genuinely different views
    for (color, radii, invd), frame, (wc, wr, wi, wf) in zip(outs, frames, want):
        assert torch.equal(color, wc) and torch.equal(radii, wr) and torch.equal(invd, wi) and torch.equal(frame, wf)


def test_multi_camera_recovers_from_capacity_overflow(cuda_device):
    """A lane whose binning capacity is too small flags overflow on device; ensure_valid grows it and re-renders."""
    from gsworld_amd.renderer import FrameRenderer, MultiCameraRenderer

    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=100_000, seed=4)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    cams = _two_cameras(dev)
    mc = MultiCameraRenderer(2, dev)
    kw = dict(shs=shs, scales=sc, rotations=rot)
    mc.render(cams, means, op, **kw)          # exact frame sizes the capacities
    mc.lanes[1].r_capacity = 1 << 10          # far too small for the next frame
    outs = mc.render(cams, means, op, **kw)
    assert mc.lanes[1].stats().overflow
    stats = mc.ensure_valid(lambda: mc.render(cams, means, op, **kw))
    assert not any(s.overflow for s in stats)
    ref = FrameRenderer(dev).render(cams[1], means, op, **kw)[0]
    assert torch.equal(outs[1][0], ref)  # renderer-owned output tensor now holds the re-rendered frame


def test_split_sh_storage_is_bit_identical_to_the_concatenated_call(cuda_device):
    """features_dc / features_rest read in place (GsrInputs.shs_rest) vs upstream's per-frame cat."""
    from gsworld_amd.renderer import FrameRenderer

    dev = cuda_device
    cam = scenes.sensor_camera("xarm6_align").to(dev)
    raw = scenes.tabletop_scene("xarm6_align", n=200_000, seed=7)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    dc, rest = raw.features_dc.to(dev).contiguous(), raw.features_rest.to(dev).contiguous()
    assert torch.equal(torch.cat((dc, rest), dim=1), shs)
    for deg in (3, 2, 0):
        a = FrameRenderer(dev).render(cam, means, op, shs=shs, scales=sc, rotations=rot, sh_degree=deg)
        b = FrameRenderer(dev).render(cam, means, op, shs=dc, shs_rest=rest, scales=sc, rotations=rot, sh_degree=deg)
        assert all(torch.equal(x, y) for x, y in zip(a, b)), f"degree {deg}"
    with pytest.raises(ValueError):
        FrameRenderer(dev).render(cam, means, op, shs=shs, shs_rest=rest, scales=sc, rotations=rot)


def test_frame_and_step_replay_from_a_hipgraph(cuda_device):
    """No-sync frames are hipGraph-capturable (bench.py and the closed-loop tool rely on it): a captured
    MultiCameraRenderer step -- fork onto the per-camera streams, two frames, join -- replays bit-identically, and
    follows new Gaussian positions written into the captured input buffer."""
    from gsworld_amd.renderer import MultiCameraRenderer

    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=150_000, seed=9)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    cams = _two_cameras(dev)
    mc = MultiCameraRenderer(2, dev)
    frames = [torch.empty((480, 640, 3), dtype=torch.uint8, device=dev) for _ in cams]
    xyz = means.clone()  # captured input buffer

    def step():
        return mc.render(cams, xyz, op, rgb8_out=frames, shs=shs, scales=sc, rotations=rot)

    step()
    step()
    assert not any(s.overflow for s in mc.ensure_valid(step))
    eager = [f.clone() for f in frames]
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        step()
    for f in frames:
        f.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(frames, eager))
    xyz += torch.tensor([0.02, -0.01, 0.0], device=dev)  # the scene moves; same graph
    g.replay()
    torch.cuda.synchronize()
    moved = [f.clone() for f in frames]
    assert not torch.equal(moved[0], eager[0])
    step()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(frames, moved)) and not any(s.overflow for s in
                                                                              mc.ensure_valid(step))


def test_uint8_frame_is_the_same_from_every_compositor_variant(cuda_device):
    """GsrOutputs.out_rgb8: written inside the default compositing kernel, by a conversion pass behind the A/B
    variants -- and always equal to GSWorld's own conversion of the float image (gs_world_wrapper.py:268-270)."""
    from gsworld_amd.renderer import FrameRenderer

    dev = cuda_device
    cam = scenes.sensor_camera("xarm6_align").to(dev)
    raw = scenes.tabletop_scene("xarm6_align", n=120_000, seed=11)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    shs = shs.clone()
    shs[::3, 0] += 3.0  # a third of the splats far brighter than 1: exercises the clamp at 255
    bg = torch.tensor([0.9, 1.0, 0.4], device=dev)
    frames = {}
    try:
        for variant in (4, 0, 3):
            dbg.set_render_variant(variant, 0)
            out = torch.zeros((480, 640, 3), dtype=torch.uint8, device=dev)
            color, _, _ = FrameRenderer(dev).render(cam, means, op, shs=shs, scales=sc, rotations=rot, bg=bg,
                                                    rgb8_out=out)
            want = (color.permute(1, 2, 0) * 255).clamp(0, 255).to(torch.uint8)
            assert torch.equal(out, want), f"variant {variant}"
            frames[variant] = out
    finally:
        dbg.set_render_variant(4, 0)
    assert torch.equal(frames[4], frames[0]) and torch.equal(frames[4], frames[3])
    assert int(frames[4].max()) == 255


def test_split_quadrants_leave_the_image_state_bit_identical(cuda_device):
    """With GsrSettings.render_split = 1 the compositor cuts the quadrants that were costliest in the PREVIOUS frame on the
    same renderer state into two 8x4 halves on two waves (and by default it deals the quadrants to workgroups in the order
    of that cost).  Frame after frame on one state (so the split list is the real one, not a fresh state's
    arbitrary one), colour / inverse depth / uint8 frame must equal the unsplit render bit for bit."""
    from gsworld_amd.renderer import FrameRenderer

    dev = cuda_device
    cam = scenes.sensor_camera("xarm6_align").to(dev)
    raw = scenes.tabletop_scene("xarm6_align", n=400_000, seed=3)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    bg = torch.tensor([0.1, 0.3, 0.2], device=dev)

    def frames(n, split):
        dbg.set_render_split(split)
        try:
            r = FrameRenderer(dev)
            out = []
            for _ in range(n):
                rgb8 = torch.zeros((480, 640, 3), dtype=torch.uint8, device=dev)
                color, _, invd = r.render(cam, means, op, shs=shs, scales=sc, rotations=rot, bg=bg, rgb8_out=rgb8)
                out.append((color.clone(), invd.clone(), rgb8))
            assert not r.stats().overflow
            return out
        finally:
            dbg.set_render_split(False)

    ref = frames(1, False)[0]
    for color, invd, rgb8 in frames(4, True):
        assert torch.equal(color.view(torch.int32), ref[0].view(torch.int32))
        assert torch.equal(invd.view(torch.int32), ref[1].view(torch.int32))
        assert torch.equal(rgb8, ref[2])


@pytest.mark.parametrize("view", ["dense", "sensor"])
def test_cooperative_quadrants_leave_the_frame_bit_identical(cuda_device, view):
    """Inference frames hand the quadrants that were costliest in the previous frame to cooperative workgroups (render.hip
    render_coop_quadrant: three waves cull the rounds of 64 candidates in turn, the fourth composites their survivor lists in
    round order).  Frame after frame on one state -- so the choice is the real one -- colour, inverse depth and the uint8
    frame equal those of a compositor with one wave per quadrant (render_split = 3) bit for bit, one frame per launch and
    four; from above the table the robot's base makes some quadrants cooperative."""
    from gsworld_amd.renderer import FrameRenderer, MultiCameraRenderer

    dev = cuda_device
    cam = (scenes.dense_view_camera("xarm6_align") if view == "dense" else scenes.sensor_camera("xarm6_align")).to(dev)
    raw = scenes.tabletop_scene("xarm6_align", n=700_000, seed=4)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    bg = torch.tensor([0.1, 0.3, 0.2], device=dev)
    kw = dict(shs=shs, scales=sc, rotations=rot, bg=bg)

    def frames(n, mode):
        dbg.set_render_split(mode)
        try:
            r = FrameRenderer(dev, forward_only=True, want_radii=False)
            out, used = [], []
            for _ in range(n):
                rgb8 = torch.zeros((480, 640, 3), dtype=torch.uint8, device=dev)
                color, _, invd = r.render(cam, means, op, rgb8_out=rgb8, **kw)
                r.ensure_valid(lambda: r.render(cam, means, op, rgb8_out=rgb8, **kw))
                used.append(dbg.sort_state(r.geom)["coop_quads"])
                out.append((color.clone(), invd.clone(), rgb8))
            assert not r.stats().overflow
            return out, used
        finally:
            dbg.set_render_split(0)

    (ref,), used_off = frames(1, 3)
    assert used_off == [0]
    got, used = frames(5, 0)
    for color, invd, rgb8 in got:
        assert torch.equal(color.view(torch.int32), ref[0].view(torch.int32))
        assert torch.equal(invd.view(torch.int32), ref[1].view(torch.int32))
        assert torch.equal(rgb8, ref[2])
    assert all(0 <= u <= 64 for u in used), used
    if view == "dense":
        assert used[-1] > 0, used  # (the first frames of a state see no costs yet)
    mc = MultiCameraRenderer(4, dev, batched=True, forward_only=True, want_radii=False)
    for _ in range(3):
        outs = mc.render([cam] * 4, means, op, **kw)
    mc.ensure_valid(lambda: None)
    used4 = [dbg.sort_state(lane.geom)["coop_quads"] for lane in mc.lanes]
    assert all(0 <= u <= 64 for u in used4) and (view != "dense" or all(u > 0 for u in used4)), used4
    assert all(torch.equal(o[0], ref[0]) and torch.equal(o[2], ref[1]) for o in outs)


def test_static_camera_keeps_its_splitters_whatever_the_scene_does(cuda_device):
    """Under a bit-identical view matrix the depth sort takes the splitters the state holds without sampling
    (depthsort.hip).  Splitters only balance the buckets, never decide the order: a scene that changes completely under
    the static camera -- other Gaussians, other depth range, all depths equal (one bucket larger than the LDS: the
    global-memory fallback) -- must still come out as a fresh renderer renders it, bit for bit, and the frames after
    it (the sort samples again once a bucket came out far above its share) as well."""
    from gsworld_amd.renderer import FrameRenderer

    dev = cuda_device
    cam = scenes.identity_camera(320, 240, 60.0).to(dev)
    n = 120_000

    def scene(seed, squash=None, shift=0.0):
        raw = scenes.random_scene_camera_frame(n, seed=seed, near_fraction=0.0)
        raw.scaling -= 1.5
        if squash is not None:
            raw.xyz[:, 2] = squash  # identity view: depth = z
        raw.xyz[:, 2] += shift
        return [t.to(dev) for t in raw.activated()]

    scenes_ = [scene(50), scene(51, shift=3.0), scene(52, squash=2.5), scene(53), scene(53)]
    r = FrameRenderer(dev)
    for k, (means, shs, op, sc, rot) in enumerate(scenes_):
        got = [t.clone() for t in r.render(cam, means, op, shs=shs, scales=sc, rotations=rot, exact=True)]
        want = FrameRenderer(dev).render(cam, means, op, shs=shs, scales=sc, rotations=rot, exact=True)
        for a, b, what in zip(got, want, ("color", "radii", "invdepth")):
            assert torch.equal(a, b), f"frame {k}: {what} differs from a fresh renderer's"
        assert int(got[1].count_nonzero()) > 10_000
    # a model of another size on the same state: the header survives in the buffer, every array behind it moves
    for n2 in (70_000, 150_000, 70_000):
        raw = scenes.random_scene_camera_frame(n2, seed=60 + n2 % 7, near_fraction=0.0)
        raw.scaling -= 1.5
        means, shs, op, sc, rot = [t.to(dev) for t in raw.activated()]
        for _ in range(2):
            got = [t.clone() for t in r.render(cam, means, op, shs=shs, scales=sc, rotations=rot, exact=True)]
        want = FrameRenderer(dev).render(cam, means, op, shs=shs, scales=sc, rotations=rot, exact=True)
        for a, b, what in zip(got, want, ("color", "radii", "invdepth")):
            assert torch.equal(a, b), f"model of {n2}: {what} differs from a fresh renderer's"


@pytest.mark.parametrize("case", ["tabletop_640x480", "odd_grid_70x50", "one_tile_16x16", "huge_splats_400x304",
                                  "tall_33x257", "chunk_placement", "raw_split_sh", "beyond_resident_796x648",
                                  "wide_4100x40", "needles_400x304"])
def test_forward_only_frames_are_bit_identical(cuda_device, case):
    """GsrSettings.forward_only (include/gsr.h): instances binned per super-tile of 2 x 1 tiles, the compositor applying the
    reference's per-tile rect test itself, nothing a backward reads written -- the colour image, inverse depth, uint8
    frame and radii must be the very bits of the default frame (which the other tests hold against the oracle), on
    even and odd tile grids, with splats that cover many super-tiles, on the exact AND the no-sync capacity path, and
    the super-tile lists must be well below the per-tile ones.  Grids the super-tile compositor does not take (more
    tiles than resident quadrant waves -- found by tools/fuzz_forward_only.py --, more than 255 tile columns) keep
    per-tile lists and must still be the same bits."""
    from gsworld_amd._lib import RAW_OPACITY, RAW_ROTATIONS, RAW_SCALES
    from gsworld_amd.renderer import FrameRenderer

    dev = cuda_device
    W, H = {"tabletop_640x480": (640, 480), "odd_grid_70x50": (70, 50), "one_tile_16x16": (16, 16),
            "huge_splats_400x304": (400, 304), "tall_33x257": (33, 257), "chunk_placement": (640, 480),
            "raw_split_sh": (640, 480), "beyond_resident_796x648": (796, 648), "wide_4100x40": (4100, 40),
            "needles_400x304": (400, 304)}[case]
    if case in ("tabletop_640x480", "chunk_placement", "raw_split_sh"):
        raw, cam = scenes.tabletop_scene("xarm6_align", n=300_000, seed=12), scenes.sensor_camera("xarm6_align")
    else:
        raw = scenes.random_scene_camera_frame(30_000, seed=13)
        cam = scenes.identity_camera(W, H, 70.0)
        if case == "huge_splats_400x304":
            raw.scaling += 1.8
            raw.opacity -= 2.0
        if case == "needles_400x304":
            # needle-shaped splats at every angle: conics up to the limit B^2 = 0.999 A C below which inference frames
            # shrink the tile rect to the ellipse alpha >= 1/255 -- where the compositor's float error on the power is
            # largest against the ellipse's extent (the margin of the shrunken rect follows the conditioning)
            raw.scaling[:, 0] += 2.5
            raw.scaling[:, 1] -= 2.0
            raw.scaling[:, 2] -= 2.0
            raw.opacity += 2.0
    cam = cam.to(dev)
    bg = torch.tensor([0.3, 0.1, 0.6], device=dev)
    if case == "raw_split_sh":
        r_ = raw.to(dev)
        args = (r_.xyz, r_.opacity)
        kw = dict(shs=r_.features_dc, shs_rest=r_.features_rest, scales=r_.scaling, rotations=r_.rotation,
                  param_space=RAW_OPACITY | RAW_SCALES | RAW_ROTATIONS, bg=bg)
    else:
        means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
        args, kw = (means, op), dict(shs=shs, scales=sc, rotations=rot, bg=bg)
    if case == "chunk_placement":
        dbg.set_binning_mode(4)
    try:
        full, fast, bare = FrameRenderer(dev), FrameRenderer(dev, forward_only=True), \
            FrameRenderer(dev, forward_only=True, want_radii=False)
        f8 = [torch.zeros((H, W, 3), dtype=torch.uint8, device=dev) for _ in range(3)]
        for it in range(3):  # frame 0 exact (sizes the capacity), frames 1, 2 on the no-sync path
            want = full.render(cam, *args, rgb8_out=f8[0], **kw)
            got = fast.render(cam, *args, rgb8_out=f8[1], **kw)
            got2 = bare.render(cam, *args, rgb8_out=f8[2], **kw)
            torch.cuda.synchronize()
            assert torch.equal(got[0], want[0]) and torch.equal(got[2], want[2]), f"{case}: frame {it} differs"
            assert torch.equal(got[1], want[1]), "radii"
            assert got2[1] is None and torch.equal(got2[0], want[0]) and torch.equal(got2[2], want[2])
            assert torch.equal(f8[1], f8[0]) and torch.equal(f8[2], f8[0])
        sf, ss = full.ensure_valid(lambda: None), fast.ensure_valid(lambda: None)
        # (inference frames list a Gaussian only in the tiles it can colour -- alpha >= 1/255 somewhere inside: fewer
        #  instances, and a Gaussian that colours nothing is not counted visible)
        assert not sf.overflow and not ss.overflow and 0 < ss.num_visible <= sf.num_visible
        assert ss.num_rendered <= sf.num_rendered
        if case == "tabletop_640x480":
            assert ss.num_rendered < 0.7 * sf.num_rendered
        assert int(f8[0].max()) > 60
    finally:
        dbg.set_binning_mode(1)


def test_forward_only_capacity_overflow_is_recovered(cuda_device):
    from gsworld_amd.renderer import FrameRenderer

    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=150_000, seed=14)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    cam = scenes.sensor_camera("xarm6_align").to(dev)
    kw = dict(shs=shs, scales=sc, rotations=rot)
    ref = FrameRenderer(dev).render(cam, means, op, **kw)[0].clone()
    r = FrameRenderer(dev, forward_only=True)
    r.render(cam, means, op, **kw)
    assert r.stats().overflow_frames == 0  # a fresh buffer's header is garbage: the count must not be
    r.r_capacity = 1 << 10
    for _ in range(3):
        bg_only = r.render(cam, means, op, **kw)[0]
    assert r.stats().overflow and r.stats().overflow_frames == 3  # counted on the device, frame by frame
    assert float(bg_only.abs().max()) == 0.0                      # an overflowed frame shows the background only
    s = r.ensure_valid(lambda: r.render(cam, means, op, **kw))
    assert not s.overflow and s.overflow_frames == 3              # ... and valid frames do not clear the count
    assert torch.equal(r.render(cam, means, op, **kw)[0], ref)
    # the default (training-capable) frames and the alternative binning paths count the same way
    from gsworld_amd import _lib

    # ... and every path leaves the count where a host that does not wait can read it (GsrOutputs.overflow_mirror: two pinned
    # words the frame's capacity check writes itself on the counting placements, an 8-byte copy on the A/B paths)
    for tune in ({}, {"binning_path": 2}, {"binning_path": 1}, {"binning_path": 3}, {"depth_sort": 1}):
        saved = dict(_lib.TUNING)
        _lib.TUNING.update(tune)
        try:
            d = FrameRenderer(dev, overflow_mirror=True)
            assert d._mirror_dev, "pinned host memory has a device-visible address on this platform"
            d.render(cam, means, op, **kw)
            torch.cuda.synchronize()
            assert d.overflows_seen() == 0, tune
            d.r_capacity = 1 << 10
            d.render(cam, means, op, **kw)
            d.render(cam, means, op, **kw)
            assert d.stats().overflow_frames == 2, tune
            assert d.overflows_seen() == 2, tune  # (stats() has synchronised: the mirror is what the device wrote)
            m = FrameRenderer(dev, forward_only=True, overflow_mirror=True)
            m.render(cam, means, op, **kw)
            m.r_capacity = 1 << 10
            m.render(cam, means, op, **kw)
            torch.cuda.synchronize()
            assert m.overflows_seen() == 1 and m.stats().overflow_frames == 1, tune
        finally:
            _lib.TUNING.update(saved)


def test_state_buffers_recycled_between_layouts(cuda_device):
    """The same byte buffers used first by a default (full-layout) frame, then by inference frames (lean layout) and
    back: what the first frame left in the header -- kept splitters, placement cuts, static-camera flags -- belongs to
    arrays that have moved.  The kept tables are tied to the layout they were written under (depthsort.hip `sig`), so
    such a hand-over costs one sampling frame; every frame must be the reference frame bit for bit either way."""
    from gsworld_amd.renderer import FrameRenderer

    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=250_000, seed=15)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    cam = scenes.sensor_camera("xarm6_align").to(dev)
    kw = dict(shs=shs, scales=sc, rotations=rot)
    full, lean = FrameRenderer(dev), FrameRenderer(dev, forward_only=True, want_radii=False)
    want = full.render(cam, means, op, **kw)[0].clone()
    for _ in range(3):
        full.render(cam, means, op, **kw)  # static camera: the header now says "take the splitters blind"
    for a, b in ((full, lean), (lean, full), (full, lean)):
        b.geom, b.binning, b.image = a.geom, a.binning, a.image  # hand the very same storage over
        b.r_capacity = 0
        for _ in range(3):
            got = b.render(cam, means, op, **kw)[0]
            assert torch.equal(got, want)
        assert not b.ensure_valid(lambda: None).overflow


def test_fixed_camera_takes_kept_splitters_blind_only_while_the_scene_stands_still(cuda_device):
    """Depth-sort splitters kept from frame to frame (depthsort.hip).  Under a fixed camera a scene that stands still
    earns the right to skip the sample check ("blind") after a few balanced frames; a scene that MOVES under the same
    camera must lose it at once and not get it back while it moves -- an arm swinging into a depth range that was empty a
    frame ago lands in one wide kept bucket, and a bucket beyond the LDS is sorted in global memory for a millisecond
    (found with the forward-kinematics rollout: a quarter of its frames did).  Every frame is the exact-mode frame bit
    for bit whatever the policy does."""
    from gsworld_amd.renderer import FrameRenderer

    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=400_000, seed=21)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    cam = scenes.sensor_camera("xarm6_align").to(dev)
    kw = dict(shs=shs, scales=sc, rotations=rot)
    arm = (raw.semantics.reshape(-1) > 0).to(dev)
    r = FrameRenderer(dev, forward_only=True, want_radii=False)
    for _ in range(10):  # (a recycled state may postpone the first check of its table by three frames: ss_vfail)
        r.render(cam, means, op, **kw)
    assert dbg.sort_state(r.geom)["blind"], "a static scene under a fixed camera should reuse its splitters unchecked"
    toward_camera = (cam.camera_center - means[arm].mean(0))
    toward_camera = toward_camera / toward_camera.norm()
    states = []
    for k in range(1, 7):
        moved = means.clone()
        moved[arm] += 0.12 * k * toward_camera  # the robot's clusters travel 12 cm in depth per frame
        got = r.render(cam, moved, op, **kw)[0].clone()
        states.append(dbg.sort_state(r.geom))
        want = FrameRenderer(dev).render(cam, moved, op, exact=True, **kw)[0]
        assert torch.equal(got, want), f"frame {k} of the moving scene"
    assert states[0]["blind"] and states[0]["bad"], states[0]      # the first moved frame could not know
    assert not any(s["blind"] for s in states[1:]), states         # ... the following ones sample
    for _ in range(10):                                            # the scene stops: trust comes back
        r.render(cam, moved, op, **kw)
    assert dbg.sort_state(r.geom)["blind"]


def test_slowly_moving_camera_takes_the_previous_frame_s_quantiles_unchecked(cuda_device):
    """A camera that moves a little per frame (a wrist camera riding on the arm: millimetres, a fraction of a degree) over
    the splitter table its previous frame left: once the frames before it have classified with the kept table and come out
    balanced, the frame skips the samples (depthsort.hip ss_prepare, ``near``); a jump to another view samples again.
    Splitters only decide the balance of the depth buckets: every frame is the exact-mode frame of a fresh renderer bit for
    bit, whichever route it took."""
    import math

    from gsworld_amd.camera import look_at_view
    from gsworld_amd.renderer import FrameRenderer

    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=400_000, seed=23)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    kw = dict(shs=shs, scales=sc, rotations=rot)

    def wrist(k):  # 2.5 mm and ~0.3 degrees per frame
        a = 0.025 * k
        return look_at_view([0.55 - 0.10 * math.sin(a), 0.35, 0.25 + 0.05 * math.sin(2.0 * a)], [0.35, 0.05, 0.05],
                            [0, 0, 1], 0.9715089, 0.7551448, 640, 480).to(dev)

    r = FrameRenderer(dev, forward_only=True, want_radii=False, min_capacity=1 << 25)
    r.render(wrist(0), means, op, **kw)
    r.geom[:256].zero_()  # (whatever header the allocator handed back with the buffer: a state no frame has sorted on)
    states = []
    for k in range(12):
        cam = wrist(k)
        got = r.render(cam, means, op, **kw)[0].clone()
        states.append(dbg.sort_state(r.geom))
        want = FrameRenderer(dev).render(cam, means, op, exact=True, **kw)[0]
        assert torch.equal(got, want), f"frame {k}: {states[-1]}"
    assert not states[0]["blind"] and not states[1]["blind"], states[:2]   # nothing to trust yet
    assert any(s["near"] for s in states), states                          # ... then the kept table is taken as it is
    assert all(s["blind"] for s in states if s["near"]) and not any(s["stride"] != 1 for s in states if s["near"]), states
    far = scenes.dense_view_camera("xarm6_align").to(dev)                  # a jump: samples again
    got = r.render(far, means, op, **kw)[0].clone()
    st = dbg.sort_state(r.geom)
    assert not st["blind"] and not st["near"], st
    r.ensure_valid(lambda: r.render(far, means, op, **kw))
    got = r.render(far, means, op, **kw)[0].clone()
    assert torch.equal(got, FrameRenderer(dev).render(far, means, op, exact=True, **kw)[0])


def test_resting_camera_halves_its_bucket_count_and_a_moving_one_takes_it_back(cuda_device):
    """Frames that sample cut the depth order into buckets of <= 512 records, frames that take the kept exact quantiles
    unchecked into buckets of <= 1024 (depthsort.hip ss_prepare): the first blind frame reads every second entry of the
    kept table (stride 2), the following ones a table of their own count; a camera that moves again draws a table of
    the larger count.  Every frame is the exact-mode frame of a fresh renderer bit for bit, state included."""
    from gsworld_amd.renderer import FrameRenderer

    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align")  # full size: 1 468 850 Gaussians, about 176 k of them visible
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    cam = scenes.sensor_camera("xarm6_align").to(dev)
    kw = dict(shs=shs, scales=sc, rotations=rot)
    want = FrameRenderer(dev)
    w_color, w_radii, w_invd = (t.clone() for t in want.render(cam, means, op, exact=True, **kw))
    V = want.stats().num_visible
    assert 131072 < V <= 262144, V  # 512 buckets of <= 512 records, 256 of <= 1024
    r = FrameRenderer(dev)  # (a default-mode renderer: the point list and the ranges are there to compare)
    seen = []
    for k in range(8):
        color, radii, invd = r.render(cam, means, op, **kw)
        st = dbg.sort_state(r.geom)
        seen.append((st["blind"], st["buckets"], st["stride"]))
        assert torch.equal(color, w_color) and torch.equal(invd, w_invd) and torch.equal(radii, w_radii), f"frame {k}"
        sa, sb = (dbg.state_view(raw.num, 640, 480, x.stats().num_rendered, x.stats().num_visible, x.geom, x.binning,
                                 x.image, r_capacity=x.r_capacity) for x in (want, r))
        for name in ("point_list", "ranges", "depth_order"):
            assert torch.equal(sa[name], sb[name]), f"frame {k}: {name}"
    assert seen[0] == (False, 512, 1), seen
    first_blind = next(k for k, s in enumerate(seen) if s[0])
    assert seen[first_blind] == (True, 256, 2), seen            # every second entry of the 512-quantile table
    assert all(s == (True, 256, 1) for s in seen[first_blind + 1:]) and first_blind + 1 < len(seen), seen
    moved = scenes.dense_view_camera("xarm6_align").to(dev)  # another view: most of the scene on screen
    r.render(moved, means, op, **kw)
    st = dbg.sort_state(r.geom)
    assert not st["blind"] and st["fresh"] and st["stride"] == 1 and st["buckets"] > 256, st
    r.ensure_valid(lambda: r.render(moved, means, op, **kw))  # (the new view holds more instances than the old capacity)
    got = r.render(moved, means, op, **kw)[0].clone()
    assert not r.stats().overflow
    assert torch.equal(got, FrameRenderer(dev).render(moved, means, op, exact=True, **kw)[0])


@pytest.mark.parametrize("case", ["ten_thousand_into_one_bucket", "forty_thousand_into_one_bucket", "five_thousand_equal_depths"])
def test_depth_bucket_beyond_the_lds_is_still_sorted_exactly(cuda_device, case):
    """A depth bucket of the sample sort that outgrows the LDS (kBucketCap = 2048 records): the kept splitters were
    taken unchecked under a fixed camera and the scene jumped.  The bucket's workgroup cuts it once more by
    sub-splitters drawn from its own keys and sorts the pieces in the LDS (by index, then by key); a piece that still
    does not fit -- thousands of EQUAL depths -- goes through the global-memory network.  Whatever route, the frame is the
    exact-mode frame of a fresh renderer bit for bit, and so is the next one (the splitters that frame left behind)."""
    from gsworld_amd.renderer import FrameRenderer

    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=400_000, seed=22)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    cam = scenes.sensor_camera("xarm6_align").to(dev)
    kw = dict(shs=shs, scales=sc, rotations=rot)
    arm = torch.nonzero((raw.semantics.reshape(-1) > 0).to(dev)).reshape(-1)
    # (binning capacity for whatever lands in front of the camera: this test is about the depth sort)
    r = FrameRenderer(dev, forward_only=True, want_radii=False, min_capacity=1 << 26)
    for _ in range(10):  # (a recycled state may postpone the first check of its table by three frames: ss_vfail)
        r.render(cam, means, op, **kw)
    assert dbg.sort_state(r.geom)["blind"]
    gen = torch.Generator().manual_seed(5)
    # a spot in the middle of what the camera sees, a tenth of the way towards it (depths nothing else has)
    seen = FrameRenderer(dev).render(cam, means, op, **kw)[1] > 0
    spot = means[seen].mean(0)
    spot = spot + 0.1 * (cam.camera_center - spot)
    moved = means.clone()
    if case == "ten_thousand_into_one_bucket":
        pick = arm[torch.randperm(arm.numel(), generator=gen)[:10_000].to(dev)]
        moved[pick] = spot + 0.002 * torch.randn(pick.numel(), 3, generator=gen).to(dev)
    elif case == "forty_thousand_into_one_bucket":  # (V stays below the next bucket count: the kept table is taken)
        pick = arm[torch.randperm(arm.numel(), generator=gen)[:40_000].to(dev)]
        moved[pick] = spot + 0.002 * torch.randn(pick.numel(), 3, generator=gen).to(dev)
    else:
        pick = arm[torch.randperm(arm.numel(), generator=gen)[:5_000].to(dev)]
        moved[pick] = spot  # one position: 5 000 identical depth keys, ordered by index alone
    for k in range(2):
        got = r.render(cam, moved, op, **kw)[0].clone()
        st = dbg.sort_state(r.geom)
        want = FrameRenderer(dev).render(cam, moved, op, exact=True, **kw)[0]
        assert not r.stats().overflow
        assert torch.equal(got, want), f"{case}: frame {k} after the jump"
        if k == 0:
            assert st["blind"] and st["bad"], (st, r.stats())
            assert r.stats().num_visible > 40_000 + (4_000 if case != "forty_thousand_into_one_bucket" else 30_000)


def test_bounded_capacity_renderer_never_reads_back(cuda_device):
    """FrameRenderer(bound_capacity=True): the instance list is sized by P x tiles (no frame can exceed it), so there is
    no exact-mode first frame and no overflow flag worth reading -- the drop-in render() uses it to return without a host
    synchronisation.  Frames are the default renderer's bits from the first call on, also after the model or the image
    size changes; a bound beyond the budget falls back to the read-back protocol."""
    from gsworld_amd import _C
    from gsworld_amd.renderer import FrameRenderer

    dev = cuda_device
    r = FrameRenderer(dev, forward_only=True, want_radii=False, bound_capacity=True)
    for n, (w, h) in ((60_000, (320, 240)), (60_000, (200, 120)), (25_000, (200, 120))):
        raw = scenes.random_scene_camera_frame(n, seed=n + w)
        means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
        cam = scenes.identity_camera(w, h, 60.0).to(dev)
        kw = dict(shs=shs, scales=sc, rotations=rot)
        want = FrameRenderer(dev).render(cam, means, op, exact=True, **kw)
        for _ in range(2):
            got = r.render(cam, means, op, **kw)
            assert r.bounded and r.r_capacity == n * ((w + 15) // 16) * ((h + 15) // 16)
            assert torch.equal(got[0], want[0]) and torch.equal(got[2], want[2])
        assert not r.stats().overflow and r.stats().overflow_frames == 0
    budget, _C.NOSYNC_LIST_BYTES = _C.NOSYNC_LIST_BYTES, 1 << 12
    try:
        r2 = FrameRenderer(dev, forward_only=True, want_radii=False, bound_capacity=True)
        got = r2.render(cam, means, op, **kw)
        assert not r2.bounded and torch.equal(got[0], want[0])
        assert not r2.ensure_valid(lambda: r2.render(cam, means, op, **kw)).overflow
    finally:
        _C.NOSYNC_LIST_BYTES = budget


@pytest.mark.gpu
def test_a_cooperative_quadrant_that_gives_up_is_reported_not_drawn(cuda_device, monkeypatch):
    """The replaying wave of a cooperative quadrant bounds its waits for the culling waves (render.hip kCoopSpinLimit); a
    hand-off that never came would leave a truncated quadrant.  That cannot happen by the counters' construction -- and if it
    ever does, the frame has to SAY so, the way a binning overflow does.  libgsr_hip_coopspin.so
    is the same library with the limit at ONE poll (csrc/Makefile): from above the table the second frame on a state has
    cooperative quadrants, their first hand-offs time out, and gsr_frame_stats reports the frame as truncated
    (GSR_E_TRUNCATED, GsrFrameStats.truncated / coop_timeouts); ensure_valid re-renders once and then raises."""
    import os

    from gsworld_amd import _C, _lib
    from gsworld_amd.renderer import FrameRenderer

    path = os.path.join(os.path.dirname(_lib.LIB_PATH), "libgsr_hip_coopspin.so")
    if not os.path.exists(path):
        pytest.skip("libgsr_hip_coopspin.so not built (make -C gsworld_amd/csrc)")
    dev = cuda_device
    cam = scenes.dense_view_camera("xarm6_align").to(dev)
    raw = scenes.tabletop_scene("xarm6_align", n=700_000, seed=4)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    kw = dict(shs=shs, scales=sc, rotations=rot, bg=torch.zeros(3, device=dev))
    # the committed library first: same frames, nothing truncated
    r = FrameRenderer(dev, forward_only=True, want_radii=False)
    for _ in range(4):
        r.render(cam, means, op, **kw)
        r.ensure_valid(lambda: r.render(cam, means, op, **kw))
    assert dbg.sort_state(r.geom)["coop_quads"] > 0
    s = r.stats()
    assert not s.truncated and s.coop_timeouts == 0
    monkeypatch.setattr(_C, "_ext", None)        # (the compiled binding is linked against the committed library)
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", path)
    assert b"gfx950" in _lib.lib().gsr_version()
    r2 = FrameRenderer(dev, forward_only=True, want_radii=False)
    seen = []
    for _ in range(4):
        r2.render(cam, means, op, **kw)
        s2 = r2.stats()
        seen.append((s2.truncated, s2.coop_timeouts, dbg.sort_state(r2.geom)["coop_quads"]))
    # (a state without cooperative quadrants has nothing to time out; the first frame of THIS renderer may well have some --
    #  the allocator hands it the image state of the renderer above, last frame's costs included)
    assert all(q > 0 for t, _, q in seen if t), seen
    assert seen[-1][2] > 0 and seen[-1][0] is True and seen[-1][1] >= seen[-1][2] > 0, seen
    with pytest.raises(RuntimeError, match="timed out"):
        r2.ensure_valid(lambda: r2.render(cam, means, op, **kw))
