"""BASELINE.json configs[2] -- the closed-loop rollout -- on the GPU: the fused per-step glue
(gsworld_amd.closed_loop.ClosedLoopRenderer: device-side pose table, one fused transform pass, all frames of a step in
flight, optional hipGraph replay) must reproduce the frames of the wrapper's own glue restated op for op in torch
(oracle/wrapper_glue_ref.py: deep copies, 18 isin() masks + transform_gaussians, masked write-backs, upstream render()
activations) on the same seeded pose sequence.  Harness shape: 1 reset + ep_len steps x 2 cameras, seeded random
actions (/root/reference/examples/maniskill/gsworld_rand_action_tabletop.py:99-133; gs_world_wrapper.py:110-162,232-275).

Bar: uint8 frames within 1 LSB (the two glues activate scales / rotations with different but <= 1 ulp exp / normalize,
and the uint8 cast truncates), and only on a small fraction of the pixels."""
import math
import types

import pytest
import torch

from gsworld_amd import closed_loop as cl
from gsworld_amd import debug as dbg
from gsworld_amd import scenes
from gsworld_amd.camera import look_at_view
from oracle import wrapper_glue_ref as ref

pytestmark = pytest.mark.gpu


def _setup(dev, n, num_envs=1, seed=1, fuse_transform=True, rollout=None, keep_float=False):
    raw = scenes.tabletop_scene("xarm6_align", n=n, seed=seed)
    cams = {"right_cam": scenes.sensor_camera("xarm6_align"),
            "wrist_cam": look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, 640, 480)}
    parts, actors = cl.xarm6_parts() if rollout is None else cl.xarm6_rollout_parts(rollout)
    loop = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, num_envs=num_envs, device=dev,
                                 fuse_transform=fuse_transform, keep_float=keep_float)
    rawd = raw.to(dev)
    model = types.SimpleNamespace(_xyz=rawd.xyz, _scaling=rawd.scaling, _rotation=rawd.rotation,
                                  _opacity=rawd.opacity.reshape(-1, 1, 1), _semantics=rawd.semantics,
                                  _features_dc=rawd.features_dc, _features_rest=rawd.features_rest)
    cams_d = {k: v.to(dev) for k, v in cams.items()}
    return raw, cams_d, parts, actors, loop, model


def _rasterize(view, means3D, shs, opacities, scales, rotations, bg):
    """Upstream's exact-mode call through the drop-in module (fresh state, D2H of num_rendered)."""
    from gsworld_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

    rs = GaussianRasterizationSettings(view.image_height, view.image_width, view.tanfovx, view.tanfovy, bg, 1.0,
                                       view.world_view_transform, view.full_proj_transform, 3, view.camera_center,
                                       False, False, False)
    color, _, _ = GaussianRasterizer(rs)(means3D=means3D, means2D=torch.zeros_like(means3D), shs=shs,
                                         opacities=opacities, scales=scales, rotations=rotations)
    return color


def _compare(got: dict, want: dict, what):
    for name in want:
        a, b = got[name].to(torch.int16), want[name].to(torch.int16)
        assert a.shape == b.shape, (what, name, a.shape, b.shape)
        d = (a - b).abs()
        assert int(d.max()) <= 1, f"{what} {name}: uint8 frames differ by {int(d.max())} LSB"
        assert float((d > 0).float().mean()) < 0.01, f"{what} {name}: {float((d > 0).float().mean()):.4f} of the bytes differ"
        assert int(b.max()) > 100, "frame is not trivially dark"


def test_fused_glue_reproduces_the_wrapper_glue_over_a_rollout(cuda_device):
    dev = cuda_device
    raw, cams, parts, actors, loop, model = _setup(dev, 200_000)
    sim2gs = torch.tensor(scenes.SIM2GS_XARM_TRANS)
    poses = list(cl.random_walk_poses(sim2gs, len(parts), len(actors), steps=11, seed=0))
    loop.reset(*poses[0])
    _compare({k: v.clone() for k, v in loop.frames.items()},
             ref.render_step(model, parts, cams, poses[0][0], poses[0][1], _rasterize, actors), "reset")
    eager = []
    for M, s in poses[1:]:
        frames = loop.step(M, s)
        eager.append({k: v.clone() for k, v in frames.items()})
        _compare(eager[-1], ref.render_step(model, parts, cams, M, s, _rasterize, actors), "step")
    assert not any(st.overflow for st in loop.ensure_valid())
    # the moved parts really move the image: consecutive frames differ
    assert not torch.equal(eager[0]["right_cam"], eager[-1]["right_cam"])
    # the same rollout replayed from ONE hipGraph per step gives the same bytes
    loop.capture()
    for (M, s), want in zip(poses[1:], eager):
        frames = loop.step(M, s)
        torch.cuda.synchronize()
        for name in want:
            assert torch.equal(frames[name], want[name]), f"graph replay differs from eager ({name})"
    assert not any(st.overflow for st in loop.ensure_valid())


@pytest.mark.parametrize("num_envs", [1, 2])
def test_transform_inside_preprocess_is_bit_identical_to_the_transform_pass(cuda_device, num_envs):
    """the default closed loop applies the step's rigid transforms inside every frame's preprocess
    (GsrInputs.part_*: no transformed copy of the model, one pose table per environment) -- the frames must be the very
    bytes of the two-pass path (gsr_transform_gaussians_batch, then frames over its outputs): same arithmetic."""
    dev = cuda_device
    _, _, parts, actors, fused, _ = _setup(dev, 150_000, num_envs=num_envs, seed=2, fuse_transform=True)
    _, _, _, _, twopass, _ = _setup(dev, 150_000, num_envs=num_envs, seed=2, fuse_transform=False)
    sim2gs = torch.tensor(scenes.SIM2GS_XARM_TRANS)
    poses = list(cl.random_walk_poses(sim2gs, len(parts), len(actors), steps=5, seed=9, num_envs=num_envs))
    fused.reset(*poses[0])
    twopass.reset(*poses[0])
    for M, s in poses:
        a, b = fused.step(M, s), twopass.step(M, s)
        torch.cuda.synchronize()
        for name in a:
            assert torch.equal(a[name], b[name]), f"{name}: in-preprocess transform differs from the transform pass"
    assert int(a["right_cam"].max()) > 100


def test_env_batch_matches_the_wrapper_glue(cuda_device):
    """num_envs = 3: the wrapper's `for i in range(self.num_envs)` loop (gs_world_wrapper.py:241-242) with (E,n,.) moved
    parts, against slices of the batched fused transform rendered from E x C lanes."""
    dev = cuda_device
    E = 3
    raw, cams, parts, actors, loop, model = _setup(dev, 60_000, num_envs=E, seed=3)
    sim2gs = torch.tensor(scenes.SIM2GS_XARM_TRANS)
    poses = list(cl.random_walk_poses(sim2gs, len(parts), len(actors), steps=3, seed=5, num_envs=E))
    loop.reset(*poses[0])
    for M, s in poses[1:]:
        frames = loop.step(M, s)
        assert frames["right_cam"].shape == (E, 480, 640, 3)
        _compare({k: v.clone() for k, v in frames.items()}, ref.render_step(model, parts, cams, M, s, _rasterize, actors),
                 "env batch")
    # environments really differ
    assert not torch.equal(loop.frames["right_cam"][0], loop.frames["right_cam"][1])


def test_full_size_rollout_200_steps(cuda_device):
    """configs[2] at size: 1,468,850 Gaussians, 1 reset + 200 steps x 2 cameras = 402 frames, replayed from the captured
    step graph.  The robot links follow forward kinematics of the reference's xarm6 URDF along a seeded random-action
    rollout (tests/golden/xarm6_rollout.npz; ``link6`` carries two labels, as ``xarm_gs_semantics`` has it), through
    the wrapper's own pose arithmetic (closed_loop.part_poses_from_sim).  Properties: no capacity overflow anywhere,
    frames keep changing, and steps 100 and 200 equal the wrapper glue on their poses (<= 1 LSB) -- i.e. nothing
    drifted over the rollout."""
    dev = cuda_device
    rollout = cl.xarm6_rollout()
    raw, cams, parts, actors, loop, model = _setup(dev, scenes.XARM6_ALIGN_NUM_GAUSSIANS, rollout=rollout)
    poses = list(cl.rollout_poses(rollout, len(actors), steps=201, seed=0))
    loop.reset(*poses[0])
    loop.capture()
    sums = []
    for i, (M, s) in enumerate(poses[1:]):
        frames = loop.step(M, s)
        if i % 20 == 0:
            sums.append((int(frames["wrist_cam"].sum().item()), int(frames["right_cam"].sum().item())))
        if i == 99:
            mid = {k: v.clone() for k, v in loop.frames.items()}
    last = {k: v.clone() for k, v in loop.frames.items()}
    stats = loop.ensure_valid()
    assert not any(st.overflow for st in stats), "a lane overflowed its binning capacity during the rollout"
    assert loop.overflow_frames() == 0, "some frame of the rollout exceeded its lane's binning capacity"
    assert len(set(sums)) == len(sums), f"frames stopped changing: {sums}"
    M, s = poses[100]
    _compare(mid, ref.render_step(model, parts, cams, M, s, _rasterize, actors), "step 100")
    M, s = poses[-1]
    _compare(last, ref.render_step(model, parts, cams, M, s, _rasterize, actors), "step 200")


def test_moving_wrist_camera_follows_through_eager_steps_and_graph_replay(cuda_device):
    """The wrapper recomputes its cameras on every render (gs_world_wrapper.py:238), so a wrist camera moves with the
    arm.  ClosedLoopRenderer.set_cameras copies the new matrices into the tensors the step reads -- also the CAPTURED
    step: every frame of a rollout with a moving wrist camera must equal the wrapper glue's for that step's camera."""
    dev = cuda_device
    raw, cams, parts, actors, loop, model = _setup(dev, 120_000, seed=4)
    sim2gs = torch.tensor(scenes.SIM2GS_XARM_TRANS)
    poses = list(cl.random_walk_poses(sim2gs, len(parts), len(actors), steps=7, seed=3))

    def wrist(k):
        return look_at_view([0.55 - 0.02 * k, 0.35, 0.25 + 0.01 * k], [0.35, 0.05 + 0.01 * k, 0.05], [0, 0, 1],
                            0.9715089, 0.7551448, 640, 480)

    loop.reset(*poses[0])
    loop.capture()
    loop.eager_when_ahead = False  # (this test is about the REPLAYED step; step(ensure=False) would issue eager launches)
    seen = []
    for k, (M, s) in enumerate(poses[1:]):
        w = wrist(k)
        frames = loop.step(M, s, cameras={"wrist_cam": w})
        torch.cuda.synchronize()
        want_cams = dict(cams, wrist_cam=w.to(dev))
        _compare({n: v.clone() for n, v in frames.items()},
                 ref.render_step(model, parts, want_cams, M, s, _rasterize, actors), f"step {k}")
        seen.append(frames["wrist_cam"].clone())
    assert not torch.equal(seen[0], seen[-1])
    assert not any(st.overflow for st in loop.ensure_valid())
    with pytest.raises(ValueError):
        loop.set_cameras({"wrist_cam": look_at_view([0.5, 0.3, 0.2], [0.3, 0.0, 0.0], [0, 0, 1], 0.9, 0.7, 320, 240)})


@pytest.mark.parametrize("E", [1, 3])
def test_frames_match_what_the_reference_glue_itself_handed_to_render(cuda_device, E):
    """Pinned to the reference, not to a restatement: tests/golden/wrapper_glue.npz holds the ``gs4render`` tensors the
    reference's own ``_render_gsworld`` passed to ``render()`` (tools/make_golden.py runs gs_world_wrapper.py:110-162,
    232-325 unmodified on a fake simulator) together with the ``rot_mat`` / ``translation`` / ``scale`` it computed for
    every part and the camera's R, T and field of view.  ClosedLoopRenderer gets the base model and those poses; its
    frames must be the frames of the captured tensors rendered the upstream way (exact-mode rasterizer, torch
    activations) within 1 LSB."""
    import os

    import numpy as np

    from gsworld_amd import camera as gcam
    from tests.test_wrapper_glue_cpu import model_of, part_labels_of

    dev = cuda_device
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "wrapper_glue.npz"))
    model, labels = model_of(z), part_labels_of(z)
    names = z[f"E{E}.part_names"].tolist()
    actors = tuple(n for n in names if f"E{E}.call.{n}.scale" in z.files)
    sem = model._semantics.reshape(-1)
    # a part with exactly num_envs Gaussians takes a degenerate branch in the reference (its rows are broadcast over
    # the part, tests/test_wrapper_glue_cpu.py): the product does not mirror that; those 3 Gaussians are made
    # transparent on both sides
    hide = sem == 15 if E == 3 else torch.zeros_like(sem, dtype=torch.bool)
    model._opacity = model._opacity.clone()
    model._opacity[hide] = -30.0
    cams = {}
    for c in ("right_cam", "wrist_cam"):
        size = z[f"E{E}.cam.{c}.size"]
        cams[c] = gcam.view_params(z[f"E{E}.cam.{c}.R"], z[f"E{E}.cam.{c}.T"], float(z[f"E{E}.cam.{c}.fov"][0]),
                                   float(z[f"E{E}.cam.{c}.fov"][1]), int(size[0]), int(size[1]))
    K = len(names)
    M, S = torch.eye(4).repeat(E, K, 1, 1), torch.ones(E, K)
    for k, n in enumerate(names):
        M[:, k, :3, :3] = torch.from_numpy(z[f"E{E}.call.{n}.rot_mat"])
        M[:, k, :3, 3] = torch.from_numpy(z[f"E{E}.call.{n}.translation"])
        if n in actors:
            S[:, k] = torch.from_numpy(z[f"E{E}.call.{n}.scale"])
    if E == 1:
        M, S = M[0], S[0]
    loop = cl.ClosedLoopRenderer(model, labels, cams, scaled_parts=actors, num_envs=E, device=dev)
    got = {k: v.clone() for k, v in loop.reset(M.contiguous(), S.contiguous()).items()}
    assert not any(st.overflow for st in loop.ensure_valid())
    cams_d = {k: v.to(dev) for k, v in cams.items()}
    bg = torch.zeros(3, device=dev)
    want = {c: [] for c in cams}
    for e in range(E):
        gs = types.SimpleNamespace(**{a: getattr(model, a).to(dev) for a in ("_features_dc", "_features_rest")})
        for a in ("_xyz", "_scaling", "_rotation", "_opacity"):
            setattr(gs, a, torch.from_numpy(z[f"E{E}.gs4render.env{e}.{a}"]).to(dev))
        gs._opacity[hide.to(dev)] = -30.0
        for c in cams:
            want[c].append(ref.render_model(gs, cams_d[c], _rasterize, bg))
    want = {c: torch.vstack(v) for c, v in want.items()}
    for c in cams:
        a, b = got[c].to(torch.int16), want[c].to(torch.int16)
        assert a.shape == b.shape == (E, 480, 640, 3)
        d = (a - b).abs()
        assert int(d.max()) <= 1, f"{c}: {int(d.max())} LSB"
        assert float((d > 0).float().mean()) < 0.01
    assert int(want["right_cam"].max()) > 100 and int((want["right_cam"].sum(-1) > 0).sum()) > 2000


def test_part_poses_from_sim_takes_device_tensors(cuda_device):
    """A GPU simulator hands over device tensors (ManiSkill link / actor poses live on the GPU): the pose arithmetic runs
    where they live and gives what the host path gives (the offsets and scales used to be created on
    the CPU)."""
    from gsworld_amd import closed_loop as cl

    dev = cuda_device
    ro = cl.xarm6_rollout()
    E, L = 2, ro["link_scan"].shape[0]
    now = torch.stack([ro["link_now"][3], ro["link_now"][40]])
    g = torch.Generator().manual_seed(0)
    actor_now = torch.eye(4).repeat(E, 2, 1, 1)
    actor_now[:, :, :3, 3] = torch.randn(E, 2, 3, generator=g) * 0.1
    sim2gs_obj = torch.eye(4).repeat(2, 1, 1) * 1.0
    sim2gs_obj[:, :3, :3] *= 0.5
    off, sc = torch.randn(2, 3, generator=g) * 0.01, torch.tensor([1.0, 0.7])
    want = cl.part_poses_from_sim(ro["sim2gs_arm"], now, ro["link_scan"], ro["link_offset"], actor_now, sim2gs_obj, off, sc)
    got = cl.part_poses_from_sim(ro["sim2gs_arm"].to(dev), now.to(dev), ro["link_scan"].to(dev), ro["link_offset"],
                                 actor_now.to(dev), sim2gs_obj.to(dev), off.to(dev), sc.to(dev))
    assert got[0].device.type == "cuda" and got[1].device.type == "cuda" and got[0].shape == (E, L + 2, 4, 4)
    assert float((got[0].cpu() - want[0]).abs().max()) < 1e-5 and float((got[1].cpu() - want[1]).abs().max()) < 1e-5


def test_overflow_under_graph_replay_is_never_returned(cuda_device):
    """Six lanes (3 environments x 2 cameras) replayed as one hipGraph, their instance lists sized with NO margin from
    the reset frame, while the arm swings through the wrist camera's view: with ``step(ensure=True)`` -- the closed
    loop's natural sync point, the policy reads the frames -- every frame handed out equals the frame of a loop whose
    lists cannot overflow (bound capacity), the overflows are noticed through the pinned mirrors and the graph is
    re-captured; in throughput mode (``ensure=False``) the same rollout notices them late but never stays wrong."""
    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=150_000, seed=9)
    rollout = cl.xarm6_rollout()
    parts, actors = cl.xarm6_rollout_parts(rollout)
    cams = {"right_cam": scenes.sensor_camera("xarm6_align"),
            "wrist_cam": look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, 640, 480)}
    E, steps = 3, 60
    poses = list(cl.rollout_poses(rollout, len(actors), steps=steps, seed=1, num_envs=E))
    safe = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, num_envs=E, device=dev, bound_capacity=True)
    tight = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, num_envs=E, device=dev, bound_capacity=False,
                                  growth=1.0, min_capacity=1 << 16)
    safe.reset(*poses[0])
    tight.reset(*poses[0])
    assert all(l.bounded for l in safe.multi.lanes) and not any(l.bounded for l in tight.multi.lanes)
    safe.capture()
    tight.capture()
    for k in range(1, steps):
        want = {n: f.clone() for n, f in safe.step(*poses[k]).items()}
        got = tight.step(*poses[k], ensure=True)
        torch.cuda.synchronize()
        for n in want:
            assert torch.equal(got[n], want[n]), f"step {k}, {n}: an overflowed frame was returned"
    assert tight.recovered_steps > 0, "the rollout never outgrew a list: the test proves nothing"
    assert tight.late_overflow_frames == 0 and tight.captured
    # throughput mode: no wait per step; whatever overflowed is noticed (late) and the loop ends valid
    loose = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, num_envs=E, device=dev, bound_capacity=False,
                                  growth=1.0, min_capacity=1 << 16)
    loose.reset(*poses[0])
    loose.capture()
    for k in range(1, steps):
        loose.step(*poses[k])
    torch.cuda.synchronize()
    got = loose.step(*poses[steps - 1], ensure=True)
    want = safe.step(*poses[steps - 1])
    torch.cuda.synchronize()
    assert all(torch.equal(got[n], want[n]) for n in want)
    assert loose.recovered_steps > 0 and loose.overflow_frames() >= loose.late_overflow_frames


def test_pipelined_loop_gives_the_plain_loop_s_frames(cuda_device):
    """PipelinedClosedLoop(depth=2): consecutive steps on alternating loops over one shared copy of the model, enqueued
    without waiting for each other.  Every step's frames are the plain loop's, bit for bit -- with host poses and
    wait=False (the throughput mode of bench.py), with device poses and the default wait, eager and under graph replay,
    with a wrist camera that moves every step."""
    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=120_000, seed=10)
    rollout = cl.xarm6_rollout()
    parts, actors = cl.xarm6_rollout_parts(rollout)
    cams = {"right_cam": scenes.sensor_camera("xarm6_align"),
            "wrist_cam": look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, 640, 480)}
    steps = 14
    poses = list(cl.rollout_poses(rollout, len(actors), steps=steps, seed=2))

    def wrist(k):
        return look_at_view([0.55 - 0.01 * k, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, 640, 480)

    plain = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev)
    plain.reset(*poses[0])
    want = []
    for k in range(1, steps):
        f = plain.step(*poses[k], cameras={"wrist_cam": wrist(k)})
        torch.cuda.synchronize()
        want.append({n: t.clone() for n, t in f.items()})
    for graph in (False, True):
        pipe = cl.PipelinedClosedLoop(raw, parts, cams, depth=2, scaled_parts=actors, device=dev)
        assert pipe.loops[1].xyz.data_ptr() == pipe.loops[0].xyz.data_ptr()  # one copy of the model
        pipe.reset(*poses[0])
        if graph:
            pipe.capture()
        got = []
        for k in range(1, steps):
            M, s = poses[k]
            if k % 2:  # host poses, nobody waits
                f = pipe.step(M.pin_memory(), s.pin_memory(), cameras={"wrist_cam": wrist(k)}, wait=False)
                pipe.wait_for(f)
            else:      # device poses, the current stream waits for the frames
                f = pipe.step(M.to(dev), s.to(dev), cameras={"wrist_cam": wrist(k)})
            got.append({n: t.clone() for n, t in f.items()})  # (cloned on the current stream, which waits for the step)
        torch.cuda.synchronize()
        for k, (g, w) in enumerate(zip(got, want)):
            for n in w:
                assert torch.equal(g[n], w[n]), f"graph={graph} step {k + 1} {n}"
        assert pipe.overflow_frames() == 0


def test_pipelined_loop_does_not_overwrite_frames_that_are_still_being_read(cuda_device):
    """Write-after-read across streams: the consumer stream reads step k's frames LATE (a spin kernel sits in front of the
    clone) while steps k + 1 and k + 2 are enqueued at once; with depth 2, step k + 2 renders into the buffers of step k on
    another stream and has to wait for that read."""
    if not hasattr(torch.cuda, "_sleep"):
        pytest.skip("torch.cuda._sleep not available")
    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=60_000, seed=11)
    rollout = cl.xarm6_rollout()
    parts, actors = cl.xarm6_rollout_parts(rollout)
    cams = {"right_cam": scenes.sensor_camera("xarm6_align")}
    poses = list(cl.rollout_poses(rollout, len(actors), steps=6, seed=3))
    plain = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev)
    plain.reset(*poses[0])
    want = plain.step(*poses[1])["right_cam"].clone()
    other = plain.step(*poses[3])["right_cam"].clone()
    torch.cuda.synchronize()
    assert not torch.equal(want, other), "the arm did not move between the steps: the test proves nothing"
    pipe = cl.PipelinedClosedLoop(raw, parts, cams, depth=2, scaled_parts=actors, device=dev)
    pipe.reset(*poses[0])
    pipe.capture()
    f = pipe.step(poses[1][0].pin_memory(), poses[1][1].pin_memory())   # the current stream waits for step 1 ...
    torch.cuda._sleep(200_000_000)                                       # ... and gets to its read ~0.1 s later
    late = f["right_cam"].clone()
    pipe.step(poses[2][0].pin_memory(), poses[2][1].pin_memory(), wait=False)
    pipe.step(poses[3][0].pin_memory(), poses[3][1].pin_memory(), wait=False)  # same loop as step 1
    torch.cuda.synchronize()
    assert torch.equal(late, want)
    assert torch.equal(pipe.loops[0].frames["right_cam"], other)


def test_a_tile_grid_beyond_the_default_path_keeps_the_caller_s_order(cuda_device):
    """a camera whose tile grid exceeds 16384 tiles takes the radix placement, whose frames are no
    inference frames and reject a permuted model (GsrInputs.orig_index).  The loop then keeps the model's order and
    hands over the block bounds alone -- every step renders, and equals the loop without a layout."""
    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=30_000, seed=4)
    W, H = 2304, 1920  # 144 x 120 = 17280 tiles
    cams = {"big": scenes.sensor_camera("xarm6_align", W, H)}
    parts, actors = cl.xarm6_parts()
    sim2gs = torch.tensor(scenes.SIM2GS_XARM_TRANS)
    poses = list(cl.random_walk_poses(sim2gs, len(parts), len(actors), steps=3, seed=2))
    loops = [cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev, layout=lay, bound_capacity=False)
             for lay in (True, False)]
    assert loops[0].layout is not None and loops[0].layout[1] is None  # bounds only, no permutation
    for lp in loops:
        lp.reset(*poses[0])
    for M, s in poses[1:]:
        a, b = (lp.step(M, s, ensure=True)["big"].clone() for lp in loops)
        assert torch.equal(a, b) and int(a.max()) > 100
    # at a size the default path takes, the same scene does get its Morton-ordered copy
    small = cl.ClosedLoopRenderer(raw, parts, {"cam": scenes.sensor_camera("xarm6_align", 640, 480)}, scaled_parts=actors,
                                  device=dev)
    assert small.layout is not None and small.layout[1] is not None


def test_a_rollout_frame_pair_with_moved_parts_meets_the_oracle(cuda_device, capsys):
    """configs[2], one step deep into the FK rollout (every link moved, both tracked actors moved and rescaled), both
    cameras, FULL size -- against the CPU oracle fed with the model the REFERENCE's glue builds for that step
    (oracle/wrapper_glue_ref.py + oracle/transform_ref.py on the host: deep copy, per-part transform_gaussians, masked
    write-backs).  Everywhere else the closed-loop tests compare two glues through the same HIP rasterizer; here a frame
    with moved parts meets the oracle's arithmetic end to end: float colour <= 1e-4 off the borderline pixels.
    (The two sides differ by an ulp here and there BEFORE the rasterizer -- torch's matmul against the kernel's fused
    multiply-adds in the rigid transform, torch.sigmoid against the device's -- so a handful of threshold decisions the
    oracle does not flag could still flip; none does on this step: every pixel off the oracle's borderline set within
    1e-4, measured 4.8e-7, and every pixel within 1e-3, measured 1.0e-5.)"""
    import numpy as np

    from oracle import gs_oracle as go
    from tests import helpers as hp

    dev = cuda_device
    rollout = cl.xarm6_rollout()
    raw, cams, parts, actors, loop, _ = _setup(dev, scenes.XARM6_ALIGN_NUM_GAUSSIANS, rollout=rollout, keep_float=True)
    poses = list(cl.rollout_poses(rollout, len(actors), steps=121, seed=0))
    M, s = poses[120]
    loop.reset(*poses[0])
    loop.step(M, s, ensure=True)
    got = {n: loop.multi.lanes[c]._out[0].cpu().numpy() for c, n in enumerate(loop.names)}
    got8 = {n: loop.frames[n][0].cpu() for n in loop.names}
    cpu = types.SimpleNamespace(_xyz=raw.xyz.clone(), _scaling=raw.scaling.clone(), _rotation=raw.rotation.clone(),
                                _opacity=raw.opacity.reshape(-1, 1, 1).clone(), _semantics=raw.semantics,
                                _features_dc=raw.features_dc, _features_rest=raw.features_rest)
    moved = ref.transform_parts(cpu, parts, M[None], s[None], actors)
    gs = ref.assemble_env(cpu, parts, moved, 0, 1)
    assert not torch.equal(gs._xyz, raw.xyz) and not torch.equal(gs._scaling, raw.scaling)  # links moved, actors rescaled
    # scales and rotations: the canonical activations the fused path evaluates inside preprocess; opacity: torch.sigmoid,
    # as the loop activates it once at load time
    _, sc, rot = go.activate_params(None, gs._scaling.numpy(), gs._rotation.numpy(), flags=6)
    shs = torch.cat((gs._features_dc, gs._features_rest), dim=1).numpy()
    opac = torch.sigmoid(gs._opacity.reshape(-1)).numpy()
    lines = []
    for name, cam in cams.items():
        cam = cam.to("cpu")
        inp = dict(means3D=gs._xyz.numpy(), shs=shs, opacities=opac, scales=sc, rotations=rot,
                   viewmatrix=cam.world_view_transform.numpy().reshape(-1),
                   projmatrix=cam.full_proj_transform.numpy().reshape(-1), campos=cam.camera_center.numpy())
        o = hp.oracle_forward(inp, hp.oracle_settings(cam), np.zeros(3, np.float32))
        d = np.abs(got[name] - o["color"]).max(0)
        b = o["borderline"] != 0
        over = int((d[~b] > 1e-4).sum())
        lines.append(f"{name}: V {int((o['geom']['radii'] > 0).sum())} R {int(o['binning']['num_rendered'])} worst off "
                     f"borderline {float(d[~b].max()):.3e} ({over} pixels > 1e-4), all pixels {float(d.max()):.3e}, "
                     f"borderline {int(b.sum())}")
        assert over == 0, lines[-1]                 # (measured: 4.8e-7 off the borderline pixels, 1.0e-5 on all)
        assert float(d.max()) <= 1e-3, lines[-1]
        want8 = (torch.from_numpy(o["color"]).clamp(0, 1).permute(1, 2, 0) * 255).clamp(0, 255).to(torch.uint8)
        d8 = (want8.to(torch.int16) - got8[name].to(torch.int16)).abs()
        assert int(d8.max()) <= 1 and int(want8.max()) > 100
    with capsys.disabled():
        print("\n[configs[2] step 120 against the oracle] " + "; ".join(lines))


def test_host_values_staged_inside_the_graph_give_the_eager_loop_s_frames(cuda_device):
    """Round 6: a loop whose poses and cameras arrive on the host captures ONE GRAPH PER PINNED RING SLOT, each starting with
    the gsr_stage_step launch that reads its slot (the step's part matrices, scales and camera matrices -> the device
    vector the kernels read, and the 17-float pose table packed on the way): nothing is enqueued between two replays.
    More steps than slots (the ring wraps), a wrist camera that moves every step: every frame equals the frame of a loop
    that never captured, and the pose table the staging kernel packed equals gsr_pack_part_transforms of the same matrices
    bit for bit.  A pose handed over as a DEVICE tensor is never staged over: the loop drops to one graph of the frames."""
    dev = cuda_device
    raw, cams, parts, actors, loop, _ = _setup(dev, 120_000, seed=6)
    eager = cl.ClosedLoopRenderer(raw, parts, {k: v.to("cpu") for k, v in cams.items()}, scaled_parts=actors, device=dev)
    sim2gs = torch.tensor(scenes.SIM2GS_XARM_TRANS)
    poses = list(cl.random_walk_poses(sim2gs, len(parts), len(actors), steps=14, seed=5))

    def wrist(k):
        return look_at_view([0.55 - 0.01 * k, 0.35, 0.25 + 0.005 * k], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448,
                            640, 480)

    loop.reset(*poses[0])
    eager.reset(*poses[0])
    loop.capture()
    assert loop._graphs is not None and len(loop._graphs) == len(loop._ring) and loop._graph is None
    from gsworld_amd.transform import FusedPartTransform

    packer = FusedPartTransform(parts, raw.semantics.to(dev), scaled_parts=actors)
    for k, (M, s) in enumerate(poses[1:]):
        w = wrist(k)
        # (odd steps: waited for -> the slot's graph is replayed; even steps: enqueued ahead -> the same launches one by one
        #  from the argument pack, the host values staged by the same kernel -- both walk the one ring of slots)
        got = {n: f.clone() for n, f in loop.step(M, s, cameras={"wrist_cam": w}, ensure=bool(k & 1)).items()}
        want = eager.step(M, s, cameras={"wrist_cam": w})
        torch.cuda.synchronize()
        for n in want:
            assert torch.equal(got[n], want[n]), f"step {k}, {n}"
        table = packer.pack_on_device(M.to(dev, torch.float32).contiguous(), s.to(dev, torch.float32).contiguous())
        assert torch.equal(loop._table.view(torch.int32), table.view(torch.int32)), f"step {k}: pose table"
    assert loop._ring_k > len(loop._ring)  # the ring wrapped
    # device poses: staged graphs would write the host mirror over them -- the loop re-captures without the staging
    M, s = poses[3]
    got = {n: f.clone() for n, f in loop.step(M.to(dev), s.to(dev), cameras={"wrist_cam": wrist(2)}).items()}
    want = eager.step(M, s, cameras={"wrist_cam": wrist(2)})
    torch.cuda.synchronize()
    assert loop._graphs is None and loop._graph is not None
    for n in want:
        assert torch.equal(got[n], want[n]), n
    # ... and back on the host (scales not handed over: read back once), same frames
    M2, _ = poses[5]
    got = {n: f.clone() for n, f in loop.step(M2).items()}
    want = eager.step(M2, s)
    torch.cuda.synchronize()
    for n in want:
        assert torch.equal(got[n], want[n]), n


def test_block_cache_keeps_what_did_not_move_and_every_frame_is_the_uncached_one(cuda_device):
    """Inference frames whose caller vouches for the model (GSR_MODEL_VERSION in param_space: ClosedLoopRenderer owns its copy)
    leave a block of 256 Gaussians as the previous frame on the state computed it when camera, settings and the block's pose
    row are that frame's bit for bit (csrc/preprocess.hip prep_block_cached).  Under the fixed right_cam everything that is
    no robot link and no tracked object keeps its records; under a wrist camera that moves nothing does.  Eager and under
    graph replay, with a camera that rests, moves and rests again, and with poses that stand still for a step: every frame
    is the frame of a loop built with ``block_cache=False``, byte for byte."""
    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=300_000, seed=33)
    rollout = cl.xarm6_rollout()
    parts, actors = cl.xarm6_rollout_parts(rollout)
    cams = {"right_cam": scenes.sensor_camera("xarm6_align"),
            "wrist_cam": look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, 640, 480)}
    steps = 18
    poses = list(cl.rollout_poses(rollout, len(actors), steps=steps, seed=4))

    def wrist(k):  # rests for the first steps, then moves, then rests again
        a = 0.05 * min(max(k - 5, 0), 6)
        return look_at_view([0.55 - 0.10 * math.sin(a), 0.35, 0.25 + 0.05 * math.sin(2.0 * a)], [0.35, 0.05, 0.05], [0, 0, 1],
                            0.9715089, 0.7551448, 640, 480)

    for captured in (False, True):
        cached = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev)
        plain = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev, block_cache=False)
        assert cached._model_version != 0 and plain._model_version == 0
        cached.reset(*poses[0])
        plain.reset(*poses[0])
        if captured:
            cached.capture()
            plain.capture()
        kept = {n: 0 for n in cams}
        tiles = {n: 0 for n in cams}
        for k in range(1, steps):
            M, s = poses[k if k % 5 else k - 1]  # (every fifth step repeats the poses of the step before: nothing moved)
            got = cached.step(M, s, cameras={"wrist_cam": wrist(k)}, ensure=True)
            want = plain.step(M, s, cameras={"wrist_cam": wrist(k)}, ensure=True)
            for n in cams:
                assert torch.equal(got[n], want[n]), f"captured={captured} step {k} {n}"
            for n, lane in zip(cached.names, cached.multi.lanes):
                kept[n] += int(dbg.sort_state(lane.geom)["kept_blocks"])
                tiles[n] += int(dbg.sort_state(lane.geom)["kept_tiles"])
            assert not any(dbg.sort_state(lane.geom)["kept_blocks"] for lane in plain.multi.lanes)
        assert kept["right_cam"] >= steps - 3, kept       # the fixed camera keeps its static blocks on (nearly) every step
        assert 0 < kept["wrist_cam"] < steps - 1, kept    # the wrist camera only while it rests
        # ... and the compositor leaves the tiles none of the recomputed Gaussians touches (render.hip, tile reuse): the uint8
        # frame already holds their pixels
        assert tiles["right_cam"] >= steps - 3 and 0 < tiles["wrist_cam"] < steps - 1, tiles


def test_tile_reuse_skips_only_what_nothing_touched_and_repaints_when_the_frame_s_inputs_change(cuda_device):
    """GSR_FRAME_KEPT (ClosedLoopRenderer(tile_reuse=True), the default): the loop's uint8 frames are its own, and a 16 x 16
    tile that no recomputed Gaussian touches under an unchanged camera and background is not composited again
    (csrc/render.hip "tile reuse").  Shown from the outside by breaking the promise on purpose: a frame filled with a marker
    byte keeps it exactly in the tiles the step did not have to draw -- everywhere else, and everywhere at all once the
    background changes or the loop is built with ``tile_reuse=False``, the bytes are those of a loop that keeps nothing."""
    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=300_000, seed=34)
    rollout = cl.xarm6_rollout()
    parts, actors = cl.xarm6_rollout_parts(rollout)
    cams = {"right_cam": scenes.sensor_camera("xarm6_align"),
            "wrist_cam": look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, 640, 480)}
    # (the surrogate scene scatters its 17 moving parts over the whole view and the rollout moves every one of them every
    #  step -- hardly a tile is left then; here the arm rests and the gripper's last links and the two objects move)
    walk = list(cl.rollout_poses(rollout, len(actors), steps=8, seed=5))
    poses = []
    for M, s in walk:
        M0, s0 = walk[0][0].clone(), walk[0][1].clone()
        M0[-4:], s0[-4:] = M[-4:], s[-4:]
        poses.append((M0, s0))
    MARK = 7
    for captured in (False, True):
        reuse = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev)
        every = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev, tile_reuse=False)
        plain = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev, block_cache=False)
        loops = (reuse, every, plain)
        for lp in loops:
            lp.reset(*poses[0])
            if captured:
                lp.capture()
        for k in (1, 2, 3):
            for lp in loops:
                lp.step(*poses[k], ensure=True)
        for lp in (reuse, every):  # (the promise broken on purpose)
            for n in cams:
                lp.frames[n].fill_(MARK)
        got, all_tiles, want = (lp.step(*poses[4], ensure=True) for lp in loops)
        for n in cams:
            assert torch.equal(all_tiles[n], want[n]), f"captured={captured} {n}: tile_reuse=False composites every tile"
            g, w = got[n][0], want[n][0]
            H, W = g.shape[:2]
            tiles = lambda x: x.reshape(H // 16, 16, W // 16, 16, 3).permute(0, 2, 1, 3, 4).reshape(H // 16, W // 16, -1)  # noqa: E731
            left = (tiles(g) == MARK).all(dim=2)
            drawn = (tiles(g) == tiles(w)).all(dim=2)
            assert bool((left | drawn).all()), f"captured={captured} {n}: a tile is either left whole or drawn whole"
            # both cameras stand still: everything the arm and the objects do not cover is left (most of either frame), and
            # what they cover is drawn
            assert (0.4 * left.numel() if n == "right_cam" else 0) < int(left.sum()) < left.numel(), (n, int(left.sum()))
            assert any(dbg.sort_state(lane.geom)["kept_tiles"] for lane in reuse.multi.lanes)
        assert not any(dbg.sort_state(lane.geom)["kept_tiles"] for lane in every.multi.lanes)
        # another background: every pixel may depend on it, every tile is drawn -- the marker is gone
        for lp in loops:
            lp.bg.copy_(torch.tensor([0.25, 0.5, 0.75], device=dev))
        got, _, want = (lp.step(*poses[5], ensure=True) for lp in loops)
        for n in cams:
            assert torch.equal(got[n], want[n]), f"captured={captured} {n}: background changed"
        # ... and from there on the loop keeps its promise again: frames equal while tiles are left
        for k in (6, 7):
            got, _, want = (lp.step(*poses[k], ensure=True) for lp in loops)
            for n in cams:
                assert torch.equal(got[n], want[n]), f"captured={captured} step {k} {n}"
            assert any(dbg.sort_state(lane.geom)["kept_tiles"] for lane in reuse.multi.lanes)


@pytest.mark.parametrize("E", [1, 2])
def test_tile_reuse_random_rests_and_moves_give_the_frames_of_a_loop_that_keeps_nothing(cuda_device, E):
    """A seeded soak of what the block cache and the tile reuse key on: per step a random subset of the parts moves (sometimes
    none, sometimes all), the wrist camera rests or moves, now and then the background changes -- every frame of every
    environment is the frame of a loop built with ``block_cache=False``, eager for the first half and under graph replay for
    the second."""
    import random

    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=200_000, seed=35)
    rollout = cl.xarm6_rollout()
    parts, actors = cl.xarm6_rollout_parts(rollout)
    cams = {"right_cam": scenes.sensor_camera("xarm6_align"),
            "wrist_cam": look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, 640, 480)}
    steps = 64
    walk = list(cl.rollout_poses(rollout, len(actors), steps=steps, seed=6, num_envs=E))
    rng = random.Random(1234 + E)
    K = walk[0][0].shape[-3]

    def wrist(a):
        return look_at_view([0.55 - 0.10 * math.sin(a), 0.35, 0.25 + 0.05 * math.sin(2.0 * a)], [0.35, 0.05, 0.05], [0, 0, 1],
                            0.9715089, 0.7551448, 640, 480)

    reuse = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev, num_envs=E)
    plain = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev, num_envs=E, block_cache=False)
    M, s = walk[0][0].clone(), walk[0][1].clone()
    for lp in (reuse, plain):
        lp.reset(M, s)
    angle, skipped = 0.0, 0
    for k in range(1, steps):
        if k == steps // 2:
            for lp in (reuse, plain):
                lp.capture()
        mode = rng.random()
        moving = [] if mode < 0.2 else (range(K) if mode > 0.85 else rng.sample(range(K), rng.randint(1, 4)))
        for p in moving:  # (the chosen parts take the rollout's pose of this step, in every environment; the others rest)
            M[..., p, :, :], s[..., p] = walk[k][0][..., p, :, :], walk[k][1][..., p]
        if rng.random() < 0.3:
            angle += 0.05
        if rng.random() < 0.08:
            bg = torch.tensor([rng.random(), rng.random(), rng.random()], device=dev)
            for lp in (reuse, plain):
                lp.bg.copy_(bg)
        got = reuse.step(M.clone(), s.clone(), cameras={"wrist_cam": wrist(angle)}, ensure=True)
        want = plain.step(M.clone(), s.clone(), cameras={"wrist_cam": wrist(angle)}, ensure=True)
        for n in cams:
            assert torch.equal(got[n], want[n]), f"E={E} step {k} {n} (moving {list(moving)})"
        skipped += sum(int(dbg.sort_state(lane.geom)["kept_tiles"]) for lane in reuse.multi.lanes)
    assert skipped > steps // 2, skipped  # (the soak did exercise the reuse)


def test_arm_shaped_scene_keeps_blocks_and_tiles_and_gives_the_frames_of_a_loop_that_keeps_nothing(cuda_device):
    """scenes.arm_tabletop_scene: the robot's Gaussians on the links of the reference's URDF (the rollout fixture's scan pose),
    two objects on the table -- under the forward-kinematics rollout an arm then moves the way an arm does, and the fixed
    sensor camera leaves a good part of its tiles alone on every step.  Frames byte for byte those of a loop built with
    ``block_cache=False``; one and two environments."""
    dev = cuda_device
    rollout = cl.xarm6_rollout()
    raw = scenes.arm_tabletop_scene(rollout["link_scan"], rollout["labels"], n=250_000, seed=3)
    labels = set(int(v) for v in raw.semantics.reshape(-1).unique().tolist())
    assert labels == set(range(0, 19))  # background, the 16 link labels, the two objects
    parts, actors = cl.xarm6_rollout_parts(rollout)
    cams = {"right_cam": scenes.sensor_camera("xarm6_align"),
            "wrist_cam": look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, 640, 480)}
    for E in (1, 2):
        poses = list(cl.rollout_poses(rollout, len(actors), steps=10, seed=9, num_envs=E))
        kept = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev, num_envs=E)
        plain = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev, num_envs=E, block_cache=False)
        for lp in (kept, plain):
            lp.reset(*poses[0])
            lp.capture()
        left = 0
        for k in range(1, 10):
            got = kept.step(*poses[k], ensure=True)
            want = plain.step(*poses[k], ensure=True)
            for n in cams:
                assert torch.equal(got[n], want[n]), f"E={E} step {k} {n}"
            left += sum(int(dbg.sort_state(lane.geom)["kept_tiles"]) for lane in kept.multi.lanes)
        assert left >= 9 * E  # (at least the fixed camera of every environment, every step)
        assert kept.overflow_frames() == 0


def test_steps_of_eight_frames_and_more_go_as_sets_on_streams_of_their_own(cuda_device):
    """Three environments and more (six frames per step) are rendered as at least two sets of launches in flight at once
    (ClosedLoopRenderer -> MultiCameraRenderer.set_frames / max_set_streams).  Same frames as one set after the other on one
    stream, eager and under graph replay; five environments = two sets of five."""
    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=120_000, seed=37)
    rollout = cl.xarm6_rollout()
    parts, actors = cl.xarm6_rollout_parts(rollout)
    cams = {"right_cam": scenes.sensor_camera("xarm6_align"),
            "wrist_cam": look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, 640, 480)}
    for E, per_set in ((3, 3), (4, 4), (5, 5), (9, 8)):
        poses = list(cl.rollout_poses(rollout, len(actors), steps=6, seed=10, num_envs=E))
        sets = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev, num_envs=E)
        one = cl.ClosedLoopRenderer(raw, parts, cams, scaled_parts=actors, device=dev, num_envs=E)
        assert sets.multi.set_frames == per_set
        one.multi.max_set_streams, one.multi.set_frames = 1, 8
        for lp in (sets, one):
            lp.reset(*poses[0])
        for k in range(1, 6):
            if k == 3:
                for lp in (sets, one):
                    lp.capture()
            got = sets.step(*poses[k], ensure=True)
            want = one.step(*poses[k], ensure=True)
            for n in cams:
                assert torch.equal(got[n], want[n]), f"E={E} step {k} {n}"
        assert sets.overflow_frames() == one.overflow_frames() == 0
