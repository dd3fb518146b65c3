"""CPU suite (-m "not gpu"): golden vectors captured from the reference's own Python (tools/make_golden.py), the
host-side mirrors of the reference interface, and the C-ABI library's exported symbols (no compute calls)."""
import ctypes
import math
import os

import numpy as np
import pytest
import torch

from gsworld_amd import _lib, camera, scenes, transform
from oracle import transform_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_cabi_library_loads_and_exports_every_declared_symbol():
    lib = _lib.lib()
    names = _lib.exported_symbols()
    assert "gsr_forward" in names and "gsr_mark_visible" in names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert b"gfx950" in lib.gsr_version()
    # sizes are pure host arithmetic: monotone and 256-byte aligned
    g1, g2 = lib.gsr_geom_bytes(1000, 640, 480), lib.gsr_geom_bytes(2000, 640, 480)
    assert 0 < g1 < g2 and g1 % 256 == 0
    assert lib.gsr_binning_bytes(10_000) % 256 == 0 and lib.gsr_image_bytes(640, 480) % 256 == 0
    # argument validation never touches the GPU
    assert lib.gsr_forward(None, None, None, None, 0, None, None) == _lib.GSR_E_INVALID
    assert b"null" in lib.gsr_last_error()
    assert lib.gsr_profile_enable(7) == _lib.GSR_E_INVALID


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libgsr_hip.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_rasterizer_rejects_cpu_tensors():
    from gsworld_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

    cam = scenes.identity_camera(32, 32)
    rs = GaussianRasterizationSettings(32, 32, cam.tanfovx, cam.tanfovy, torch.zeros(3), 1.0,
                                       cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center,
                                       False, False, False)
    r = GaussianRasterizer(rs)
    m = torch.zeros(4, 3)
    with pytest.raises(Exception, match="excatly one of either SHs"):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1))
    with pytest.raises(Exception, match="excatly one of either SHs"):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1), shs=torch.zeros(4, 16, 3), colors_precomp=m)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1), shs=torch.zeros(4, 16, 3), scales=m)
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1), shs=torch.zeros(4, 16, 3), scales=m,
          rotations=torch.ones(4, 4))
    with pytest.raises(RuntimeError, match=r"means3D must have dimensions \(num_points, 3\)"):
        r(means3D=torch.zeros(4, 2), means2D=m, opacities=torch.ones(4, 1), shs=torch.zeros(4, 16, 3), scales=m,
          rotations=torch.ones(4, 4))
    # GSWorld probes this name to choose the separate_sh call path (gs_world_wrapper.py:22-26): must be absent
    import gsworld_amd.rasterizer as rz
    assert not hasattr(rz, "SparseGaussianAdam")


def test_constants_match_reference_capture():
    ref = np.load(os.path.join(GOLD, "reference_constants.npz"))
    np.testing.assert_array_equal(scenes.SIM2GS_ARM_TRANS, ref["sim2gs_arm_trans"])
    np.testing.assert_array_equal(scenes.SIM2GS_XARM_TRANS, ref["sim2gs_xarm_trans"])
    np.testing.assert_array_equal(scenes.RS_D435I_RGB_K, ref["rs_d435i_rgb_k"])
    np.testing.assert_array_equal(scenes.RIGHT2BASE, ref["right2base"])
    np.testing.assert_array_equal(scenes.XARM_RIGHT2BASE, ref["xarm_right2base"])


def test_extract_rigid_transform_golden():
    ref = np.load(os.path.join(GOLD, "extract_rigid_transform.npz"))
    for i in range(3):
        rigid, scale, R, t = camera.extract_rigid_transform(torch.from_numpy(ref["single_in"][i]))
        np.testing.assert_allclose(rigid.numpy(), ref["single_rigid"][i], atol=1e-6)
        np.testing.assert_allclose(scale.numpy(), ref["single_scale"][i], atol=1e-6)
    # scale of sim2gs_xarm_trans = 1.0016118, first rigid row (-0.9685, 0.2244, 0.1082, 0.3279)
    rigid, scale, _, _ = camera.extract_rigid_transform(torch.tensor(scenes.SIM2GS_XARM_TRANS))
    assert abs(float(scale) - 1.0016118) < 1e-6
    np.testing.assert_allclose(rigid[0].numpy(), [-0.9685, 0.2244, 0.1082, 0.3279], atol=1e-4)
    rigid, scale, R, t = camera.extract_rigid_transform(torch.from_numpy(ref["batch_in"]))
    np.testing.assert_allclose(rigid.numpy(), ref["batch_rigid"], atol=1e-5)
    np.testing.assert_allclose(scale.numpy(), ref["batch_scale"], atol=1e-6)
    np.testing.assert_allclose(R.numpy(), ref["batch_R"], atol=1e-5)
    np.testing.assert_allclose(t.numpy(), ref["batch_t"], atol=0)
    with pytest.raises(ValueError):
        camera.extract_rigid_transform(torch.zeros(3, 3))


def test_transform_gaussians_golden():
    ref = np.load(os.path.join(GOLD, "transform_gaussians.npz"))
    import types
    g = types.SimpleNamespace(_xyz=torch.from_numpy(ref["xyz"]), _scaling=torch.from_numpy(ref["scaling"]),
                              _rotation=torch.from_numpy(ref["rotation"]), _opacity=torch.from_numpy(ref["opacity"]))
    sel = torch.from_numpy(ref["selected"])
    cases = sorted({k.split(".")[0] for k in ref.files if ".out." in k})
    assert cases == ["actor_env1", "actor_env2", "link_env1", "link_env3", "translate_vec3"]
    for c in cases:
        kw = {k: (torch.from_numpy(ref[f"{c}.in.{k}"]) if f"{c}.in.{k}" in ref.files else None)
              for k in ("scale", "rot_mat", "translation")}
        out = transform_ref.transform_gaussians(g, sel, **kw)
        for name, got in zip(("xyz", "scaling", "rotation", "opacity"), out):
            want = ref[f"{c}.out.{name}"]
            assert tuple(got.shape) == want.shape, (c, name, got.shape, want.shape)
            np.testing.assert_allclose(got.numpy(), want, atol=2e-6, rtol=1e-6, err_msg=f"{c}.{name}")
    # the shapes the wrapper's `shape[0] == num_envs` tests depend on (gs_world_wrapper.py:246-265), num_envs = 1
    assert ref["link_env1.out.xyz"].shape == (1, 300, 3) and ref["link_env1.out.rotation"].shape == (1, 300, 4)
    assert ref["link_env1.out.scaling"].shape == (300, 3)
    np.testing.assert_allclose(transform_ref.inverse_sigmoid(torch.from_numpy(ref["inverse_sigmoid.in"])).numpy(),
                               ref["inverse_sigmoid.out"], atol=1e-6)


def test_quaternion_helpers_roundtrip():
    gen = torch.Generator().manual_seed(3)
    q = torch.randn(64, 4, generator=gen)
    q = transform.standardize_quaternion(q / q.norm(dim=1, keepdim=True))
    w, x, y, z = q.unbind(1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    np.testing.assert_allclose(transform.matrix_to_quaternion(R).numpy(), q.numpy(), atol=1e-5)
    ident = torch.tensor([[1.0, 0, 0, 0]]).expand(64, 4)
    np.testing.assert_allclose(transform.quaternion_multiply(ident, q).numpy(), q.numpy(), atol=1e-7)


def test_sensor_camera_matches_survey_values():
    """cam_maniskill2gs math (gs_world_wrapper.py:280-301) on the right_cam calibration: FoVs of SURVEY.md 8c."""
    cam = scenes.sensor_camera("xarm6_align")
    assert abs(cam.FoVx - 0.9715089) < 1e-6 and abs(cam.FoVy - 0.7551448) < 1e-6
    assert abs(cam.tanfovx - 0.527947) < 1e-6 and abs(cam.tanfovy - 0.3966005) < 1e-6
    assert (cam.image_width, cam.image_height) == (640, 480)
    V = cam.world_view_transform  # W2C^T
    W2C = V.T
    np.testing.assert_allclose((W2C[:3, :3] @ W2C[:3, :3].T).numpy(), np.eye(3), atol=1e-5)
    assert abs(float(torch.det(W2C[:3, :3])) - 1.0) < 1e-5
    # camera centre = -R^T t and equals the calibrated pose pushed through scale + rigid sim->GS
    c = -(W2C[:3, :3].T @ W2C[:3, 3])
    np.testing.assert_allclose(cam.camera_center.numpy(), c.numpy(), atol=1e-5)
    rigid, scale, _, _ = camera.extract_rigid_transform(torch.tensor(scenes.SIM2GS_XARM_TRANS))
    p_sim = torch.tensor(scenes.XARM_RIGHT2BASE[:3, 3]) + torch.tensor([0.0, 0.0, 0.03])
    want = rigid[:3, :3] @ (p_sim * scale) + rigid[:3, 3]
    np.testing.assert_allclose(cam.camera_center.numpy(), want.numpy(), atol=1e-5)
    # projection: P[0,0] = 1/tanfovx, P[1,1] = 1/tanfovy, w = z; full = view @ proj
    P = camera.get_projection_matrix(0.01, 100.0, cam.FoVx, cam.FoVy)
    assert abs(float(P[0, 0]) - 1 / cam.tanfovx) < 1e-5 and abs(float(P[1, 1]) - 1 / cam.tanfovy) < 1e-5
    assert float(P[3, 2]) == 1.0 and abs(float(P[2, 2]) - 100.0 / 99.99) < 1e-6
    np.testing.assert_allclose(cam.full_proj_transform.numpy(), (V @ P.T).numpy(), atol=1e-6)


def test_scene_generators_are_seeded_and_shaped():
    a = scenes.tabletop_scene("xarm6_align", n=5000, seed=1)
    b = scenes.tabletop_scene("xarm6_align", n=5000, seed=1)
    c = scenes.tabletop_scene("fr3_align", n=5000, seed=2)
    assert torch.equal(a.xyz, b.xyz) and not torch.equal(a.xyz[:, 0], c.xyz[:, 0])
    assert a.features_dc.shape == (5000, 1, 3) and a.features_rest.shape == (5000, 15, 3)
    assert a.opacity.shape == (5000, 1) and a.scaling.shape == (5000, 3) and a.rotation.shape == (5000, 4)
    means, shs, op, sc, rot = a.activated()
    assert shs.shape == (5000, 16, 3) and shs.is_contiguous()
    assert float(op.min()) > 0 and float(op.max()) < 1 and float(sc.min()) > 0
    np.testing.assert_allclose(rot.norm(dim=1).numpy(), 1.0, atol=1e-5)
    assert scenes.XARM6_ALIGN_NUM_GAUSSIANS == 1_468_850 and len(scenes.SCENE_NAMES) == 8
    r = scenes.random_scene_camera_frame(1000, seed=0)
    assert int(((r.xyz[:, 2] > 0.05) & (r.xyz[:, 2] < 0.2)).sum()) == 10  # the 1 % that exercises the 0.05f cull


def test_frame_stats_algorithmic_bytes():
    from gsworld_amd.renderer import FrameStats

    s = FrameStats(1_468_850, 881_310, 3_525_240, False)
    # SURVEY.md 8d worked example: 70.5 + 246.8 + 225.6 + 4.9 = 548 MB
    assert abs(s.algorithmic_bytes(640, 480) / 1e6 - 547.8) < 0.5


def test_tuning_env_selects_the_ab_paths_and_rejects_unknown_names():
    """GSWORLD_AMD_TUNING="depth_sort=1,binning_path=3" presets the A/B selectors every GsrSettings of this package
    carries (tools and bench.py runs without editing them); an unknown selector is an error, not a silent no-op."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "from gsworld_amd import _lib; print(_lib.TUNING['depth_sort'], _lib.TUNING['binning_path'])"
    env = dict(os.environ, GSWORLD_AMD_TUNING="depth_sort=1,binning_path=3", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=root)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["1", "3"]
    env["GSWORLD_AMD_TUNING"] = "no_such_selector=1"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=root)
    assert out.returncode != 0 and "unknown selector" in out.stderr


def test_closed_loop_pose_generator_is_seeded_and_shaped():
    """The stand-in for the simulator (closed_loop.random_walk_poses): same seed -> same poses, (K,4,4) / (E,K,4,4)
    matrices with unit scales for links and a per-step scale for the tracked actors, rigid up to that scale."""
    import torch

    from gsworld_amd import closed_loop as cl, scenes

    parts, actors = cl.xarm6_parts()
    sim2gs = torch.tensor(scenes.SIM2GS_XARM_TRANS)
    a = list(cl.random_walk_poses(sim2gs, len(parts), len(actors), steps=4, seed=3))
    b = list(cl.random_walk_poses(sim2gs, len(parts), len(actors), steps=4, seed=3))
    c = list(cl.random_walk_poses(sim2gs, len(parts), len(actors), steps=4, seed=4))
    assert all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) for x, y in zip(a, b))
    assert not torch.equal(a[-1][0], c[-1][0])
    M, s = a[1]
    assert M.shape == (len(parts), 4, 4) and s.shape[0] == len(parts)
    assert torch.allclose(M[:, 3], torch.tensor([0.0, 0.0, 0.0, 1.0]).expand(len(parts), 4))
    E = list(cl.random_walk_poses(sim2gs, len(parts), len(actors), steps=2, seed=3, num_envs=3))
    assert E[0][0].shape == (3, len(parts), 4, 4)
    assert not torch.equal(a[0][0], a[-1][0])  # the poses do walk


def test_xarm6_rollout_fixture_is_a_kinematic_chain():
    """tests/golden/xarm6_rollout.npz (tools/make_xarm6_rollout.py: forward kinematics of the reference's xarm6 URDF
    along a seeded random-action rollout, 1 reset + 200 steps).  What must hold of ANY valid run of that chain:
    every link pose is rigid; the base never moves; a fixed joint keeps parent and child together
    (``gripper_fix`` has a zero origin: xarm_gripper_base_link == link6); joint1 turns about z, so link1's origin stays
    0.267 m above the base; no joint moves faster than its URDF velocity limit allows per control step; the scan pose is
    the FK of ``xarm_gs_qpos``, which differs from the reset pose only in joints 3 and 5 (constants.py:75-103)."""
    import numpy as np
    import torch

    from gsworld_amd import closed_loop as cl

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "xarm6_rollout.npz"))
    names = [str(n) for n in z["link_names"]]
    now, scan, q = z["link_now"].astype(np.float64), z["link_scan"].astype(np.float64), z["qpos"]
    assert now.shape == (201, 16, 4, 4) and scan.shape == (16, 4, 4) and q.shape == (201, 12)
    R = now[..., :3, :3]
    assert np.abs(R @ R.swapaxes(-1, -2) - np.eye(3)).max() < 1e-5 and np.allclose(np.linalg.det(R), 1, atol=1e-5)
    assert np.allclose(now[..., 3, :], [0, 0, 0, 1])
    for fixed in ("world", "link_base"):
        assert np.allclose(now[:, names.index(fixed)], np.eye(4))
    assert np.allclose(now[:, names.index("xarm_gripper_base_link")], now[:, names.index("link6")])
    assert np.allclose(now[:, names.index("link1"), :3, 3], [0, 0, 0.267], atol=1e-6)
    dq = np.abs(np.diff(q, axis=0))
    assert dq[:, :6].max() <= 3.14 / 20 + 1e-6 and dq[:, 6:].max() <= 2.0 / 20 + 1e-6 and dq[:, :6].max() > 0.1
    assert np.allclose(q[:, 6:], q[:, 6:7])  # the six finger joints mimic the drive joint
    changed = np.nonzero(np.abs(z["qpos_scan"] - q[0]) > 1e-6)[0].tolist()
    assert changed == [2, 4]
    # joint3 at -pi/3 instead of -pi/4: link3 turns by 15 degrees about its own z relative to link2
    rel = lambda P, a, b: np.linalg.inv(P[names.index(a)]) @ P[names.index(b)]  # noqa: E731
    turn = np.linalg.inv(rel(scan, "link2", "link3")) @ rel(now[0], "link2", "link3")
    assert np.allclose(turn[:3, 3], 0, atol=1e-6) and np.isclose(np.arctan2(turn[1, 0], turn[0, 0]), -np.pi / 12, atol=1e-5)

    # the generator built on it: links through the wrapper's pose arithmetic, actors random-walk; seeded; envs differ
    r = cl.xarm6_rollout()
    parts, actors = cl.xarm6_rollout_parts(r)
    assert parts["link6"] == [7, 8] and "world" not in parts and len(parts) == 17
    assert sorted(v for x in parts.values() for v in (x if isinstance(x, list) else [x])) == list(range(1, 19))
    a = list(cl.rollout_poses(r, len(actors), steps=201, seed=0))
    b = list(cl.rollout_poses(r, len(actors), steps=3, seed=0))
    assert all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) for x, y in zip(a, b))
    M, s = a[200]
    assert M.shape == (17, 4, 4) and s.shape == (17,) and torch.all(s[:15] == 1)
    RR = M[:15, :3, :3]
    assert (RR @ RR.transpose(1, 2) - torch.eye(3)).abs().max() < 1e-5       # sim2gs's scale cancels: links stay rigid
    want, _ = cl.part_poses_from_sim(r["sim2gs_arm"], r["link_now"][200][None], r["link_scan"], r["link_offset"])
    assert torch.equal(M[:15], want[0])
    e3 = list(cl.rollout_poses(r, len(actors), steps=2, seed=0, num_envs=3))
    assert e3[0][0].shape == (3, 17, 4, 4) and e3[0][1].shape == (3, 17)
    assert torch.equal(e3[0][0][0, :15], a[0][0][:15]) and torch.equal(e3[0][0][1, :15], a[17][0][:15])


def test_instance_bound_that_replaces_the_read_back():
    """_C.nosync_capacity: P x tiles -- a Gaussian touches at most every tile, so no frame exceeds it -- while the list
    (4 B per instance) fits the budget, the offsets stay 31-bit and the grid is one the counting placement takes;
    None otherwise (the caller then reads num_rendered back as upstream does)."""
    from gsworld_amd import _C, _lib

    assert _C.nosync_capacity(500_000, 800, 800) == 500_000 * 50 * 50             # configs[4]: 5 GB
    assert _C.nosync_capacity(1_468_850, 480, 640) == 1_468_850 * 40 * 30         # configs[1]: 7 GB
    assert _C.nosync_capacity(1_468_850, 1080, 1920) is None                      # 12 G instances
    assert _C.nosync_capacity(0, 480, 640) is None
    assert _C.nosync_capacity(10, 16, 16 * 257) is None                           # 257 tile columns
    assert _C.nosync_capacity(10, 17, 33) == 10 * 3 * 2                           # partial tiles count
    assert _C.nosync_capacity(10, 16 * 129, 16 * 128) is None                     # 16 512 tiles > GSR_MAX_COUNT_TILES:
    assert _C.nosync_capacity(10, 16 * 128, 16 * 128) == 10 * 16384               #   the radix placement is 24 B / instance
    saved = dict(_lib.TUNING)
    try:
        _lib.TUNING["binning_path"] = 2
        assert _C.nosync_capacity(1000, 64, 64) is None                           # A/B paths size their lists exactly
    finally:
        _lib.TUNING.update(saved)


def test_a_forced_inference_frame_never_reaches_the_training_paths():
    """GSWORLD_AMD_TUNING / TUNING["forward_only"] = 1 is an A/B aid for the frame renderers.  The rasterize_gaussians*
    paths keep their state for gsr_backward, which carves the FULL layout: they must stay training frames whatever is
    forced."""
    from gsworld_amd import _C, _lib

    saved = dict(_lib.TUNING)
    try:
        _lib.TUNING["forward_only"] = 1
        assert _C._tuning_list(None)[5] == 0          # training paths
        assert _C._tuning_list(False)[5] == 1         # a frame renderer's choice IS overridden (that is the A/B)
        st = _lib.GsrSettings(16, 16, 1.0, 1.0, 1.0, 3, 16, 0, 0, 0, 0.05)
        _lib.apply_tuning(st, allow_forward_only=False)
        assert st.forward_only == 0
        _lib.apply_tuning(st)
        assert st.forward_only == 1
        _lib.TUNING["forward_only"] = -1
        assert _C._tuning_list(True)[5] == 1 and _C._tuning_list(None)[5] == 0
    finally:
        _lib.TUNING.update(saved)


def test_frame_gather_logs_are_bounded():
    """FrameGather.waits only records with timing=True and never grows without bound."""
    import torch

    from gsworld_amd.distributed import FrameGather

    fg = FrameGather(4, 4, batch=2, device=torch.device("cpu"), world=1, buffers=1, timing=False)
    for i in range(10_000):
        fg.wait_reusable(i)
        fg.step_done(i)
    assert len(fg.waits) == 0 and fg.waits.maxlen == 4096


@pytest.mark.skipif(not os.path.isdir("/root/reference/gsworld"), reason="the reference tree is only present in the authoring container")
def test_xarm6_rollout_fixture_is_what_its_script_writes(tmp_path):
    """tests/golden/xarm6_rollout.npz against a fresh run of tools/make_xarm6_rollout.py on the reference's URDF and
    constants (authoring container only): the committed fixture is that script's output, array for array."""
    import importlib.util

    import numpy as np

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_xarm6_rollout", os.path.join(root, "tools", "make_xarm6_rollout.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = str(tmp_path / "rollout.npz")
    mod.main(out)
    a, b = np.load(out), np.load(os.path.join(root, "tests", "golden", "xarm6_rollout.npz"))
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k


def test_plan_query_is_the_librarys_own_decision_and_knows_the_sample_sorts_model_limit():
    """gsr_plan_query = csrc/api.hip make_plan on the host (no device): an inference frame at configs[1] takes a permuted
    model on the default path; beyond 8 388 608 Gaussians (more block counts than ss_prepare's LDS holds) the frame falls
    back to the LSD radix depth sort, which takes NO permuted model -- the closed loop and SceneLayout ask this instead of
    mirroring the rules (the mirror had forgotten the model limit, and every frame of such a loop failed)."""
    from gsworld_amd import _lib
    from gsworld_amd.closed_loop import ClosedLoopRenderer

    p = _lib.plan_query(640, 480, 1_468_850, permuted=True, tuned=False)
    assert p is not None and p["infer"] == 1 and p["super"] == 1 and p["lean"] == 1 and p["radix_depth"] == 0
    limit = 32768 * 256
    assert _lib.plan_query(640, 480, limit, permuted=True, tuned=False)["radix_depth"] == 0
    assert _lib.plan_query(640, 480, limit + 1, permuted=True, tuned=False) is None
    big = _lib.plan_query(640, 480, limit + 1, permuted=False, tuned=False)
    assert big is not None and big["radix_depth"] == 1 and big["infer"] == 0
    # tile grids the counting placement does not take keep the caller's order as well
    assert _lib.plan_query(8192, 4096, 1000, permuted=True, tuned=False) is None

    class Cam:
        image_width, image_height = 640, 480

    assert ClosedLoopRenderer._takes_permuted_model(Cam, 1_468_850)
    assert not ClosedLoopRenderer._takes_permuted_model(Cam, limit + 1)


def test_stage_host_values_writes_mirror_and_slot_or_nothing():
    """csrc_torch/ext.cpp stage_host_values (the closed loop's per-step host values in one call): every (offset, tensor) pair
    lands in the mirror, the mirror in the slot; a tensor that is not plain host float32, or does not fit, leaves both
    untouched and the caller takes its general path."""
    import torch

    from gsworld_amd import _C

    if _C._ext is None or not hasattr(_C._ext, "stage_host_values"):
        pytest.skip("compiled binding not built")
    mirror, slot = torch.arange(40, dtype=torch.float32), torch.zeros(40)
    a, b = torch.full((2, 4), 7.0), torch.full((3,), 9.0)
    assert _C._ext.stage_host_values(mirror, slot, [(4, a), (30, b)])
    want = torch.arange(40, dtype=torch.float32)
    want[4:12], want[30:33] = 7.0, 9.0
    assert torch.equal(mirror, want) and torch.equal(slot, want)
    before = mirror.clone()
    for bad in ([(4, a.double())], [(38, b)], [(-1, b)], [(0, a.t())], [(4, a), (0, torch.ones(3, requires_grad=True))]):
        assert not _C._ext.stage_host_values(mirror, slot, bad)
        assert torch.equal(mirror, before) and torch.equal(slot, before)
