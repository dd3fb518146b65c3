"""Pins the CPU oracle (oracle/gs_oracle.c) with analytic known-answer tests (KATs 1-8) and with an
independently written PyTorch-CPU restatement.  The reference holds no golden vectors for this path and its
CUDA source is not vendored ("parity unpinned"), so these are what the oracle stands on."""
import math

import numpy as np
import pytest
import torch

from gsworld_amd import scenes
from gsworld_amd.camera import view_params
from oracle import gs_oracle as go
from oracle import torch_cpu_render as tcr
from tests import helpers as hp

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199


def _identity_cam(W, H, tanfov=1.0):
    f = 2 * math.atan(tanfov)
    return view_params(np.eye(3), np.zeros(3), f, f, W, H)


def _one(xyz, scale=0.05, opacity=0.8, dc=(0.0, 0.0, 0.0), quat=(1, 0, 0, 0), rest=None):
    n = len(xyz)
    sh = np.zeros((n, 16, 3), np.float32)
    sh[:, 0, :] = np.asarray(dc, np.float32)
    if rest is not None:
        sh[:, 1:, :] = rest
    return dict(means3D=np.asarray(xyz, np.float32), shs=sh, opacities=np.full(n, opacity, np.float32),
                scales=np.full((n, 3), scale, np.float32), rotations=np.tile(np.asarray(quat, np.float32), (n, 1)))


def _fwd(inp, cam, bg=(0, 0, 0), **kw):
    st = hp.oracle_settings(cam, **kw)
    full = dict(inp, viewmatrix=cam.world_view_transform.numpy().reshape(-1),
                projmatrix=cam.full_proj_transform.numpy().reshape(-1), campos=cam.camera_center.numpy())
    return hp.oracle_forward(full, st, np.asarray(bg, np.float32), border_eps=0.0), st


def test_kat1_higher_msb():
    # rasterizer_impl.cu getHigherMsb; sort bits = 32 + msb (1200 -> 11, 256 -> 9, 2500 -> 12)
    for n, want in [(256, 9), (1200, 11), (2500, 12), (1, 1), (2, 2), (3, 2), (4, 3), (65535, 16), (65536, 17)]:
        assert go.higher_msb(n) == want, n


def test_kat2_sh_degree0_and_degree1_signs():
    cam = _identity_cam(64, 64)
    dc = (0.3, -0.2, 1.1)
    o, _ = _fwd(_one([[0, 0, 2.0]], dc=dc), cam, sh_degree=0)
    want = np.maximum(0, np.float32(SH_C0) * np.asarray(dc, np.float32) + np.float32(0.5))
    np.testing.assert_allclose(o["geom"]["rgb"][0], want, rtol=0, atol=1e-7)
    # degree 1: direction (x,y,z) normalised; colour = C0 sh0 - C1 y sh1 + C1 z sh2 - C1 x sh3 + 0.5
    rest = np.zeros((1, 15, 3), np.float32)
    rest[0, 0] = (1.0, 0, 0)   # sh1 -> -y
    rest[0, 1] = (0, 1.0, 0)   # sh2 -> +z
    rest[0, 2] = (0, 0, 1.0)   # sh3 -> -x
    p = np.array([0.3, -0.4, 2.0])
    o, _ = _fwd(_one([p], dc=(0, 0, 0), rest=rest), cam, sh_degree=1)
    d = p / np.linalg.norm(p)
    want = np.maximum(0, np.array([-SH_C1 * d[1], SH_C1 * d[2], -SH_C1 * d[0]]) + 0.5)
    np.testing.assert_allclose(o["geom"]["rgb"][0], want, atol=2e-7)
    # clamp flags: a strongly negative dc clamps to 0 and raises the flag
    o, _ = _fwd(_one([[0, 0, 2.0]], dc=(-5, 0, 5)), cam, sh_degree=0)
    assert o["geom"]["rgb"][0, 0] == 0 and list(o["geom"]["clamped"][0]) == [1, 0, 0]


def test_kat3_near_cull_is_gsworlds():
    """/root/reference/README.md:33: cull at p_view.z <= 0.05f, not stock 0.2f."""
    cam = _identity_cam(64, 64)
    z_above = np.nextafter(np.float32(0.05), np.float32(1))
    inp = _one([[0, 0, 0.05], [0, 0, z_above], [0, 0, 0.19], [0, 0, -1.0]], scale=0.001)
    o, _ = _fwd(inp, cam)
    assert list(o["geom"]["radii"] > 0) == [False, True, True, False]
    o, _ = _fwd(inp, cam, near_plane=0.2)
    assert list(o["geom"]["radii"] > 0) == [False, False, False, False]
    assert list(go.mark_visible(inp["means3D"], cam.world_view_transform.numpy().reshape(-1), 0.05)) == \
        [False, True, True, False]


def test_kat4_single_isotropic_gaussian_on_axis():
    W = H = 64
    cam = _identity_cam(W, H, 1.0)
    s, z, op = 0.1, 2.0, 0.8
    dc = (1.0, 0.5, -0.25)
    o, _ = _fwd(_one([[0, 0, z]], scale=s, opacity=op, dc=dc), cam, bg=(0.1, 0.2, 0.3), sh_degree=0)
    g = o["geom"]
    fx = W / 2.0
    var = (fx * s / z) ** 2 + 0.3
    np.testing.assert_allclose(g["means2D"][0], [31.5, 31.5], atol=1e-5)  # ((0+1)*64-1)/2
    np.testing.assert_allclose(g["conic_opacity"][0], [1 / var, 0, 1 / var, op], rtol=1e-6, atol=1e-9)
    assert g["radii"][0] == math.ceil(3 * math.sqrt(var))
    rgb = np.maximum(0, SH_C0 * np.asarray(dc) + 0.5)
    alpha = min(0.99, op * math.exp(-0.5 * (0.25 + 0.25) / var))  # pixel (32,32): d = (-0.5,-0.5)
    want = rgb * alpha + (1 - alpha) * np.array([0.1, 0.2, 0.3])
    np.testing.assert_allclose(o["color"][:, 32, 32], want, rtol=1e-5)
    np.testing.assert_allclose(o["invdepth"][0, 32, 32], alpha / z, rtol=1e-5)
    assert o["n_contrib"][32, 32] == 1 and abs(o["final_T"][32, 32] - (1 - alpha)) < 1e-6
    # far corner: untouched -> background, T = 1
    np.testing.assert_allclose(o["color"][:, 0, 0], [0.1, 0.2, 0.3], atol=1e-7)


def test_kat5_radius_rect_and_border_clamp():
    W, H = 64, 48
    cam = _identity_cam(W, H, 1.0)
    # centre of the image: pix = (31.5, 23.5); scale chosen so that radius is known
    o, st = _fwd(_one([[0, 0, 2.0]], scale=0.02), cam)
    g = o["geom"]
    fx, fy = W / 2.0, H / 2.0
    vx, vy = (fx * 0.02 / 2.0) ** 2 + 0.3, (fy * 0.02 / 2.0) ** 2 + 0.3
    mid, det = 0.5 * (vx + vy), vx * vy
    lam = mid + math.sqrt(max(0.1, mid * mid - det))
    r = math.ceil(3 * math.sqrt(lam))
    assert g["radii"][0] == r
    px, py = 31.5, 23.5
    want = [int((px - r) / 16), int((py - r) / 16), int((px + r + 15) / 16), int((py + r + 15) / 16)]
    assert list(g["rects"][0]) == want
    assert g["tiles_touched"][0] == (want[2] - want[0]) * (want[3] - want[1])
    # splat far outside to the right: rect clamps to an empty range -> dropped
    o, _ = _fwd(_one([[50.0, 0, 2.0]], scale=0.02), cam)
    assert o["geom"]["radii"][0] == 0 and o["geom"]["tiles_touched"][0] == 0
    # splat overlapping the left border: min clamps to 0
    o, _ = _fwd(_one([[-1.99, 0, 2.0]], scale=0.05), cam)
    assert o["geom"]["radii"][0] > 0 and o["geom"]["rects"][0][0] == 0


def test_kat6_keys_stable_ties_and_order_invariance():
    cam = _identity_cam(64, 64)
    xyz = [[0.1, 0.0, 3.0], [0.1, 0.0, 2.0], [0.1, 0.0, 2.0], [-0.1, 0.0, 2.5]]
    o, st = _fwd(_one(xyz, scale=0.05, dc=(1, 1, 1)), cam)
    b, g = o["binning"], o["geom"]
    gx = 4
    keys = b["keys"]
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    dbits = (keys & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    assert np.all(np.diff(tiles) >= 0)
    for t in np.unique(tiles):
        sel = tiles == t
        d, idx = dbits[sel], b["point_list"][sel]
        assert np.all(np.diff(d.astype(np.int64)) >= 0)  # ascending depth bits (positive floats are monotone)
        for k in range(1, len(d)):
            if d[k] == d[k - 1]:
                assert idx[k] > idx[k - 1]  # equal depth -> ascending Gaussian index
        np.testing.assert_array_equal(d, g["depths"][idx].view(np.uint32))
        r = b["ranges"][t]
        assert r[1] - r[0] == sel.sum() and np.all(tiles[r[0]:r[1]] == t)
    assert 0 <= tiles.min() and tiles.max() < gx * gx
    # emission order is (Gaussian index, y, x)
    ku = b["keys_unsorted"]
    first = int(g["tiles_touched"][0])
    t0 = (ku[:first] >> np.uint64(32)).astype(np.int64)
    r = g["rects"][0]
    want = [y * gx + x for y in range(r[1], r[3]) for x in range(r[0], r[2])]
    assert list(t0) == want
    # distinct depths: the image does not depend on the input order
    rng = np.random.default_rng(0)
    raw = scenes.random_scene_camera_frame(500, seed=12)
    cam2 = scenes.identity_camera(64, 64, 60.0)
    inp = hp.np_inputs(raw, cam2)
    st2 = hp.oracle_settings(cam2)
    o1 = hp.oracle_forward(inp, st2, np.zeros(3, np.float32))
    perm = rng.permutation(500)
    inp2 = dict(inp, **{k: inp[k][perm] for k in ("means3D", "shs", "opacities", "scales", "rotations")})
    o2 = hp.oracle_forward(inp2, st2, np.zeros(3, np.float32))
    np.testing.assert_array_equal(o1["color"], o2["color"])


def test_kat7_compositing_thresholds():
    W = H = 16
    cam = _identity_cam(W, H, 1.0)
    # (a) alpha cap 0.99 and termination excludes the terminating instance: 3 opaque splats stacked on the axis
    xyz = [[0, 0, 1.0 + 0.25 * k] for k in range(7)]
    o, _ = _fwd(_one(xyz, scale=50.0, opacity=0.8, dc=(1, 1, 1)), cam, bg=(1, 0, 0), sh_degree=0)
    # huge splats: exp(power) ~ 1, alpha = 0.8 each.  T: 1, .2, .04, .008, .0016, .00032; the 6th would give
    # 6.4e-5 < 1e-4 -> it terminates the pixel and is NOT blended; the 7th is never examined.
    T = o["final_T"][8, 8]
    assert abs(T - 0.2 ** 5) < 1e-7 and o["n_contrib"][8, 8] == 5
    c = SH_C0 + 0.5
    want_r = c * 0.8 * sum(0.2 ** k for k in range(5)) + T * 1.0
    np.testing.assert_allclose(o["color"][0, 8, 8], want_r, rtol=1e-5)
    # the 0.99 cap: opacity 1.0 and exp(power) ~ 1 gives alpha = 0.99, T = 1 - 0.99 after one splat
    o, _ = _fwd(_one([[0, 0, 1.0]], scale=50.0, opacity=1.0, dc=(1, 1, 1)), cam, sh_degree=0)
    assert abs(o["final_T"][8, 8] - (np.float32(1) - np.float32(0.99))) < 1e-9
    # (b) alpha < 1/255 is skipped entirely: opacity below the threshold leaves the background
    o, _ = _fwd(_one([[0, 0, 1.0]], scale=5.0, opacity=0.0039, dc=(1, 1, 1)), cam, bg=(0.5, 0.5, 0.5))
    assert o["n_contrib"][8, 8] == 0 and o["final_T"][8, 8] == 1.0
    np.testing.assert_array_equal(o["color"][:, 8, 8], np.float32([0.5, 0.5, 0.5]))
    o, _ = _fwd(_one([[0, 0, 1.0]], scale=5.0, opacity=0.004, dc=(1, 1, 1)), cam, bg=(0.5, 0.5, 0.5))
    assert o["n_contrib"][8, 8] == 1
    # (c) n_contrib counts examined instances: a skipped splat in front still advances the counter
    o, _ = _fwd(dict(_one([[0, 0, 1.0], [0, 0, 2.0]], scale=5.0, dc=(1, 1, 1)),
                     opacities=np.float32([0.001, 0.5])), cam)
    assert o["n_contrib"][8, 8] == 2


def test_kat8_tiled_float32_matches_bruteforce_float64():
    raw = scenes.random_scene_camera_frame(3000, seed=13)
    cam = scenes.identity_camera(80, 48, 60.0)
    inp = hp.np_inputs(raw, cam)
    st = hp.oracle_settings(cam)
    bg = np.float32([0.1, 0.3, 0.7])
    o = hp.oracle_forward(inp, st, bg, border_eps=1e-5, border_eps_T=1e-4)
    c64, d64 = go.render_bruteforce(st, o["geom"], bg)
    ok = o["borderline"] == 0
    assert np.abs(o["color"] - c64)[:, ok].max() <= 1e-5
    assert np.abs(o["invdepth"] - d64)[:, ok].max() <= 1e-5 * max(1.0, np.abs(d64).max())


@pytest.mark.parametrize("n,w,h,seed", [(20_000, 96, 64, 21), (100_000, 256, 256, 0)])
def test_c_oracle_agrees_with_torch_cpu_restatement(n, w, h, seed):
    """BASELINE.json configs[0] gate: two independent CPU restatements agree (image <= 1e-5, integers exact)."""
    raw = scenes.random_scene_camera_frame(n, seed=seed)
    cam = scenes.identity_camera(w, h, 60.0)
    inp = hp.np_inputs(raw, cam)
    st = hp.oracle_settings(cam)
    bg = np.float32([0.2, 0.1, 0.4])
    o = hp.oracle_forward(inp, st, bg, border_eps=2e-5, border_eps_T=2e-4)
    means, shs, op, sc, rot = raw.activated()
    t = tcr.render(means, shs, op, sc, rot, cam.world_view_transform, cam.full_proj_transform, cam.camera_center,
                   torch.from_numpy(bg), h, w, cam.tanfovx, cam.tanfovy)
    geom = o["geom"]
    # integer stages: the two restatements round differently (explicit FMA vs plain ops), so a radius / rect can
    # differ only where the float it was cut from sits on an integer boundary; require near-total agreement
    same_r = (t["radii"].numpy() == geom["radii"])
    assert same_r.mean() > 0.9995, same_r.mean()
    agree = same_r & (geom["radii"] > 0)
    np.testing.assert_allclose(t["means2D"].numpy()[agree], geom["means2D"][agree], atol=2e-3)
    np.testing.assert_allclose(t["rgb"].numpy()[agree], geom["rgb"][agree], atol=2e-6)
    assert abs(t["num_rendered"] - o["binning"]["num_rendered"]) <= 0.001 * o["binning"]["num_rendered"] + 8
    if same_r.all() and t["num_rendered"] == o["binning"]["num_rendered"]:
        assert (t["point_list"].numpy() == o["binning"]["point_list"]).mean() > 0.999
    ok = o["borderline"] == 0
    diff = np.abs(t["color"].numpy() - o["color"])
    # pixels whose tile list differs (a flipped radius) are excluded with the borderline ones
    frac_bad = (diff.max(0) > 1e-5)[ok].mean()
    assert frac_bad < 2e-3, frac_bad
    assert np.median(diff) < 1e-6


def test_canonical_activations_of_the_raw_parameter_path():
    """gso_expf / gso_activate_params (the float32 order the HIP raw-parameter path reproduces):
    within ~1 ulp of the true exponential, and what upstream's getters compute (sigmoid / exp / F.normalize)."""
    import torch

    from oracle import gs_oracle as go

    x = np.concatenate((np.linspace(-30, 20, 4001), [-104.5, -103.0, -87.5, 0.0, 1e-8, 88.7, 88.73, 100.0]))
    x = x.astype(np.float32)
    e = go.expf(x).astype(np.float64)
    ref = np.exp(x.astype(np.float64))
    finite = (ref < 3.4e38) & (ref > 1.2e-38)
    assert (np.abs(e - ref)[finite] / ref[finite]).max() <= 1.3e-7
    assert np.isinf(e[x > 88.73]).all() and (e[x < -104] == 0).all() and go.expf([0.0])[0] == 1.0
    rng = np.random.default_rng(0)
    logit = rng.uniform(-8, 8, 5000).astype(np.float32)
    logs = rng.uniform(-9, 2, (5000, 3)).astype(np.float32)
    quat = (rng.standard_normal((5000, 4)) * 1.7).astype(np.float32)
    quat[0] = 0.0  # F.normalize clamps the norm at 1e-12: the zero quaternion stays zero
    op, sc, ro = go.activate_params(logit, logs, quat, flags=7)
    assert np.abs(op - torch.sigmoid(torch.from_numpy(logit)).numpy()).max() <= 2e-7
    t_sc = torch.exp(torch.from_numpy(logs)).numpy()
    assert (np.abs(sc - t_sc) / t_sc).max() <= 3e-7
    assert np.abs(ro - torch.nn.functional.normalize(torch.from_numpy(quat)).numpy()).max() <= 2e-7
    assert (ro[0] == 0).all()
    op2, sc2, ro2 = go.activate_params(logit, logs, quat, flags=0)  # no flag: copied through
    assert (op2 == logit).all() and (sc2 == logs).all() and (ro2 == quat).all()
