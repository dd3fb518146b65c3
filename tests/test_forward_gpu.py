"""GPU parity of the forward rasterizer: HIP path (through the C ABI) vs the CPU oracle, stage by stage.

Bar (north_star): tile / key indices bit-exact, RGB and inverse depth <= 1e-4 max abs.  The oracle is the
project's own restatement (parity vs the CUDA reference is unpinned -- see DESIGN.md)."""
import numpy as np
import pytest
import torch

from tests import helpers as hp
from gsworld_amd import debug as dbg, scenes

pytestmark = pytest.mark.gpu


def _run(raw, cam, bg=(0.0, 0.0, 0.0), all_pixel_tol=0.02, **kw):
    inp = hp.np_inputs(raw, cam)
    st = hp.oracle_settings(cam, **kw)
    bg = np.asarray(bg, np.float32)
    o = hp.oracle_forward(inp, st, bg)
    g = hp.gpu_forward(inp, st, bg)
    return hp.compare_forward(o, g, st, all_pixel_tol=all_pixel_tol)


def test_config1_100k_256(cuda_device):
    """BASELINE.json configs[0]: 100k random Gaussians, one 256x256 camera (numerics gate)."""
    rep = _run(scenes.random_scene_camera_frame(100_000, seed=0), scenes.identity_camera(256, 256, 60.0),
               all_pixel_tol=hp.RGB_TOL)  # north_star's 1e-4 on EVERY pixel, borderline decisions included
    assert rep["V"] > 50_000 and rep["R"] > rep["V"]


@pytest.mark.parametrize("w,h", [(70, 50), (16, 16), (33, 17), (640, 480)])
def test_ragged_image_sizes(cuda_device, w, h):
    _run(scenes.random_scene_camera_frame(20_000, seed=3), scenes.identity_camera(w, h, 70.0), bg=(0.2, 0.5, 0.9))


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_degrees(cuda_device, deg):
    _run(scenes.random_scene_camera_frame(5_000, seed=4), scenes.identity_camera(128, 96, 60.0), sh_degree=deg)


def test_antialiasing_and_scale_modifier(cuda_device):
    _run(scenes.random_scene_camera_frame(20_000, seed=5), scenes.identity_camera(200, 120, 60.0), antialiasing=True,
         scale_modifier=0.7)


def test_stock_near_plane(cuda_device):
    """0.2f (stock upstream) vs GSWorld's 0.05f cull differ exactly on the near band."""
    raw = scenes.random_scene_camera_frame(20_000, seed=6, near_fraction=0.2)
    cam = scenes.identity_camera(128, 128, 60.0)
    r_gs = _run(raw, cam, near_plane=0.05)
    r_stock = _run(raw, cam, near_plane=0.2)
    assert r_gs["V"] > r_stock["V"]


def test_precomputed_colors_and_cov3d(cuda_device):
    raw = scenes.random_scene_camera_frame(10_000, seed=7)
    cam = scenes.identity_camera(160, 160, 60.0)
    inp = hp.np_inputs(raw, cam)
    st = hp.oracle_settings(cam)
    bg = np.zeros(3, np.float32)
    o0 = hp.oracle_forward(inp, st, bg)
    colors = np.random.default_rng(0).random((10_000, 3), dtype=np.float32)
    cov = o0["geom"]["cov3D"].copy()
    # Gaussians culled by the oracle have no cov3D: give them a small isotropic one
    cov[o0["geom"]["radii"] == 0] = np.array([1e-4, 0, 0, 1e-4, 0, 1e-4], np.float32)
    o = hp.oracle_forward(inp, st, bg, colors_precomp=colors, cov3D_precomp=cov)
    g = hp.gpu_forward(inp, st, bg, colors_precomp=colors, cov3D_precomp=cov)
    hp.compare_forward(o, g, st)


def test_depth_ties_break_by_index(cuda_device):
    """Duplicated Gaussians (equal depth bits) must keep ascending index order inside every tile."""
    raw = scenes.random_scene_camera_frame(3_000, seed=8)
    dup = scenes.RawGaussians(*[torch.cat((t, t, t)) for t in (
        raw.xyz, raw.features_dc, raw.features_rest, raw.opacity, raw.scaling, raw.rotation)])
    _run(dup, scenes.identity_camera(96, 96, 60.0))


def test_empty_and_single(cuda_device):
    from gsworld_amd import _C

    dev = cuda_device
    e = torch.empty(0, device=dev)
    bg = torch.tensor([0.3, 0.2, 0.1], device=dev)
    eye = torch.eye(4, device=dev)
    R, color, radii, gb, bb, ib, invd = _C.rasterize_gaussians(
        bg, torch.empty(0, 3, device=dev), e, torch.empty(0, 1, device=dev), torch.empty(0, 3, device=dev),
        torch.empty(0, 4, device=dev), 1.0, e, eye, eye, 1.0, 1.0, 32, 48, torch.empty(0, 16, 3, device=dev), 3,
        torch.zeros(3, device=dev), False, False, False)
    assert R == 0 and color.shape == (3, 32, 48) and float(color.abs().max()) == 0.0 and radii.numel() == 0
    # everything culled (behind the camera): image = background, R = 0
    raw = scenes.random_scene_camera_frame(100, seed=9)
    raw.xyz[:, 2] = -1.0
    rep = _run(raw, scenes.identity_camera(48, 32, 60.0), bg=(0.3, 0.2, 0.1))
    assert rep["V"] == 0 and rep["R"] == 0
    rep = _run(scenes.random_scene_camera_frame(1, seed=10, near_fraction=0.0), scenes.identity_camera(64, 64, 60.0))
    assert rep["P"] == 1


def test_tabletop_config2_full(cuda_device):
    """BASELINE.json configs[1] at full size: 1,468,850 Gaussians, 640x480 sensor camera."""
    rep = _run(scenes.tabletop_scene("xarm6_align"), scenes.sensor_camera("xarm6_align"), all_pixel_tol=hp.RGB_TOL)
    assert rep["P"] == scenes.XARM6_ALIGN_NUM_GAUSSIANS


# All-pixel bound of the full-size scenes.  A pixel whose alpha >= 1/255 (or T >= 1e-4) decision sits within an exp() ulp of
# its threshold may gain or lose one contribution of at most T / 255 = 3.9e-3 (v_exp_f32(power * log2 e) against libm,
# DESIGN.md section 2).  Measured at full size (round 5, bench.py `parity.per_scene_worst`): six scenes <= 3.0e-5 on every
# pixel, xarm6_rot_banana 3.9e-4 and fr3_pour 6.1e-4 on one flipped pixel each; off the pixels the oracle flags as
# borderline every scene is within 4.2e-7.  north_star's 1e-4 therefore holds on every pixel that has no decision inside
# the exp() band, and the two BASELINE headline configurations (test_config1_100k_256, test_tabletop_config2_full) hold
# it on ALL pixels.
ALL_PIXEL_TOL_CONFIG4 = 1e-3


@pytest.mark.parametrize("name", scenes.SCENE_NAMES)
def test_all_eight_scenes_of_config4(cuda_device, name, monkeypatch):
    """BASELINE.json configs[3]: every scene of /root/reference/configs/*.json (xarm6_* use sim2gs_xarm_trans and the
    xarm camera, fr3_* sim2gs_arm_trans and right2base) at FULL size (1,468,850 Gaussians) against the oracle -- every
    index bit-exact, every preprocess float bit-exact, RGB / inverse depth <= 1e-4 off the borderline pixels, every pixel
    within ALL_PIXEL_TOL_CONFIG4 -- plus size-independent properties: stable under a permutation of the Gaussians, every
    tile list sorted by (depth bits, index), ranges partition [0, R)."""
    seed = 1 + scenes.SCENE_NAMES.index(name)  # gsworld_amd.distributed.scene_for_rank
    cam = scenes.sensor_camera(name)
    raw = scenes.tabletop_scene(name, seed=seed)
    inp, st, bg = hp.np_inputs(raw, cam), hp.oracle_settings(cam), np.zeros(3, np.float32)
    o = hp.oracle_forward(inp, st, bg)
    rep = hp.compare_forward(o, hp.gpu_forward(inp, st, bg), st, all_pixel_tol=ALL_PIXEL_TOL_CONFIG4)
    print(f"config4 {name}: P {rep['P']} V {rep['V']} R {rep['R']} worst pixel off borderline {rep['rgb_max_abs']:.3e}, "
          f"all pixels {rep['rgb_max_abs_all']:.3e}, borderline pixels {rep['borderline_pixels']}")
    assert rep["P"] == scenes.XARM6_ALIGN_NUM_GAUSSIANS and rep["V"] > 50_000 and rep["R"] > rep["V"]
    _full_size_properties(raw, cam)
    # ... and north_star's 1e-4 on EVERY pixel, borderline decisions included, with the library's exp-accurate build
    # (libgsr_hip_expacc.so: the same sources with -DGSR_EXP_ACCURATE=1, csrc/Makefile -- the product power x log2(e) keeps its
    # rounding error out of v_exp_f32's argument).  The shipped default differs from the oracle's libm
    # exp by one flipped threshold decision on two of the eight scenes (3.9e-4, 6.1e-4), which is a difference between this
    # project's own two implementations of exp(), not one against the reference.
    import os

    from gsworld_amd import _C, _lib

    path = os.path.join(os.path.dirname(_lib.LIB_PATH), "libgsr_hip_expacc.so")
    if not os.path.exists(path):
        pytest.skip("libgsr_hip_expacc.so not built (make -C gsworld_amd/csrc)")
    monkeypatch.setattr(_C, "_ext", None)  # (the compiled binding is linked against the shipped library)
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", path)
    assert b"gfx950" in _lib.lib().gsr_version()
    rep2 = hp.compare_forward(o, hp.gpu_forward(inp, st, bg), st, all_pixel_tol=hp.RGB_TOL)
    print(f"config4 {name}, exp-accurate build: all pixels {rep2['rgb_max_abs_all']:.3e}")
    assert rep2["rgb_max_abs_all"] <= hp.RGB_TOL


def _full_size_properties(raw, cam, device="cuda"):
    inp, st = hp.np_inputs(raw, cam), hp.oracle_settings(cam)
    bg = np.zeros(3, np.float32)
    g = hp.gpu_forward(inp, st, bg, device=device)
    v = g["views"]
    R, V = g["num_rendered"], g["num_visible"]
    assert V == int((g["radii"] > 0).sum()) and R == int(v["tiles_touched"].astype(np.int64).sum())
    ranges = v["ranges"].astype(np.int64)
    nonempty = ranges[:, 1] > ranges[:, 0]
    # ranges partition [0, R) in tile order
    starts, ends = ranges[nonempty, 0], ranges[nonempty, 1]
    assert starts[0] == 0 and ends[-1] == R and np.array_equal(starts[1:], ends[:-1])
    # inside every tile: ascending (depth bits, Gaussian index) -- the reference's stable 64-bit key order
    pl = v["point_list"].astype(np.int64)
    key = (v["point_tiles"].astype(np.uint64) << np.uint64(52)) | \
        (hp._bits(v["depths"][pl]).astype(np.uint64) << np.uint64(21)) | pl.astype(np.uint64)
    assert pl.max() < (1 << 21) and np.all(key[1:] > key[:-1]), "point list is not sorted by (tile, depth bits, index)"
    # every instance lies inside its Gaussian's tile rect
    gx = (cam.image_width + 15) // 16
    ty, tx = np.divmod(v["point_tiles"].astype(np.int64), gx)
    rc = v["rects"][pl].astype(np.int64)
    assert np.all((tx >= rc[:, 0]) & (tx < rc[:, 2]) & (ty >= rc[:, 1]) & (ty < rc[:, 3]))
    # permuting the Gaussians changes index tie-breaks only: same image within float noise, same R and V
    perm = np.random.default_rng(0).permutation(inp["means3D"].shape[0])
    inp2 = dict(inp, **{k: inp[k][perm] for k in ("means3D", "shs", "opacities", "scales", "rotations")})
    g2 = hp.gpu_forward(inp2, st, bg, device=device)
    assert g2["num_rendered"] == R and g2["num_visible"] == V
    assert np.array_equal(g2["radii"], g["radii"][perm])
    d = np.abs(g2["color"] - g["color"])  # (equal depth bits of two overlapping splats swap their blend order there)
    assert float((d > 1e-5).mean()) <= 0.02 and float(d.mean()) <= 1e-5 and float(d.max()) <= 0.1
    # rendering again is bit-identical (no order-dependent atomics on the forward path)
    g3 = hp.gpu_forward(inp, st, bg, device=device)
    assert np.array_equal(g3["color"].view(np.uint32), g["color"].view(np.uint32))
    assert np.array_equal(g3["views"]["point_list"], v["point_list"])


@pytest.mark.parametrize("mode", [0, 2, 3, 4])
def test_alternative_binning_paths_match_too(cuda_device, mode):
    """Mode 0 (depth sort + emit + tile radix sort; also what tile grids above 16384 tiles use) and mode 2 (unordered
    binning + per-tile LDS sort) must give the same point list as the default mode 1 (depth sort + counting)."""
    raw = scenes.random_scene_camera_frame(30_000, seed=14)
    dbg.set_binning_mode(mode)
    try:
        _run(raw, scenes.identity_camera(200, 120, 60.0))   # 104 tiles: 1 radix pass (result side 1 -> copied)
        _run(raw, scenes.identity_camera(640, 480, 60.0))   # 1200 tiles: 2 passes
    finally:
        dbg.set_binning_mode(1)


@pytest.mark.parametrize("case", ["lsd_radix_variant", "all_equal_depth", "two_depths", "one_million_visible", "tiny",
                                  "index_coherent", "quantised_depths", "a_few_equal_depths"])
def test_depth_sort_paths(cuda_device, case):
    """The depth order (ascending depth bits, ties by index) from every path of the depth sort: the default sample sort
    with buckets of every size class (LDS radix; a bucket too large for the LDS -> global-memory bitonic fallback;
    B = 2048 buckets for a million visible Gaussians) and round 1's 3-pass LSD radix sort (GsrSettings.depth_sort = 1).
    compare_forward checks `depth_order`, the point list and the reconstructed 64-bit keys bit for bit."""
    if case == "lsd_radix_variant":
        dbg.set_depth_sort(1)
        try:
            _run(scenes.random_scene_camera_frame(30_000, seed=14), scenes.identity_camera(200, 120, 60.0))
            _run(scenes.tabletop_scene("xarm6_align", n=300_000, seed=5), scenes.sensor_camera("xarm6_align"))
        finally:
            dbg.set_depth_sort(0)
        return
    if case == "index_coherent":
        # a model stored in spatial order (as real scans often are): every visible Gaussian sits in a few thousand
        # consecutive preprocess blocks -- the sort's work shares follow the VISIBLE counts, not the block index
        raw = scenes.tabletop_scene("xarm6_align", n=400_000, seed=6)
        perm = torch.argsort(raw.xyz[:, 0] * 3.0 + raw.xyz[:, 2])
        raw = scenes.RawGaussians(*[None if t is None else t[perm].contiguous() for t in (
            raw.xyz, raw.features_dc, raw.features_rest, raw.opacity, raw.scaling, raw.rotation, raw.semantics)])
        rep = _run(raw, scenes.sensor_camera("xarm6_align"))
        assert rep["V"] > 20_000
        return
    if case == "one_million_visible":
        raw = scenes.random_scene_camera_frame(1_100_000, seed=31, near_fraction=0.0)
        raw.scaling -= 2.0  # small splats: the oracle's compositing stays cheap
        rep = _run(raw, scenes.identity_camera(256, 256, 75.0))
        assert rep["V"] > 900_000
        return
    if case in ("quantised_depths", "a_few_equal_depths"):
        # equal depth bits inside ordinary buckets (two Gaussians with the same depth bits are common: ~1 300 pairs per
        # config-2 frame): the reference's key sort leaves them by ascending index
        raw = scenes.random_scene_camera_frame(120_000, seed=36, near_fraction=0.0)
        raw.scaling -= 1.5
        if case == "quantised_depths":
            raw.xyz[:, 2] = torch.round(raw.xyz[:, 2] * 200.0) / 200.0  # ~600 distinct depths, ~200 Gaussians each
        else:
            src = torch.randperm(raw.num, generator=torch.Generator().manual_seed(1))[:400]
            raw.xyz[src[:200], 2] = raw.xyz[src[200:], 2]
        rep = _run(raw, scenes.identity_camera(192, 192, 60.0))
        assert rep["V"] > 100_000
        return
    n = {"all_equal_depth": 20_000, "two_depths": 9_000, "tiny": 3}[case]
    raw = scenes.random_scene_camera_frame(n, seed=32, near_fraction=0.0)
    raw.scaling -= 1.0
    if case == "all_equal_depth":
        raw.xyz[:, 2] = 2.0  # identity view: depth = z, 20 k equal keys -> ONE bucket larger than the LDS
    elif case == "two_depths":
        raw.xyz[: n // 2, 2] = 3.0
        raw.xyz[n // 2:, 2] = 2.0
    rep = _run(raw, scenes.identity_camera(160, 160, 60.0))
    assert rep["V"] > 0


@pytest.mark.parametrize("case", ["tabletop", "huge_splats_800", "wide_strip", "tiny", "lsd_radix_order"])
def test_chunk_placement(cuda_device, case):
    """chunkplace.hip (GsrSettings.binning_path = 4): point list, ranges and keys bit for bit -- a table-top frame (340
    chunks x 5 row groups), splats that cover most of an 800 x 800 image (thousands of spans per chunk: the staged span
    list is refilled several times per unit), a 4000-px-wide strip (250 tile columns, one row per group), three
    Gaussians, and the depth order coming from the LSD radix variant of the sort."""
    dbg.set_binning_mode(4)
    try:
        if case == "tabletop":
            rep = _run(scenes.tabletop_scene("xarm6_align", n=400_000, seed=8), scenes.sensor_camera("xarm6_align"))
            assert rep["R"] > 500_000
        elif case == "huge_splats_800":
            raw = scenes.random_scene_camera_frame(6_000, seed=33, near_fraction=0.0)
            raw.scaling += 2.2  # ~9x larger: a splat covers hundreds of tiles
            raw.opacity -= 3.0
            rep = _run(raw, scenes.identity_camera(800, 800, 60.0))
            assert rep["R"] > 150 * rep["V"]
        elif case == "wide_strip":
            rep = _run(scenes.random_scene_camera_frame(40_000, seed=34), scenes.identity_camera(4000, 48, 120.0))
            assert rep["R"] > rep["V"] > 1000
        elif case == "tiny":
            _run(scenes.random_scene_camera_frame(3, seed=35, near_fraction=0.0), scenes.identity_camera(64, 64, 60.0))
        else:
            dbg.set_depth_sort(1)
            try:
                _run(scenes.tabletop_scene("xarm6_align", n=200_000, seed=9), scenes.sensor_camera("xarm6_align"))
            finally:
                dbg.set_depth_sort(0)
    finally:
        dbg.set_binning_mode(1)


def test_cull_and_rect_on_stress_inputs(cuda_device):
    """Cull / radius / rect on stress inputs: most of the cloud off screen with a band of it across every image edge,
    huge and tiny scales, splats right at the near plane.  Every output of the frame must equal the oracle's."""
    for seed, near_fraction, dscale in ((40, 0.3, 0.0), (41, 0.0, 2.5), (42, 0.05, -3.0)):
        raw = scenes.random_scene_camera_frame(40_000, seed=seed, near_fraction=near_fraction)
        raw.scaling += dscale
        raw.xyz[:, :2] *= 3.0  # most of the cloud off screen, a band of it across every edge
        _run(raw, scenes.identity_camera(208, 144, 70.0))


@pytest.mark.parametrize("scene", ["random", "tabletop", "thin"])
def test_compositing_variants_are_bit_identical(cuda_device, scene):
    """The default compositing kernel culls, per 8x8 quadrant, instances that cannot reach alpha >= 1/255 there.
    That must not change a single bit of the image state: compare with the plain tile kernel (variant 0) and the
    unculled queue kernel (variant 2)."""
    if scene == "random":
        raw, cam = scenes.random_scene_camera_frame(60_000, seed=21), scenes.identity_camera(333, 201, 60.0)
    elif scene == "tabletop":
        raw, cam = scenes.tabletop_scene("xarm6_align", n=300_000, seed=5), scenes.sensor_camera("xarm6_align")
    else:  # needle-like splats with low opacity: the cull bound is tight and the conic nearly singular
        raw, cam = scenes.random_scene_camera_frame(40_000, seed=22), scenes.identity_camera(256, 256, 60.0)
        raw.scaling[:, 0] += 2.5
        raw.scaling[:, 1:] -= 3.0
        raw.opacity -= 2.0
    inp, st = hp.np_inputs(raw, cam), hp.oracle_settings(cam)
    bg = np.asarray((0.1, 0.2, 0.3), np.float32)
    outs = {}
    try:
        for variant in (0, 2, 3, 4):
            dbg.set_render_variant(variant, 0)
            g = hp.gpu_forward(inp, st, bg)
            outs[variant] = (g["color"], g["invdepth"], g["views"]["final_T"], g["views"]["n_contrib"])
    finally:
        dbg.set_render_variant(4, 0)
    assert outs[0][3].max() > 0
    for variant in (2, 3, 4):
        for name, a, b in zip(("color", "invdepth", "final_T", "n_contrib"), outs[0], outs[variant]):
            np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32), err_msg=f"variant {variant}: {name}")


def test_large_tile_grid_and_long_tile_lists(cuda_device):
    # 120 x 68 = 8160 tiles: counting placement in seven bands of tile rows
    rep = _run(scenes.random_scene_camera_frame(30_000, seed=15), scenes.identity_camera(1920, 1080, 60.0))
    assert rep["R"] > 0
    # a grid above the counting limit (132 x 132 = 17424 tiles) -> radix fallback
    rep = _run(scenes.random_scene_camera_frame(20_000, seed=17), scenes.identity_camera(2100, 2100, 60.0))
    assert rep["R"] > 0
    # a strip wider than 2048 tiles (one row of counters would not fit the LDS) -> radix fallback as well
    rep = _run(scenes.random_scene_camera_frame(20_000, seed=18), scenes.identity_camera(33_000, 16, 60.0))
    assert rep["R"] > 0
    # tile lists longer than the 8192-key LDS sort: 30k big splats on a 64x64 image (16 tiles)
    raw = scenes.random_scene_camera_frame(30_000, seed=16)
    raw.scaling += 3.0
    dbg.set_binning_mode(2)
    try:
        rep = _run(raw, scenes.identity_camera(64, 64, 60.0))
    finally:
        dbg.set_binning_mode(1)
    assert rep["R"] / 16 > 8192, rep["R"]


def test_mark_visible(cuda_device):
    from oracle import gs_oracle as go
    from gsworld_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

    raw = scenes.random_scene_camera_frame(10_000, seed=11, near_fraction=0.3)
    raw.xyz[::7, 2] *= -1
    cam = scenes.identity_camera(64, 64, 60.0).to(cuda_device)
    rs = GaussianRasterizationSettings(64, 64, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=cuda_device), 1.0,
                                       cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center,
                                       False, False, False)
    vis = GaussianRasterizer(rs).markVisible(raw.xyz.to(cuda_device)).cpu().numpy()
    ref = go.mark_visible(raw.xyz.numpy(), cam.world_view_transform.cpu().numpy().reshape(-1), 0.05)
    np.testing.assert_array_equal(vis, ref)


def test_api_errors(cuda_device):
    from gsworld_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

    dev = cuda_device
    cam = scenes.identity_camera(32, 32, 60.0).to(dev)
    rs = GaussianRasterizationSettings(32, 32, cam.tanfovx, cam.tanfovy, torch.zeros(3, device=dev), 1.0,
                                       cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center,
                                       False, False, False)
    r = GaussianRasterizer(rs)
    m = torch.zeros(4, 3, device=dev)
    with pytest.raises(Exception, match="excatly one of either SHs"):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1, device=dev))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1, device=dev), colors_precomp=torch.ones(4, 3, device=dev))
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(means3D=m.cpu(), means2D=m.cpu(), opacities=torch.ones(4, 1), colors_precomp=torch.ones(4, 3),
          scales=torch.ones(4, 3), rotations=torch.ones(4, 4))


def test_raw_parameter_space_matches_the_canonical_activations(cuda_device):
    """logits / log scales / un-normalised quaternions handed over as stored (GSR_RAW_*); preprocess
    applies sigmoid / exp / normalize in the float32 order fixed by oracle/gs_oracle.c (gso_activate_params), so
    every downstream float and index is still bit-exact against the oracle run on the activated values."""
    from oracle import gs_oracle as go

    raw = scenes.random_scene_camera_frame(40_000, seed=31)
    cam = scenes.identity_camera(320, 200, 60.0)
    inp = hp.np_inputs(raw, cam)  # activated by torch (sigmoid / exp / normalize)
    op, sc, ro = go.activate_params(raw.opacity.numpy().reshape(-1), raw.scaling.numpy(), raw.rotation.numpy(), flags=7)
    # the canonical activations agree with torch's to the last ulp or two ...
    assert np.abs(op - inp["opacities"].reshape(-1)).max() <= 2e-7
    assert (np.abs(sc - inp["scales"]) / inp["scales"]).max() <= 3e-7
    assert np.abs(ro - inp["rotations"]).max() <= 2e-7
    # ... and the HIP raw path reproduces the oracle on them bit for bit
    st = hp.oracle_settings(cam)
    bg = np.asarray((0.0, 0.1, 0.0), np.float32)
    act = dict(inp, opacities=op, scales=sc, rotations=ro)
    o = hp.oracle_forward(act, st, bg)
    rawin = dict(inp, opacities=raw.opacity.numpy().reshape(-1), scales=raw.scaling.numpy(),
                 rotations=raw.rotation.numpy())
    g = hp.gpu_forward(rawin, st, bg, param_space=7)
    rep = hp.compare_forward(o, g, st)
    assert rep["V"] > 10_000
    # one flag at a time composes the same way (opacity only: scales / rotations arrive activated)
    g1 = hp.gpu_forward(dict(act, opacities=rawin["opacities"]), st, bg, param_space=1)
    hp.compare_forward(o, g1, st)


@pytest.mark.parametrize("case", ["huge_offscreen", "wild_quaternions", "needles", "scaled_view", "cov3d_precomp"])
def test_wild_inputs_stay_bit_exact(cuda_device, case):
    """Inputs far from a trained model -- most centres off-screen with splats large enough to reach in, un-normalised
    quaternions, needles, a non-rigid view matrix, arbitrary symmetric cov3D -- must still give radii / tiles / rects
    (and everything downstream) bit-exact against the oracle.  (Written for a conservative screen-bound early reject
    in preprocess, which held on all of these but bought no time -- the kernel is latency-bound, DESIGN.md -- and
    was not kept; the cases stay.)"""
    rng = np.random.default_rng(41)
    n = 60_000
    raw = scenes.random_scene_camera_frame(n, seed=40, near_fraction=0.05)
    raw.xyz[:, :2] *= 6.0  # most centres far outside the frustum ...
    cam = scenes.identity_camera(320, 240, 50.0)
    kw = {}
    if case == "huge_offscreen":  # ... but big enough to reach into it
        raw.scaling += torch.from_numpy(rng.uniform(0.0, 5.5, (n, 1)).astype(np.float32))
    elif case == "wild_quaternions":  # un-normalised on purpose: R is not a rotation
        raw.rotation *= torch.from_numpy(rng.uniform(0.2, 3.0, (n, 1)).astype(np.float32))
        raw.scaling += 2.0
    elif case == "needles":
        raw.scaling[:, 0] += 5.0
        raw.scaling[:, 1:] -= 2.0
    inp = hp.np_inputs(raw, cam)
    if case == "wild_quaternions":
        inp["rotations"] = raw.rotation.numpy().copy()  # bypass the normalising activation
    if case == "scaled_view":  # non-rigid view matrix (|W|_2 = 1.8): the Gershgorin term of the bound matters
        V = inp["viewmatrix"].reshape(4, 4).astype(np.float64)
        P = np.linalg.inv(V) @ inp["projmatrix"].reshape(4, 4).astype(np.float64)
        V[:3, :3] *= 1.8
        inp["viewmatrix"] = V.astype(np.float32).reshape(-1)
        inp["projmatrix"] = (V @ P).astype(np.float32).reshape(-1)
        raw.scaling += 2.5
        inp["scales"] = np.exp(raw.scaling.numpy())
    st = hp.oracle_settings(cam)
    bg = np.zeros(3, np.float32)
    if case == "cov3d_precomp":  # arbitrary PSD matrices, sizes over 4 decades (a non-PSD one is garbage upstream too:
        # NaN radius -> radius 0 with a non-empty rect, which duplicateWithKeys then skips)
        A = rng.standard_normal((n, 3, 3)).astype(np.float32) * rng.uniform(0.001, 3.0, (n, 1, 1)).astype(np.float32)
        S = A @ A.transpose(0, 2, 1)
        cov = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)
        kw = dict(cov3D_precomp=np.ascontiguousarray(cov))
    o = hp.oracle_forward(inp, st, bg, **kw)
    g = hp.gpu_forward(inp, st, bg, **kw)
    rep = hp.compare_forward(o, g, st, check_image=(case != "cov3d_precomp"))
    assert 500 < rep["V"] < rep["P"], rep["V"]


def test_a_model_of_more_than_8192_blocks(cuda_device):
    """2.4 M Gaussians = 9 375 preprocess blocks: the depth sort's prepare workgroup then keeps 16 block counts per thread
    slice instead of 8 (depthsort.hip ss_prepare_per) -- every index against the oracle, inference frames (the path the
    closed loop takes) against the default frame."""
    from gsworld_amd.renderer import FrameRenderer

    raw = scenes.random_scene_camera_frame(2_400_000, seed=17)
    cam = scenes.identity_camera(96, 64, 70.0)
    rep = _run(raw, cam)
    assert rep["P"] == 2_400_000 and rep["V"] > 100_000
    dev = cuda_device
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    camd = cam.to(dev)
    a = FrameRenderer(dev).render(camd, means, op, shs=shs, scales=sc, rotations=rot)
    b = FrameRenderer(dev, forward_only=True)
    for _ in range(3):  # exact frame, then kept splitters validated / taken blind
        out = b.render(camd, means, op, shs=shs, scales=sc, rotations=rot)
    assert torch.equal(a[0], out[0]) and torch.equal(a[2], out[2]) and torch.equal(a[1], out[1])
