"""GPU parity of the backward pass: gsr_backward vs the backward oracle, and the autograd
path of the drop-in ``GaussianRasterizer``.  Tolerance: the CUDA / HIP kernels accumulate float atomics in an
arbitrary order and use the hardware exp2, the oracle sums the same float terms in binary64: 99.9 % of the
entries within 2e-3 relative (+1e-3 of the tensor's max as absolute floor), worst entry within 2e-2."""
import numpy as np
import pytest
import torch

from gsworld_amd import scenes
from tests import helpers as hp
from tests import helpers_bwd as hb

pytestmark = pytest.mark.gpu


def test_wave_sum_dpp_selftest(cuda_device):
    from gsworld_amd import _backward

    x = torch.randn(256, device=cuda_device)
    out = _backward.selftest_wave_sum(x).cpu()
    want = x.cpu().double().view(4, 64).sum(1)
    assert torch.allclose(out[:4].double(), want, atol=1e-4), (out, want)
    # the 10-component transpose-reduce (permlane32/16 swaps + row DPP): component c sums (c+1)*x over lanes l % (c+2) == 0
    lanes = torch.arange(64)
    xw = x.cpu().double().view(4, 64)
    want10 = torch.stack([((c + 1) * xw * (lanes % (c + 2) == 0)).sum(1) for c in range(10)], 1)  # (4, 10)
    assert torch.allclose(out[4:].double().view(4, 10), want10, atol=1e-4), (out[4:].view(4, 10), want10)
    ones = torch.arange(256, device=cuda_device, dtype=torch.float32)
    out = _backward.selftest_wave_sum(ones).cpu()
    assert out[:4].tolist() == [2016.0, 6112.0, 10208.0, 14304.0]
    iw = ones.cpu().double().view(4, 64)
    want10 = torch.stack([((c + 1) * iw * (lanes % (c + 2) == 0)).sum(1) for c in range(10)], 1)
    assert torch.equal(out[4:].double().view(4, 10), want10)  # integers: exact in any summation order


@pytest.mark.parametrize("n,w,h,aa,deg", [(2000, 64, 48, False, 3), (5000, 96, 64, True, 3), (3000, 70, 50, False, 1),
                                          (20000, 160, 120, False, 3)])
def test_backward_matches_oracle(cuda_device, n, w, h, aa, deg):
    rep = hb.run_case(n, w, h, seed=100 + n, aa=aa, deg=deg, **hb.SUITE_TOLERANCES)
    assert set(rep) >= {"dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dmeans2D"}


def test_backward_without_invdepth_and_black_background(cuda_device):
    hb.run_case(3000, 64, 64, seed=7, bg=(0, 0, 0), with_invdepth=False, **hb.SUITE_TOLERANCES)


def test_backward_precomputed_colors_and_cov(cuda_device):
    from oracle import gs_oracle as go

    raw = scenes.random_scene_camera_frame(3000, seed=17)
    raw.scaling += 0.5
    cam = scenes.identity_camera(80, 64, 60.0)
    inp = hp.np_inputs(raw, cam)
    st = hp.oracle_settings(cam)
    bg = np.float32([0.2, 0.2, 0.2])
    f0 = hp.oracle_forward(inp, st, bg, border_eps=0.0)
    rng = np.random.default_rng(5)
    colors = rng.random((3000, 3), dtype=np.float32)
    cov = f0["geom"]["cov3D"].copy()
    cov[f0["geom"]["radii"] == 0] = np.float32([1e-4, 0, 0, 1e-4, 0, 1e-4])
    fwd = hp.oracle_forward(inp, st, bg, border_eps=0.0, colors_precomp=colors, cov3D_precomp=cov)
    dLc = rng.standard_normal((3, 64, 80)).astype(np.float32)
    ref = go.backward(st, fwd, inp, bg, dLc, None, colors_precomp=colors, cov3D_precomp=cov)
    got, _ = hb.gpu_forward_backward(inp, st, bg, dLc, None, colors_precomp=colors, cov3D_precomp=cov)
    hb.compare_grads(ref, got, names=("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D"))


def test_autograd_through_gaussian_rasterizer(cuda_device):
    """loss.backward() through the drop-in module returns the oracle's gradients for every differentiable input."""
    from oracle import gs_oracle as go
    from gsworld_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

    dev = cuda_device
    raw = scenes.random_scene_camera_frame(4000, seed=23)
    raw.scaling += 0.5
    cam = scenes.identity_camera(96, 80, 60.0)
    inp = hp.np_inputs(raw, cam)
    st = hp.oracle_settings(cam)
    bg = np.float32([0.1, 0.2, 0.3])
    fwd = hp.oracle_forward(inp, st, bg, border_eps=0.0)
    rng = np.random.default_rng(9)
    dLc = rng.standard_normal((3, 80, 96)).astype(np.float32)
    dLd = rng.standard_normal((1, 80, 96)).astype(np.float32)
    ref = go.backward(st, fwd, inp, bg, dLc, dLd)

    camd = cam.to(dev)
    means, shs, op, sc, rot = [t.to(dev).requires_grad_(True) for t in raw.activated()]
    means2D = torch.zeros_like(means, requires_grad=True)
    rs = GaussianRasterizationSettings(80, 96, cam.tanfovx, cam.tanfovy, torch.from_numpy(bg).to(dev), 1.0,
                                       camd.world_view_transform, camd.full_proj_transform, 3, camd.camera_center,
                                       False, False, False)
    color, radii, invd = GaussianRasterizer(rs)(means3D=means, means2D=means2D, shs=shs, opacities=op, scales=sc,
                                                rotations=rot)
    ((color * torch.from_numpy(dLc).to(dev)).sum() + (invd * torch.from_numpy(dLd).to(dev)).sum()).backward()
    got = dict(dL_dmeans3D=means.grad, dL_dmeans2D=means2D.grad, dL_dsh=shs.grad, dL_dopacity=op.grad,
               dL_dscales=sc.grad, dL_drotations=rot.grad)
    hb.compare_grads(ref, {k: v.cpu().numpy() for k, v in got.items()}, names=tuple(got))
    assert radii.dtype == torch.int32 and int((radii > 0).sum()) == int((fwd["geom"]["radii"] > 0).sum())


@pytest.mark.parametrize("aa", [False, True])
def test_fused_parameter_space_gradients_match_the_torch_packing(cuda_device, aa):
    """SURVEY.md 8f-2 for training: raw opacity / scale / rotation parameters and the two SH tensors through the
    autograd Function (activations and their chain rule inside the kernels) against upstream's packing (torch sigmoid /
    exp / normalize / cat feeding the same rasterizer, autograd doing their backward)."""
    from gsworld_amd import scenes
    from gsworld_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

    dev = cuda_device
    S = 96
    cam = scenes.training_camera(S, S, 60.0).to(dev)
    raw = scenes.random_scene_camera_frame(6_000, seed=41).to(dev)
    raw.scaling += 1.0
    bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
    rs = GaussianRasterizationSettings(S, S, cam.tanfovx, cam.tanfovy, bg, 1.0, cam.world_view_transform,
                                       cam.full_proj_transform, 3, cam.camera_center, False, False, aa)
    rast = GaussianRasterizer(rs)
    gen = torch.Generator(device="cpu").manual_seed(5)
    w_img = torch.randn((3, S, S), generator=gen).to(dev)
    w_dep = torch.randn((1, S, S), generator=gen).to(dev)
    names = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")

    def run(fused):
        ps = [getattr(raw, n).detach().clone().requires_grad_(True) for n in names]
        xyz, dc, rest, op, sc, rot = ps
        m2d = torch.zeros_like(xyz, requires_grad=True)
        if fused:
            color, radii, invd = rast(means3D=xyz, means2D=m2d, shs=dc, shs_rest=rest, opacities=op, scales=sc,
                                      rotations=rot, param_space=7)
        else:
            color, radii, invd = rast(means3D=xyz, means2D=m2d, shs=torch.cat((dc, rest), dim=1),
                                      opacities=torch.sigmoid(op), scales=torch.exp(sc),
                                      rotations=torch.nn.functional.normalize(rot))
        ((color * w_img).sum() + (invd * w_dep).sum()).backward()
        return color.detach(), radii, [p.grad for p in ps] + [m2d.grad]

    c0, r0, g0 = run(False)
    c1, r1, g1 = run(True)
    # forward: canonical exp vs torch's (last-ulp alpha changes can flip a threshold decision on a few pixels)
    assert float(((c1 - c0).abs() > 1e-5).float().mean()) <= 2e-3
    assert int((r0 != r1).sum()) <= 3
    for n, a, b in zip(names + ("means2D",), g0, g1):
        assert a is not None and b is not None and a.shape == b.shape, n
        scale = float(a.abs().max()) + 1e-12
        err = float((a - b).abs().max()) / scale
        assert err <= 2e-3, (n, err)
        assert float(b.abs().max()) > 0, n


def test_config5_resolution_800x800_against_the_oracle(cuda_device):
    """BASELINE.json configs[4] at its resolution: 800x800 = 2500 tiles takes the TWO-BAND counting placement
    (binning.hip place_band_rows) and the backward walks those lists -- forward image and all eight gradients against
    the oracle with 50 k Gaussians (the oracle needs about a second for this)."""
    rep = hb.run_case(50_000, 800, 800, seed=5, scale_boost=0.3, **hb.SUITE_TOLERANCES)
    assert set(rep) >= {"dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dmeans2D"}


def test_config5_full_size_gradients_against_the_oracle(cuda_device, capsys):
    """BASELINE.json configs[4] at FULL size -- 500 k Gaussians, 800x800 -- all eight gradients against the backward
    oracle (float64 sums; ~20 s of host time).  render_backward_kernel takes one hardware reciprocal of (1 - alpha) per
    hit instead of two correctly rounded divisions (backward.hip, round 3): this is the assertion of what that costs at
    the size the step is benchmarked at -- EVERY element within 2e-3 of its output's scale, in every run."""
    raw = scenes.random_scene_camera_frame(500_000, seed=5, near_fraction=0.0)
    # ONE run, no repeat (rounds 3-5 allowed a second try: the worst element moved between 1.9e-4 and 1.7e-3 from run to
    # run).  That was the ORDER in which a Gaussian's per-tile sums reached its record -- device-scope binary32 atomics in
    # whatever order the tiles' workgroups retire; the record is binary64 since round 6 (backward.hip, GsrGradWord) and the
    # figure is the same in every run: 9.3e-4 on this build (profiles/round6/config5_gradient_bound.txt: four runs each
    # way, same box).  What is left is float arithmetic inside a tile against the oracle's binary64: a bound, not noise.
    rep = hb.run_case(500_000, 800, 800, seed=5, scale_boost=0.0, raw=raw, **hb.SUITE_TOLERANCES)
    worst = max(v["max_norm_err"] for v in rep.values())
    with capsys.disabled():
        print(f"\n[configs[4] full size] worst normalised gradient error {worst:.3e}; per output: " +
              ", ".join(f"{k} {v['max_norm_err']:.1e} ({100 * v['frac_within']:.3f} % within 2e-3)" for k, v in rep.items()))
    assert worst <= 2e-3, f"worst normalised gradient error {worst} > 2e-3 at configs[4] size"
    assert set(rep) >= {"dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dmeans2D"}


@pytest.mark.parametrize("near_fraction", [0.01, 0.0])
def test_config5_full_size_training_step(cuda_device, near_fraction):
    """BASELINE.json configs[4] at full size: 500 k Gaussians, 800x800, loss = 0.8 L1 + 0.2 (1 - fused_ssim) through the
    drop-in modules exactly as upstream train.py calls them.  Size-independent properties: finite gradients of the right
    shapes, zero gradient exactly where the forward culled, a descent step along the gradient lowers the loss, and the
    fused-parameter path (activations + chain rule inside the kernels) agrees with upstream's torch packing.

    near_fraction = 0.01 is SURVEY.md 8d's scene (config-1 distribution): its 5000 splats at z in (0.05, 0.2) are
    hundreds of pixels wide and opaque, so every pixel saturates on a handful of them and only ~100 Gaussians receive
    any gradient -- the step is dominated by list walking and early termination.  near_fraction = 0 removes that band:
    the same step with tens of thousands of contributing Gaussians (what a real training view looks like)."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "gsworld_amd", "dropin"))
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from fused_ssim import fused_ssim

    dev = cuda_device
    S, N = 800, 500_000
    cam = scenes.training_camera(S, S, 60.0).to(dev)
    raw = scenes.random_scene_camera_frame(N, seed=5, near_fraction=near_fraction).to(dev)
    tgt = scenes.random_scene_camera_frame(N, seed=5, near_fraction=near_fraction).to(dev)
    gen = torch.Generator(device="cpu").manual_seed(6)
    tgt.xyz += (0.01 * torch.randn(tgt.xyz.shape, generator=gen)).to(dev)
    tgt.features_dc += (0.1 * torch.randn(tgt.features_dc.shape, generator=gen)).to(dev)
    bg = torch.zeros(3, device=dev)
    rs = GaussianRasterizationSettings(S, S, cam.tanfovx, cam.tanfovy, bg, 1.0, cam.world_view_transform,
                                       cam.full_proj_transform, 3, cam.camera_center, False, False, False)
    rast = GaussianRasterizer(rs)
    names = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")

    def forward(r, fused, grad):
        ps = [getattr(r, n).detach().clone().requires_grad_(grad) for n in names]
        xyz, dc, rest, op, sc, rot = ps
        m2d = torch.zeros_like(xyz, requires_grad=grad)
        if fused:
            img, radii, _ = rast(means3D=xyz, means2D=m2d, shs=dc, shs_rest=rest, opacities=op, scales=sc,
                                 rotations=rot, param_space=7)
        else:
            img, radii, _ = rast(means3D=xyz, means2D=m2d, shs=torch.cat((dc, rest), dim=1),
                                 opacities=torch.sigmoid(op), scales=torch.exp(sc),
                                 rotations=torch.nn.functional.normalize(rot))
        return img.clamp(0, 1), radii, ps, m2d

    with torch.no_grad():
        gt = forward(tgt, False, False)[0].detach()

    def step(fused):
        img, radii, ps, m2d = forward(raw, fused, True)
        loss = 0.8 * (img - gt).abs().mean() + 0.2 * (1.0 - fused_ssim(img[None], gt[None]))
        loss.backward()
        return float(loss.detach()), radii, ps, m2d

    l0, radii, ps, m2d = step(False)
    vis = radii > 0
    assert 0.3 * N < int(vis.sum()) <= N and 0.0 < l0 < 1.0
    for n, p in zip(names + ("means2D",), ps + [m2d]):
        g = p.grad
        assert g is not None and g.shape == p.shape and bool(torch.isfinite(g).all()), n
        assert float(g.abs().max()) > 0.0, f"{n}: all-zero gradient"
        flat = g.reshape(N, -1)
        if int((~vis).sum()) > 0:
            assert float(flat[~vis].abs().max()) == 0.0, f"{n}: gradient on a culled Gaussian"
    touched = int((ps[1].grad.reshape(N, -1).abs().sum(1) > 0).sum())
    assert touched > (50 if near_fraction > 0 else 20_000), touched
    # a small step against the gradient lowers the loss (not asserted on the near-band scene: its image hangs on ~100
    # screen-filling splats whose projection is violently non-linear in their position)
    for n, p in zip(names, ps):
        with torch.no_grad():
            getattr(raw, n).copy_(p.detach() - 2e-3 * p.grad / (p.grad.abs().max() + 1e-12))
    l1, r1, ps1, _ = step(False)
    assert near_fraction > 0 or l1 < l0, (l0, l1)
    # fused parameter space (raw parameters + split SH, chain rule inside the kernels): same loss, same gradients
    lf, rf, psf, _ = step(True)
    assert abs(lf - l1) < 1e-4 and int((rf != r1).sum()) <= 5
    for n, a, b in zip(names, ps1, psf):
        scale = float(a.grad.abs().max()) + 1e-12
        assert float((a.grad - b.grad).abs().max()) / scale <= 5e-3, n


def test_training_forward_without_the_mid_frame_host_read(cuda_device):
    """The fused autograd path sizes its instance list by a bound no frame can exceed (P x tiles: _C.nosync_capacity)
    instead of reading num_rendered back in the middle of the frame.  Same kernels: the image, radii and inverse depth
    are the exact-mode frame's bits, the gradients agree to the order of the atomic sums; with the budget too small
    for the bound the same call takes the exact mode again."""
    from gsworld_amd import _C, scenes
    from gsworld_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

    dev = cuda_device
    S = 160
    cam = scenes.training_camera(S, S, 60.0).to(dev)
    raw = scenes.random_scene_camera_frame(20_000, seed=43).to(dev)
    raw.scaling += 0.7
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    rast = GaussianRasterizer(GaussianRasterizationSettings(S, S, cam.tanfovx, cam.tanfovy, bg, 1.0, cam.world_view_transform,
                                                            cam.full_proj_transform, 3, cam.camera_center, False, False, False))
    w_img = torch.randn((3, S, S), generator=torch.Generator().manual_seed(6)).to(dev)
    names = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")

    def run():
        ps = [getattr(raw, n).detach().clone().requires_grad_(True) for n in names]
        xyz, dc, rest, op, sc, rot = ps
        m2d = torch.zeros_like(xyz, requires_grad=True)
        color, radii, invd = rast(means3D=xyz, means2D=m2d, shs=dc, shs_rest=rest, opacities=op, scales=sc,
                                  rotations=rot, param_space=7)
        (color * w_img).sum().backward()
        return color.detach(), radii, invd.detach(), [p.grad for p in ps] + [m2d.grad]

    assert _C.nosync_capacity(20_000, S, S) == 20_000 * 100
    fast = run()
    budget, _C.NOSYNC_LIST_BYTES = _C.NOSYNC_LIST_BYTES, 1 << 10
    try:
        assert _C.nosync_capacity(20_000, S, S) is None
        exact = run()
    finally:
        _C.NOSYNC_LIST_BYTES = budget
    for a, b, what in zip(fast[:3], exact[:3], ("color", "radii", "invdepth")):
        assert torch.equal(a, b), what
    for n, a, b in zip(names + ("means2D",), fast[3], exact[3]):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) / scale <= 1e-5, n
    assert _C.nosync_capacity(2_000_000, 1920, 1080) is None       # 65 G instances: not this way
    assert _C.nosync_capacity(100, 16, 16 * 300) is None           # wider than the counting placement takes
