"""End-to-end through the drop-in import surfaces exactly as GSWorld reaches them: ``dropin`` and
``gs_compat`` on sys.path, then ``from gaussian_renderer import render`` / ``from scene.cameras import Camera`` /
``from scene.gaussian_model import GaussianModel`` / ``from simple_knn._C import distCUDA2`` / ``fused_ssim``."""
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gs_paths():
    added = [os.path.join(ROOT, "gsworld_amd", "dropin"), os.path.join(ROOT, "gsworld_amd", "gs_compat")]
    for p in added:
        sys.path.insert(0, p)
    yield
    for p in added:
        sys.path.remove(p)
    for m in [k for k in sys.modules if k.split(".")[0] in ("scene", "gaussian_renderer", "arguments", "utils",
                                                            "diff_gaussian_rasterization", "simple_knn", "fused_ssim")]:
        del sys.modules[m]


def test_render_through_gs_compat_matches_oracle(cuda_device, gs_paths):
    from arguments import PipelineParams
    from gaussian_renderer import render
    from scene.cameras import Camera
    from scene.gaussian_model import GaussianModel

    from gsworld_amd import scenes
    from tests import helpers as hp

    try:
        import diff_gaussian_rasterization
        assert not hasattr(diff_gaussian_rasterization, "SparseGaussianAdam")  # GSWorld's probe must fail
    except ImportError:
        pytest.fail("drop-in diff_gaussian_rasterization not importable")

    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=200_000, seed=3)
    pc = GaussianModel(3)
    pc._xyz, pc._features_dc, pc._features_rest = raw.xyz.to(dev), raw.features_dc.to(dev), raw.features_rest.to(dev)
    pc._opacity = raw.opacity.to(dev)[..., None]  # (N,1,1) as Semantic3DGSWrapper.load_ply leaves it
    pc._scaling, pc._rotation = raw.scaling.to(dev), raw.rotation.to(dev)
    pc.active_sh_degree = 3
    ref_cam = scenes.sensor_camera("xarm6_align")
    # the wrapper builds the Camera from (R = world2cam[:3,:3].T, T = world2cam[:3,3]) (gs_world_wrapper.py:300-322)
    W2C = ref_cam.world_view_transform.T
    cam = Camera(resolution=(640, 480), colmap_id=0, R=W2C[:3, :3].T.numpy(), T=W2C[:3, 3].numpy(), FoVx=ref_cam.FoVx,
                 FoVy=ref_cam.FoVy, depth_params=None, image=None, invdepthmap=None, image_name="right_cam", uid=0,
                 data_device=dev)
    np.testing.assert_allclose(cam.full_proj_transform.cpu().numpy(), ref_cam.full_proj_transform.numpy(), atol=1e-6)
    pipe = PipelineParams().extract(types.SimpleNamespace())
    bg = torch.zeros(3, device=dev)
    out = render(cam, pc, pipe, bg, use_trained_exp=False, separate_sh=False)
    assert set(out) == {"render", "viewspace_points", "visibility_filter", "radii", "depth"}
    # frozen parameters (what Semantic3DGSWrapper.load_ply produces): inference fast path, SH read as stored
    # (features_dc | features_rest) with no per-frame cat.  A parameter that requires grad takes upstream's autograd
    # path (cat + Function); both must give the same image bit for bit.
    assert not out["render"].requires_grad
    pc._xyz.requires_grad_(True)
    out_ag = render(cam, pc, pipe, bg, use_trained_exp=False, separate_sh=False)
    pc._xyz.requires_grad_(False)
    assert out_ag["render"].requires_grad
    assert torch.equal(out_ag["render"].detach(), out["render"]) and torch.equal(out_ag["radii"], out["radii"])
    assert torch.equal(out_ag["depth"].detach(), out["depth"])
    # (the shortcut computes the entries GSWorld never reads on first access: same values as upstream's eager dict)
    assert torch.equal(out["visibility_filter"], out_ag["visibility_filter"])
    assert out["viewspace_points"].shape == out_ag["viewspace_points"].shape and out["viewspace_points"].requires_grad
    # the fast path renders through a cached renderer: returned frames must not alias its buffers, and a scene that
    # outgrows the instance capacity remembered from the previous frame must be re-rendered, not truncated
    keep = {k: out[k].clone() for k in ("render", "radii", "depth")}
    small = GaussianModel(3)
    sel = slice(0, 2_000)
    small._xyz, small._features_dc, small._features_rest = pc._xyz[sel], pc._features_dc[sel].contiguous(), \
        pc._features_rest[sel].contiguous()
    small._opacity, small._scaling, small._rotation = pc._opacity[sel], pc._scaling[sel], pc._rotation[sel]
    small.active_sh_degree = 3
    import gaussian_renderer as gr_mod

    out_small = render(cam, small, pipe, bg)
    assert out_small["radii"].shape == (2_000,)
    assert all(torch.equal(out[k], keep[k]) for k in keep)
    gr_mod._frame_renderer(pc._xyz.device).r_capacity = 1 << 16  # as if the previous scene had been tiny
    out_again = render(cam, pc, pipe, bg)
    assert all(torch.equal(out_again[k], keep[k]) for k in keep)
    # opt-in pipe.fused_activations: raw parameters, activations inside preprocess (canonical exp, not torch's):
    # the image moves in the last bits only and (almost) every radius is unchanged
    pipe_f = types.SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False,
                                   antialiasing=False, fused_activations=True)
    out_f = render(cam, pc, pipe_f, bg)
    # (a last-ulp change of an alpha can flip a 1/255 or termination decision: bounded like the oracle's borderline
    # pixels, and rare)
    diff = (out_f["render"] - out["render"]).abs()
    assert float(diff.max()) <= 0.02 and float((diff > 1e-5).float().mean()) <= 2e-3
    assert int((out_f["radii"] != out["radii"]).sum()) <= 20
    # the same opt-in with trainable parameters: one autograd Function over the raw parameters and the two SH tensors
    leaves = [pc._xyz, pc._opacity, pc._scaling, pc._rotation, pc._features_dc, pc._features_rest]
    for t in leaves:
        t.requires_grad_(True)
    out_t = render(cam, pc, pipe_f, bg)
    assert out_t["render"].requires_grad and torch.equal(out_t["render"].detach(), out_f["render"])
    out_t["render"].sum().backward()
    for t in leaves:
        assert t.grad is not None and t.grad.shape == t.shape and bool(torch.isfinite(t.grad).all())
        assert float(t.grad.abs().max()) > 0
        t.requires_grad_(False)
        t.grad = None
    img = out["render"].detach()
    assert img.shape == (3, 480, 640) and float(img.min()) >= 0 and float(img.max()) <= 1
    # same frame through the oracle
    inp = hp.np_inputs(raw, ref_cam)
    inp["viewmatrix"] = cam.world_view_transform.cpu().numpy().reshape(-1)
    inp["projmatrix"] = cam.full_proj_transform.cpu().numpy().reshape(-1)
    inp["campos"] = cam.camera_center.cpu().numpy()
    o = hp.oracle_forward(inp, hp.oracle_settings(ref_cam), np.zeros(3, np.float32))
    ok = o["borderline"] == 0
    assert np.abs(img.cpu().numpy() - np.clip(o["color"], 0, 1))[:, ok].max() <= 1e-4
    assert int((out["radii"] > 0).sum()) == int((o["geom"]["radii"] > 0).sum())
    # GSWorld's uint8 conversion (gs_world_wrapper.py:268-270) vs FrameRenderer.pack_rgb8
    from gsworld_amd.renderer import FrameRenderer

    want = (img.permute(1, 2, 0).unsqueeze(0) * 255).clamp(0, 255).to(torch.uint8)[0]
    got = FrameRenderer(dev).pack_rgb8(img.contiguous())
    assert torch.equal(got, want)
    # convert_SHs_python / compute_cov3D_python pipeline flags route through colors_precomp / cov3D_precomp
    pipe2 = types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=True, debug=False, antialiasing=False)
    img2 = render(cam, pc, pipe2, bg)["render"].detach()
    assert float((img2 - img).abs().max()) < 2e-3


def test_create_from_pcd_uses_distcuda2_and_ssim_imports(cuda_device, gs_paths):
    from fused_ssim import fused_ssim
    from scene.gaussian_model import GaussianModel
    from simple_knn._C import distCUDA2

    from oracle import gs_oracle as go

    rng = np.random.default_rng(0)
    pts = rng.random((3000, 3)).astype(np.float32)
    pcd = types.SimpleNamespace(points=pts, colors=rng.random((3000, 3)).astype(np.float32))
    m = GaussianModel(3)
    m.create_from_pcd(pcd, [], 1.0)
    want = np.log(np.sqrt(np.maximum(go.knn_dist2(pts), 1e-7)))
    np.testing.assert_allclose(m._scaling[:, 0].detach().cpu().numpy(), want, rtol=1e-6)
    assert m._features_dc.shape == (3000, 1, 3) and m._features_rest.shape == (3000, 15, 3)
    assert torch.allclose(m.get_opacity, torch.full_like(m.get_opacity, 0.1), atol=1e-6)
    assert distCUDA2(torch.from_numpy(pts).to(cuda_device)).shape == (3000,)
    a = torch.rand(1, 3, 32, 32, device=cuda_device)
    assert abs(float(fused_ssim(a, a)) - 1.0) < 1e-6


def test_compiled_C_extension_is_loaded_and_matches_the_ctypes_binding(cuda_device):
    """the drop-in package's `_C` is a compiled torch extension (csrc_torch/ext.cpp) with upstream's
    positional signatures.  It must be the binding in use on a GPU box, and forward / backward / mark_visible through it
    must equal the ctypes binding of the same library bit for bit (same kernels, same arguments)."""
    import numpy as np

    from gsworld_amd import _C, scenes

    assert _C._ext is not None, "gsworld_amd/_C_ext*.so was not built: run `python -m gsworld_amd.build_ext`"
    assert "gfx950" in _C._ext.version()
    dev = cuda_device
    raw = scenes.random_scene_camera_frame(30_000, seed=77)
    raw.scaling += 0.5
    cam = scenes.identity_camera(200, 152, 60.0).to(dev)
    means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    e = torch.empty(0, device=dev)
    gen = torch.Generator().manual_seed(3)
    dLc = torch.randn((3, 152, 200), generator=gen).to(dev)
    dLd = torch.randn((1, 152, 200), generator=gen).to(dev)

    def run():
        out = _C.rasterize_gaussians(bg, means, e, op, sc, rot, 1.0, e, cam.world_view_transform,
                                     cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 152, 200, shs, 3,
                                     cam.camera_center, False, False, False)
        R, color, radii, gB, bB, iB, invd = out
        grads = _C.rasterize_gaussians_backward(bg, means, radii, e, op, sc, rot, 1.0, e, cam.world_view_transform,
                                                cam.full_proj_transform, cam.tanfovx, cam.tanfovy, dLc, dLd, shs, 3,
                                                cam.camera_center, gB, R, bB, iB, False, False)
        vis = _C.mark_visible(means, cam.world_view_transform, cam.full_proj_transform)
        return R, color, radii, invd, grads, vis

    a = run()
    ext, _C._ext = _C._ext, None  # the ctypes binding of the same shared library
    try:
        b = run()
    finally:
        _C._ext = ext
    assert a[0] == b[0] and a[0] > 0
    for x, y in zip(a[1:4], b[1:4]):
        assert torch.equal(x, y)
    assert torch.equal(a[5], b[5]) and a[5].dtype == torch.bool
    assert len(a[4]) == len(b[4]) == 8
    for x, y in zip(a[4], b[4]):  # float atomics: order differs from run to run
        assert x.shape == y.shape
        np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=2e-3, atol=1e-3 * float(y.abs().max()) + 1e-12)
    # upstream's error behaviour through the compiled module
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        _C.rasterize_gaussians(bg, means.reshape(-1), e, op, sc, rot, 1.0, e, cam.world_view_transform,
                               cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 152, 200, shs, 3, cam.camera_center,
                               False, False, False)
    with pytest.raises(RuntimeError, match="no CPU path"):
        _C.mark_visible(means.cpu(), cam.world_view_transform, cam.full_proj_transform)
