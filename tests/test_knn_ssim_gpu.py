"""GPU parity of simple_knn.distCUDA2 (bit-exact vs the O(N^2) oracle) and fused_ssim (forward <= 1e-6,
gradient <= 1e-6 relative to its scale vs float64 autograd of the conv2d definition)."""
import numpy as np
import pytest
import torch

from oracle import gs_oracle as go
from oracle import ssim_ref

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,kind", [(1, "u"), (3, "u"), (4, "u"), (1000, "u"), (5000, "c"), (20000, "u"), (30011, "c")])
def test_distcuda2_bit_exact(cuda_device, n, kind):
    from gsworld_amd.dropin.simple_knn._C import distCUDA2

    rng = np.random.default_rng(n)
    if kind == "u":
        pts = rng.random((n, 3), dtype=np.float32) * 4 - 2
    else:
        centers = rng.random((30, 3)) * 10
        pts = (centers[rng.integers(0, 30, n)] + rng.normal(0, 0.03, (n, 3))).astype(np.float32)
        pts[::97] = pts[0]  # exact duplicates
    got = distCUDA2(torch.from_numpy(pts).to(cuda_device)).cpu().numpy()
    want = go.knn_dist2(pts)
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


def test_distcuda2_degenerate_axis_and_errors(cuda_device):
    from gsworld_amd.knn import distCUDA2

    pts = np.zeros((500, 3), np.float32)
    pts[:, 0] = np.linspace(0, 1, 500, dtype=np.float32)  # y, z extents are zero
    got = distCUDA2(torch.from_numpy(pts).to(cuda_device)).cpu().numpy()
    np.testing.assert_array_equal(got.view(np.uint32), go.knn_dist2(pts).view(np.uint32))
    assert distCUDA2(torch.empty(0, 3, device=cuda_device)).numel() == 0
    with pytest.raises(RuntimeError, match="no CPU path"):
        distCUDA2(torch.zeros(4, 3))
    with pytest.raises(RuntimeError, match=r"\(num_points, 3\)"):
        distCUDA2(torch.zeros(4, 2, device=cuda_device))


@pytest.mark.parametrize("shape,padding", [((1, 3, 64, 80), "same"), ((2, 3, 37, 53), "same"), ((1, 1, 16, 16), "same"),
                                           ((1, 3, 800, 800), "same"), ((2, 3, 40, 44), "valid")])
def test_fused_ssim_forward_backward(cuda_device, shape, padding):
    from gsworld_amd.dropin.fused_ssim import fused_ssim

    gen = torch.Generator().manual_seed(sum(shape))
    a = torch.rand(*shape, generator=gen)
    b = (a + 0.1 * torch.randn(*shape, generator=gen)).clamp(0, 1)
    a_ref = a.double().requires_grad_(True)
    s_ref = ssim_ref.ssim(a_ref, b.double(), padding)
    s_ref.backward()
    a_gpu = a.to(cuda_device).requires_grad_(True)
    s = fused_ssim(a_gpu, b.to(cuda_device), padding=padding)
    s.backward()
    assert abs(float(s) - float(s_ref)) < 2e-6
    g, g_ref = a_gpu.grad.cpu().double(), a_ref.grad
    assert float((g - g_ref).abs().max()) <= 2e-5 * float(g_ref.abs().max())
    # inference mode returns the same value without storing the partial maps
    s2 = fused_ssim(a.to(cuda_device), b.to(cuda_device), padding=padding, train=False)
    assert abs(float(s2) - float(s)) < 1e-7


@pytest.mark.parametrize("shape,lam,clamp", [((3, 64, 80), 0.2, True), ((3, 37, 53), 0.2, False), ((1, 16, 16), 0.5, True),
                                              ((3, 800, 800), 0.2, True), ((2, 3, 40, 44), 0.0, True), ((3, 21, 19), 1.0, True)])
def test_photometric_loss_is_the_training_loss_and_its_gradient(cuda_device, shape, lam, clamp):
    """``photometric_loss`` (one autograd node, three kernels) against what the training step composes out of torch
    operations and fused_ssim -- ``(1 - lam) * |x - t|.mean() + lam * (1 - ssim(x, t))``, ``x = img.clamp(0, 1)`` -- in
    binary64 autograd (oracle/ssim_ref.py: the conv2d definition).  The image goes beyond [0, 1] in places (the clamp's
    backward must stop the gradient there, bounds included) and equals the target in others (sign(0) = 0)."""
    from gsworld_amd.ssim import photometric_loss

    gen = torch.Generator().manual_seed(sum(shape) + int(10 * lam))
    t = torch.rand(*shape, generator=gen)
    img = t + 0.2 * torch.randn(*shape, generator=gen)      # below 0 and above 1 here and there
    img.view(-1)[::7] = t.view(-1)[::7]                      # exact hits: |.| is not differentiable, torch says 0
    img.view(-1)[3::11] = 1.0                                # on the clamp's upper bound: gradient passes
    img.view(-1)[5::13] = 0.0
    x_ref = img.double().requires_grad_(True)
    xc = x_ref.clamp(0, 1) if clamp else x_ref
    four = (lambda z: z if z.dim() == 4 else z[None])       # noqa: E731
    want = (1 - lam) * (xc - t.double()).abs().mean() + lam * (1 - ssim_ref.ssim(four(xc), four(t.double()), "same"))
    want.backward()
    x = img.to(cuda_device).requires_grad_(True)
    got = photometric_loss(x, t.to(cuda_device), lam, clamp)
    (2.0 * got).backward()                                   # an incoming gradient other than 1
    assert got.shape == () and abs(float(got) - float(want)) < 2e-6
    g, g_ref = x.grad.cpu().double(), 2.0 * x_ref.grad
    assert float((g - g_ref).abs().max()) <= 2e-5 * float(g_ref.abs().max())
    if clamp:
        outside = (img < 0) | (img > 1)
        assert outside.any() and float(g[outside].abs().max()) == 0.0
    # the same number twice (fixed summation order), and no gradient pass when none is asked for
    again = photometric_loss(img.to(cuda_device), t.to(cuda_device), lam, clamp)
    assert float(again) == float(got)


def test_photometric_loss_matches_the_composed_gpu_ops(cuda_device):
    """... and against the composition it replaces ON THE DEVICE (torch ops + this package's fused_ssim), whose float32
    rounding it shares: value within 1e-6, gradient within 1e-6 of the largest entry."""
    from gsworld_amd.dropin.fused_ssim import fused_ssim
    from gsworld_amd.ssim import photometric_loss

    gen = torch.Generator().manual_seed(77)
    t = torch.rand(3, 200, 240, generator=gen).to(cuda_device)
    img = (t.cpu() + 0.15 * torch.randn(3, 200, 240, generator=gen)).to(cuda_device)
    a = img.clone().requires_grad_(True)
    xc = a.clamp(0, 1)
    want = 0.8 * (xc - t).abs().mean() + 0.2 * (1.0 - fused_ssim(xc[None], t[None]))
    want.backward()
    b = img.clone().requires_grad_(True)
    got = photometric_loss(b, t, 0.2, True)
    got.backward()
    assert abs(float(got) - float(want)) < 1e-6
    assert float((a.grad - b.grad).abs().max()) <= 1e-6 * float(a.grad.abs().max())
    with pytest.raises(RuntimeError, match="no CPU path"):
        photometric_loss(img.cpu(), t.cpu())
