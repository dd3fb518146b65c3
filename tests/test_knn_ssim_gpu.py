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
