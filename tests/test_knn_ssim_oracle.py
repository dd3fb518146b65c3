"""Pins the KNN and SSIM oracles on the CPU (SURVEY.md 8c KATs 9 and 10)."""
import numpy as np
import torch
from scipy.spatial import cKDTree

from oracle import gs_oracle as go
from oracle import ssim_ref


def _clouds():
    rng = np.random.default_rng(0)
    uniform = rng.random((3000, 3), dtype=np.float32)
    centers = rng.random((20, 3)) * 10
    clustered = (centers[rng.integers(0, 20, 3000)] + rng.normal(0, 0.05, (3000, 3))).astype(np.float32)
    return {"uniform": uniform, "clustered": clustered}


def test_knn_oracle_matches_kdtree_and_definition():
    for name, pts in _clouds().items():
        got = go.knn_dist2(pts)
        d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
        want = (d[:, 1:4] ** 2).mean(1)
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-9, err_msg=name)
    # analytic: unit grid line 0,1,2,3,4 -> point 0: (1+4+9)/3, point 2: (1+1+4)/3
    line = np.stack([np.arange(5, dtype=np.float32), np.zeros(5, np.float32), np.zeros(5, np.float32)], 1)
    np.testing.assert_allclose(go.knn_dist2(line), [14 / 3, 6 / 3, 6 / 3, 6 / 3, 14 / 3], rtol=1e-6)
    # duplicates: self is excluded by index, the twin is not -> a zero enters the 3-best
    dup = np.float32([[0, 0, 0], [0, 0, 0], [1, 0, 0], [0, 2, 0]])
    np.testing.assert_allclose(go.knn_dist2(dup), [(0 + 1 + 4) / 3, (0 + 1 + 4) / 3, (1 + 1 + 5) / 3, (4 + 4 + 5) / 3],
                               rtol=1e-6)


def test_ssim_reference_known_answers():
    gen = torch.Generator().manual_seed(0)
    a = torch.rand(2, 3, 40, 52, generator=gen, dtype=torch.float64)
    assert abs(float(ssim_ref.ssim(a, a)) - 1.0) < 1e-12           # identical images
    b = torch.rand(2, 3, 40, 52, generator=gen, dtype=torch.float64)
    s = float(ssim_ref.ssim(a, b))
    assert -0.1 < s < 0.2                                            # independent noise
    assert abs(float(ssim_ref.ssim(a, b)) - float(ssim_ref.ssim(b, a))) < 1e-12  # symmetric
    g = ssim_ref.gaussian_window()
    assert abs(float(g.sum()) - 1) < 1e-15 and abs(float(g[5]) - 0.26601171) < 1e-7 and float(g[0]) < 0.00103
    # constant images: mu = c inside, variances 0 -> map = (2 c1 c2 + C1)/(c1^2 + c2^2 + C1) away from the border
    c1, c2 = 0.3, 0.6
    m = ssim_ref.ssim_map(torch.full((1, 1, 32, 32), c1, dtype=torch.float64),
                          torch.full((1, 1, 32, 32), c2, dtype=torch.float64))
    want = (2 * c1 * c2 + 1e-4) / (c1 * c1 + c2 * c2 + 1e-4)
    assert abs(float(m[0, 0, 16, 16]) - want) < 1e-12
    assert float(ssim_ref.ssim(a, b, "valid")) != s and ssim_ref.ssim_map(a, b)[:, :, 5:-5, 5:-5].shape == (2, 3, 30, 42)
