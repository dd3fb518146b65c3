"""PLY(+semantics) round trip and layout: property order, channel-major f_rest, load_ply shapes."""
import types

import numpy as np
import torch

from gsworld_amd import ply


def test_ply_roundtrip_and_layout(tmp_path):
    gen = torch.Generator().manual_seed(0)
    n = 37
    m = types.SimpleNamespace(
        _xyz=torch.randn(n, 3, generator=gen), _features_dc=torch.randn(n, 1, 3, generator=gen),
        _features_rest=torch.randn(n, 15, 3, generator=gen), _opacity=torch.randn(n, 1, 1, generator=gen),
        _scaling=torch.randn(n, 3, generator=gen), _rotation=torch.randn(n, 4, generator=gen),
        _semantics=torch.randint(0, 300, (n, 1), generator=gen).float(), max_sh_degree=3)
    path = str(tmp_path / "model.ply")
    ply.write_gaussian_ply(path, m)
    header = open(path, "rb").read(4096).split(b"end_header")[0].decode()
    props = [ln.split()[-1] for ln in header.splitlines() if ln.startswith("property")]
    want = (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(45)] +
            ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)] + ["semantics"])
    assert props == want and f"element vertex {n}" in header and len(props) == 63  # pcd_utils.py:68 "(…, 63)"
    cols = ply.read_ply(path)
    # f_rest is channel-major on disk: column c*15 + k holds coefficient k+1 of channel c
    np.testing.assert_array_equal(cols["f_rest_16"], m._features_rest[:, 1, 1].numpy())
    np.testing.assert_array_equal(cols["f_rest_2"], m._features_rest[:, 2, 0].numpy())
    back = types.SimpleNamespace(max_sh_degree=3)
    ply.read_gaussian_ply(path, back, device="cpu")
    for attr in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_semantics"):
        assert getattr(back, attr).shape == getattr(m, attr).shape, attr
        assert torch.equal(getattr(back, attr), getattr(m, attr)), attr
    # a file without the semantics column loads zeros (semantic_3dgs_wrapper.py:124-128)
    ply.write_gaussian_ply(path, m, with_semantics=False)
    ply.read_gaussian_ply(path, back, device="cpu")
    assert float(back._semantics.abs().sum()) == 0 and back._semantics.shape == (n, 1)
