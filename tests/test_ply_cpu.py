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


def test_merger_reproduces_the_reference_merged_model(tmp_path):
    """gsworld_amd.merger vs the tensors the REFERENCE GaussianModelMerger produced for the same three PLY files and
    config (tests/golden/merger.npz, tools/make_golden.py): label from .npy / number / PLY column, concat order,
    (N,1,1) opacity, ignored "transformation", subset merge in the order of `indices`."""
    import json
    import os

    from gsworld_amd import merger as gm

    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "merger.npz"))
    config = json.loads(bytes(ref["config_json"]).decode())
    attrs = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_semantics")
    for k, entry in enumerate(config["models"]):
        m = types.SimpleNamespace(**{a: torch.from_numpy(ref[f"in{k}.{a}"]) for a in attrs})
        path = tmp_path / entry["data_path"]
        os.makedirs(path.parent, exist_ok=True)
        ply.write_gaussian_ply(str(path), m, with_semantics=(k != 1))
    np.save(tmp_path / "scene" / "robot_semantics.npy", ref["in0.labels_npy"])
    cfg = tmp_path / "scene.json"
    cfg.write_text(json.dumps(config))

    mg = gm.GaussianModelMerger(device="cpu", asset_dir=str(tmp_path))
    assert mg.load_models_from_config(str(cfg)) == [0, 1, 2]
    merged = mg.merge_models()
    sub = mg.merge_models(indices=[2, 0])
    for name, model in (("merged", merged), ("merged_2_0", sub)):
        for a in attrs:
            want = ref[f"{name}.{a}"]
            got = getattr(model, a)
            assert tuple(got.shape) == want.shape, (name, a, got.shape, want.shape)
            assert str(got.dtype).replace("torch.", "") == str(want.dtype), (name, a, got.dtype, want.dtype)
            np.testing.assert_array_equal(got.numpy(), want, err_msg=f"{name}.{a}")
    assert merged._opacity.shape == (14, 1, 1) and merged._semantics.shape == (14, 1)
    assert merged._semantics[:, 0].tolist() == [1, 1, 2, 3, 3, 16, 0] + [201] * 4 + [7.0] * 3
    # one call = gaussian_merger.main(path)
    again = gm.merge_scene(str(cfg), asset_dir=str(tmp_path), device="cpu")
    assert torch.equal(again._xyz, merged._xyz) and again.active_sh_degree == 3
    # error behaviour of the reference
    import pytest
    with pytest.raises(FileNotFoundError):
        mg.load_config_from_json(str(tmp_path / "missing.json"))
    bad = tmp_path / "bad.json"
    bad.write_text("{\"no_models\": 1}")
    with pytest.raises(ValueError):
        mg.load_config_from_json(str(bad))
    bad.write_text("{not json")
    with pytest.raises(ValueError):
        mg.load_config_from_json(str(bad))
    with pytest.raises(FileNotFoundError):
        mg.load_model_from_config({"data_path": "./nope.ply"})
    with pytest.raises(ValueError):
        mg.load_model_from_config({})
    with pytest.raises(IndexError):
        mg.get_model(17)
    with pytest.raises(ValueError):
        gm.GaussianModelMerger(device="cpu").merge_models()
    # saving the merged model keeps the 63-column layout incl. semantics and loads back identically
    out = tmp_path / "out" / "merged.ply"
    assert mg.save_merged_model(str(out))
    back = gm.semantic_model_class()(3)
    back.load_ply(str(out), device="cpu")
    assert torch.equal(back._xyz, sub._xyz) and torch.equal(back._semantics, sub._semantics.float())  # (last merge)


def test_stock_load_ply_resumes_training(tmp_path):
    """The base GaussianModel.load_ply mirrors upstream (trainable nn.Parameters, (N,1) opacity): the stock flow
    load_ply -> training_setup works, unlike GSWorld's frozen semantic loader."""
    import os
    import sys

    compat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gsworld_amd", "gs_compat")
    if compat not in sys.path:
        sys.path.insert(0, compat)
    from scene.gaussian_model import GaussianModel

    gen = torch.Generator().manual_seed(2)
    n = 9
    m = types.SimpleNamespace(
        _xyz=torch.randn(n, 3, generator=gen), _features_dc=torch.randn(n, 1, 3, generator=gen),
        _features_rest=torch.randn(n, 15, 3, generator=gen), _opacity=torch.randn(n, 1, generator=gen),
        _scaling=torch.randn(n, 3, generator=gen), _rotation=torch.randn(n, 4, generator=gen), max_sh_degree=3)
    path = str(tmp_path / "point_cloud.ply")
    ply.write_gaussian_ply(path, m, with_semantics=False)
    if not torch.cuda.is_available():
        real = ply.read_gaussian_ply
        ply.read_gaussian_ply = lambda p, mod, device="cuda", upstream=False: real(p, mod, device="cpu", upstream=upstream)
    try:
        g = GaussianModel(3)
        g.load_ply(path)
    finally:
        if not torch.cuda.is_available():
            ply.read_gaussian_ply = real
    assert g._opacity.shape == (n, 1) and isinstance(g._xyz, torch.nn.Parameter)
    assert all(getattr(g, a).requires_grad for a in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling",
                                                      "_rotation"))
    assert torch.equal(g._features_rest.detach().cpu(), m._features_rest) and g.active_sh_degree == 3
    (g.get_opacity.sum() + g.get_scaling.sum()).backward()
    assert g._opacity.grad is not None and g._scaling.grad is not None


def test_inference_shortcut_only_for_stock_getters():
    """gaussian_renderer.render hands the STORED tensors to the rasterizer when nothing needs grad; a subclass whose
    getters compute something else (deformation, pose / scale optimisation) must not take that shortcut."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for sub in ("gs_compat", "dropin"):
        path = os.path.join(root, "gsworld_amd", sub)
        if path not in sys.path:
            sys.path.insert(0, path)
    import gaussian_renderer as gr
    from scene.gaussian_model import GaussianModel

    class Deformed(GaussianModel):
        @property
        def get_xyz(self):
            return self._xyz + 1.0

    class Labelled(GaussianModel):  # adds state, keeps the getters (GSWorld's semantic model is of this kind)
        def get_semantics(self):
            return None

    assert gr._stock_getters(GaussianModel(3)) and gr._stock_getters(Labelled(3))
    assert not gr._stock_getters(Deformed(3))
    assert not gr._stock_getters(types.SimpleNamespace(_xyz=None))
