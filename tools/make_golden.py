#!/usr/bin/env python
"""Generates tests/golden/*.npz by IMPORTING the reference's caller-side Python (runs only in the authoring
container, where /root/reference exists; the fixtures -- inputs and expected outputs, data only -- are committed
and travel to the GPU box).  Run with:  python -B tools/make_golden.py

Imported, by file path (``gsworld/__init__.py`` needs ManiSkill):
  /root/reference/gsworld/constants.py                      calibration matrices, intrinsics, semantic maps
  /root/reference/gsworld/utils/pcd_utils.py                extract_rigid_transform (:224-252)
  /root/reference/gsworld/utils/gs_utils.py                 transform_gaussians (:283-385), inverse_sigmoid (:169)
Stubs: open3d / plyfile / cv2 (import-only) and ``mani_skill.utils.geometry.rotation_conversions`` -- the latter
is PyTorch3D-derived code that is NOT in the reference tree, so the stub is this project's own
``oracle.transform_ref`` implementation and the rotation outputs are pinned only up to that stub.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    from oracle import transform_ref as mine  # the rotation_conversions stub (checker code)

    os.makedirs(OUT, exist_ok=True)
    consts = load("ref_constants", f"{REF}/gsworld/constants.py")
    np.savez(
        os.path.join(OUT, "reference_constants.npz"),
        sim2gs_arm_trans=consts.sim2gs_arm_trans, sim2gs_xarm_trans=consts.sim2gs_xarm_trans,
        sim2gs_r1_trans=consts.sim2gs_r1_trans, rs_d435i_rgb_k=consts.rs_d435i_rgb_k,
        right2base=consts.right2base, xarm_right2base=consts.xarm_right2base,
        xarm_wrist2base=consts.xarm_wrist2base,
        xarm_semantic_names=np.array(sorted(consts.xarm_gs_semantics.keys())),
        xarm_semantic_ids=np.array([np.atleast_1d(consts.xarm_gs_semantics[k])[0]
                                    for k in sorted(consts.xarm_gs_semantics.keys())]),
    )

    for name in ("open3d", "plyfile", "cv2"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["plyfile"].PlyData = object
    sys.modules["plyfile"].PlyElement = object
    sys.modules["cv2"].aruco = types.SimpleNamespace()
    pcd = load("ref_pcd_utils", f"{REF}/gsworld/utils/pcd_utils.py")
    gen = torch.Generator().manual_seed(0)
    mats = [torch.tensor(consts.sim2gs_arm_trans), torch.tensor(consts.sim2gs_xarm_trans),
            torch.tensor(consts.sim2gs_r1_trans)]
    rand = torch.eye(4).repeat(8, 1, 1)
    rand[:, :3, :] = torch.randn(8, 3, 4, generator=gen)
    single = [pcd.extract_rigid_transform(m) for m in mats]
    batch = pcd.extract_rigid_transform(rand)
    np.savez(
        os.path.join(OUT, "extract_rigid_transform.npz"),
        single_in=torch.stack(mats).numpy(),
        single_rigid=torch.stack([s[0] for s in single]).numpy(),
        single_scale=torch.stack([s[1] for s in single]).numpy(),
        batch_in=rand.numpy(), batch_rigid=batch[0].numpy(), batch_scale=batch[1].numpy(),
        batch_R=batch[2].numpy(), batch_t=batch[3].numpy(),
    )

    # transform_gaussians with the rotation_conversions stub
    for modname in ("mani_skill", "mani_skill.utils", "mani_skill.utils.geometry",
                    "mani_skill.utils.geometry.rotation_conversions"):
        sys.modules.setdefault(modname, types.ModuleType(modname))
    rc = sys.modules["mani_skill.utils.geometry.rotation_conversions"]
    rc.matrix_to_quaternion = mine.matrix_to_quaternion
    rc.quaternion_multiply = mine.quaternion_multiply
    gsu = load("ref_gs_utils", f"{REF}/gsworld/utils/gs_utils.py")

    N = 1000
    g = types.SimpleNamespace(
        _xyz=torch.randn(N, 3, generator=gen), _scaling=torch.randn(N, 3, generator=gen) * 0.5 - 4.0,
        _rotation=torch.randn(N, 4, generator=gen), _opacity=torch.randn(N, 1, generator=gen))
    sel = torch.randperm(N, generator=gen)[:300].sort().values

    def rot(b):
        q = torch.randn(b, 4, generator=gen)
        q = q / q.norm(dim=1, keepdim=True)
        w, x, y, z = q.unbind(1)
        return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1).reshape(b, 3, 3)

    cases = {
        # GSWorldWrapper link transform, num_envs = 1: rot (1,3,3), translation (1,3)   (gs_world_wrapper.py:122-129)
        "link_env1": dict(scale=None, rot_mat=rot(1), translation=torch.randn(1, 3, generator=gen)),
        # num_envs = 3
        "link_env3": dict(scale=None, rot_mat=rot(3), translation=torch.randn(3, 3, generator=gen)),
        # tracked actor: scalar scale, rot (1,3,3), translation (1,3)                     (gs_world_wrapper.py:150-157)
        "actor_env1": dict(scale=torch.tensor(1.07), rot_mat=rot(1), translation=torch.randn(1, 3, generator=gen)),
        "actor_env2": dict(scale=torch.tensor([0.9, 1.2]), rot_mat=rot(2), translation=torch.randn(2, 3, generator=gen)),
        "translate_vec3": dict(scale=None, rot_mat=None, translation=torch.randn(3, generator=gen)),
    }
    out = dict(xyz=g._xyz.numpy(), scaling=g._scaling.numpy(), rotation=g._rotation.numpy(),
               opacity=g._opacity.numpy(), selected=sel.numpy())
    for name, kw in cases.items():
        res = gsu.transform_gaussians(g, sel, new_opacity=None, **kw)
        for k, v in kw.items():
            if v is not None:
                out[f"{name}.in.{k}"] = v.numpy()
        for k, v in zip(("xyz", "scaling", "rotation", "opacity"), res):
            out[f"{name}.out.{k}"] = v.numpy()
    out["inverse_sigmoid.in"] = np.linspace(0.01, 0.99, 50, dtype=np.float32)
    out["inverse_sigmoid.out"] = gsu.inverse_sigmoid(torch.from_numpy(out["inverse_sigmoid.in"])).numpy()
    np.savez(os.path.join(OUT, "transform_gaussians.npz"), **out)
    merger_golden(consts)
    print("wrote", sorted(os.listdir(OUT)))


def merger_golden(consts):
    """tests/golden/merger.npz: the reference's GaussianModelMerger (gaussian_merger.py) + Semantic3DGSWrapper.load_ply
    (semantic_3dgs_wrapper.py:100-167) run on a tiny PLY pair.  The reference classes are imported unmodified; what is
    stubbed around them: ``plyfile`` (absent here) by a reader over this project's PLY parser, the 3DGS python layer by
    gs_compat (as GSWorld resolves it through GS_DIR), ``gsworld.constants.ASSET_DIR`` by a temp directory, and
    ``device="cuda"`` (hard-coded in load_ply, no GPU in this container) is redirected to the CPU while it runs."""
    import json
    import tempfile

    from gsworld_amd import ply

    sys.path.insert(0, os.path.join(ROOT, "gsworld_amd", "gs_compat"))
    sys.path.insert(0, os.path.join(ROOT, "gsworld_amd", "dropin"))

    class _Prop:
        def __init__(self, name):
            self.name = name

    class _Element:
        def __init__(self, cols):
            self._cols = cols
            self.properties = [_Prop(k) for k in cols]

        def __getitem__(self, k):
            return self._cols[k]  # KeyError for a missing column, as plyfile raises (load_ply's bare except)

    class _PlyData:
        def __init__(self, cols):
            self.elements = [_Element(cols)]

        @staticmethod
        def read(path):
            return _PlyData(ply.read_ply(path))

    sys.modules["plyfile"].PlyData = _PlyData
    tmp = tempfile.mkdtemp(prefix="gsworld_golden_")
    consts.ASSET_DIR = tmp
    pkg = types.ModuleType("gsworld")
    pkg.__path__ = []
    sys.modules["gsworld"] = pkg
    sys.modules["gsworld.constants"] = consts
    sem = load("ref_semantic_3dgs_wrapper", f"{REF}/gsworld/mani_skill/utils/wrappers/semantic_3dgs_wrapper.py")
    for name in ("gsworld.mani_skill", "gsworld.mani_skill.utils", "gsworld.mani_skill.utils.wrappers"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["gsworld.mani_skill.utils.wrappers"].Semantic3DGSWrapper = sem.Semantic3DGSWrapper
    gm = load("ref_gaussian_merger", f"{REF}/gsworld/utils/gaussian_merger.py")

    gen = torch.Generator().manual_seed(11)
    sizes = {"scene/robot.ply": 7, "objs/can.ply": 4, "objs/cup.ply": 3}
    out = {}
    for k, (rel, n) in enumerate(sizes.items()):
        m = types.SimpleNamespace(
            _xyz=torch.randn(n, 3, generator=gen), _features_dc=torch.randn(n, 1, 3, generator=gen),
            _features_rest=torch.randn(n, 15, 3, generator=gen), _opacity=torch.randn(n, 1, 1, generator=gen),
            _scaling=torch.randn(n, 3, generator=gen), _rotation=torch.randn(n, 4, generator=gen),
            _semantics=torch.full((n, 1), 5.0 + k))
        os.makedirs(os.path.dirname(os.path.join(tmp, rel)), exist_ok=True)
        # the robot scan carries a semantics column, the first object none, the second one that the config overrides
        ply.write_gaussian_ply(os.path.join(tmp, rel), m, with_semantics=(k != 1))
        for a in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_semantics"):
            out[f"in{k}.{a}"] = getattr(m, a).numpy()
    labels = np.array([1, 1, 2, 3, 3, 16, 0], dtype=np.int64)  # per-point link labels of the robot scan
    np.save(os.path.join(tmp, "scene/robot_semantics.npy"), labels)
    out["in0.labels_npy"] = labels
    config = {"models": [
        {"data_path": "./scene/robot.ply", "semantic_labels": "./scene/robot_semantics.npy", "transformation": []},
        {"data_path": "./objs/can.ply", "semantic_labels": 201, "transformation": []},
        {"data_path": "./objs/cup.ply", "transformation": [1, 0, 0, 5, 0, 1, 0, 5, 0, 0, 1, 5, 0, 0, 0, 1]}]}
    with open(os.path.join(tmp, "scene.json"), "w") as f:
        json.dump(config, f)
    out["config_json"] = np.frombuffer(json.dumps(config).encode(), dtype=np.uint8)

    real_tensor, real_zeros = torch.tensor, torch.zeros

    def on_cpu(fn):
        def wrapped(*a, **kw):
            if kw.get("device", None) == "cuda":
                kw["device"] = "cpu"
            return fn(*a, **kw)
        return wrapped

    torch.tensor, torch.zeros = on_cpu(real_tensor), on_cpu(real_zeros)
    try:
        merger = gm.GaussianModelMerger(device="cpu")
        merger.load_models_from_config(os.path.join(tmp, "scene.json"))
        merged = merger.merge_models()
        sub = merger.merge_models(indices=[2, 0])
    finally:
        torch.tensor, torch.zeros = real_tensor, real_zeros
    for name, mm in (("merged", merged), ("merged_2_0", sub)):
        for a in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_semantics"):
            out[f"{name}.{a}"] = getattr(mm, a).detach().numpy()
    np.savez(os.path.join(OUT, "merger.npz"), **out)


if __name__ == "__main__":
    main()
