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
``gsworld_amd.transform`` implementation and the rotation outputs are pinned only up to that stub.
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
    from gsworld_amd import transform as mine

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
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
