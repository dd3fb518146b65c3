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
    gm = merger_golden(consts)
    wrapper_glue_golden(consts, pcd, gsu, gm)
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
    return gm


def _quat_wxyz_to_matrix(q):
    """ManiSkill's ``Pose.to_transformation_matrix`` rotation block (quaternion_to_matrix, real part first) -- ManiSkill
    is not in the reference tree, so this stub is this project's own (the fixture stores the resulting matrices too)."""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


class _FakePose:
    """Stands in for mani_skill.utils.structs.pose.Pose: ``raw_pose`` (E,7) = position + wxyz quaternion."""

    def __init__(self, raw):
        self.raw_pose = raw

    @classmethod
    def create(cls, pose):
        return cls(pose)

    def to_transformation_matrix(self):
        E = self.raw_pose.shape[0]
        M = torch.eye(4).repeat(E, 1, 1)  # a FRESH tensor per call (the wrapper adds offsets in place, :117-119)
        M[:, :3, :3] = _quat_wxyz_to_matrix(self.raw_pose[:, 3:7])
        M[:, :3, 3] = self.raw_pose[:, :3]
        return M


def wrapper_glue_golden(consts, pcd, gsu, gm):
    """tests/golden/wrapper_glue.npz: the reference's OWN ``GSWorldWrapper.transform_gs_perlink`` (:110-162),
    ``_render_gsworld`` (:232-275) and ``cam_maniskill2gs`` (:277-325), run unmodified on a fake simulator.

    ``gs_world_wrapper.py`` is imported by file path.  Stubs around it: ``gymnasium.Wrapper`` (holds ``env``, forwards
    ``unwrapped``), ``mani_skill.envs.sapien_env.BaseEnv`` / ``structs.Actor`` / ``Link`` (type annotations only),
    ``structs.pose.Pose`` (``_FakePose``: position + wxyz quaternion -> 4x4, ManiSkill's formula), the 3DGS python layer by
    gs_compat (as GSWorld resolves it through GS_DIR) and ``device="cuda"`` (hard-coded at :235) redirected to the CPU.
    The wrapper object is made with ``object.__new__`` (its ``__init__`` needs SAPIEN) and given exactly the attributes
    ``__init__`` would have computed, through the reference's own ``extract_rigid_transform``.  Two hooks record what
    the reference code computes on the way: ``transform_gaussians`` (the arguments of every call, then the real
    function) and ``render`` (the ``gs4render`` tensors and the ``Camera`` it is handed; returns a black frame)."""
    import copy

    sys.modules["gsworld.utils"] = types.ModuleType("gsworld.utils")
    sys.modules["gsworld.utils.pcd_utils"] = pcd
    sys.modules["gsworld.utils.gs_utils"] = gsu
    sys.modules["gsworld.utils.gaussian_merger"] = gm
    gym = types.ModuleType("gymnasium")

    class Wrapper:
        def __init__(self, env):
            self.env = env

        @property
        def unwrapped(self):
            return self.env.unwrapped

    gym.Wrapper, gym.Env = Wrapper, object
    sys.modules["gymnasium"] = gym
    for name in ("mani_skill.envs", "mani_skill.envs.sapien_env", "mani_skill.utils.structs",
                 "mani_skill.utils.structs.pose"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["mani_skill.envs.sapien_env"].BaseEnv = object
    sys.modules["mani_skill.utils.structs"].Actor = type("Actor", (), {})
    sys.modules["mani_skill.utils.structs"].Link = type("Link", (), {})
    sys.modules["mani_skill.utils.structs.pose"].Pose = _FakePose
    gw = load("ref_gs_world_wrapper", f"{REF}/gsworld/mani_skill/utils/wrappers/gs_world_wrapper.py")

    gen = torch.Generator().manual_seed(23)

    def rigid(n, angle, shift):
        q = torch.randn(n, 4, generator=gen) * angle
        q[:, 0] = 1.0
        q = q / q.norm(dim=1, keepdim=True)
        return torch.cat((torch.randn(n, 3, generator=gen) * shift, q), 1)  # (n,7) position + wxyz

    link_names = list(consts.xarm_gs_semantics.keys())  # world, link_base, link1 .. right_finger (:453-471)
    actor_names = ["005_tomato_soup_can", "dtc_green_can", "table-workspace"]  # the last one is not tracked (:137)
    # labels: -1 background, 0..16 the links (link6 = [7, 8]), 110 / 201 the tracked actors, 50 something unlabelled.
    # Part sizes of 1 and 3 are on purpose: the write-back test `value.shape[0] == num_envs` (:246-265) also fires for
    # a part with exactly num_envs Gaussians (then for scaling and opacity too)
    counts = {-1: 260, 0: 12, 50: 9, 110: 40, 201: 37}
    counts.update({k: 24 + k for k in range(1, 17)})
    counts[15], counts[16] = 3, 1
    labels = torch.cat([torch.full((n,), float(lab)) for lab, n in counts.items()])
    labels = labels[torch.randperm(labels.numel(), generator=gen)]
    N = labels.numel()
    # positions: the robot's workspace in the simulator frame, mapped into the scan frame (what a merged scene holds),
    # so that both sensor cameras see the model when the GPU test renders it
    ws = torch.rand(N, 3, generator=gen) * torch.tensor([0.7, 0.7, 0.45]) + torch.tensor([0.05, -0.35, 0.0])
    A = torch.tensor(consts.sim2gs_xarm_trans)
    ws = (ws @ A[:3, :3].T + A[:3, 3]).contiguous()
    for n in actor_names[:2]:  # tracked objects: a 10 cm blob in the object's own scan frame (sim2gs_object_transforms)
        rows = labels == float(consts.obj_gs_semantics[n])
        B = torch.tensor(consts.sim2gs_object_transforms[n])
        loc = (torch.rand(int(rows.sum()), 3, generator=gen) - 0.5) * 0.1
        ws[rows] = loc @ B[:3, :3].T + B[:3, 3]
    model = types.SimpleNamespace(
        _xyz=ws, _features_dc=torch.randn(N, 1, 3, generator=gen) * 0.5,
        _features_rest=torch.randn(N, 15, 3, generator=gen) * 0.05, _opacity=torch.randn(N, 1, 1, generator=gen) + 1.0,
        _scaling=torch.randn(N, 3, generator=gen) * 0.4 - 4.5, _rotation=torch.randn(N, 4, generator=gen),
        _semantics=labels.reshape(N, 1))
    out = {f"model.{a}": getattr(model, a).numpy() for a in
           ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_semantics")}
    out["link_names"] = np.array(link_names)
    out["actor_names"] = np.array(actor_names)
    scan = rigid(len(link_names), 0.4, 0.3)  # the robot's link poses when it was scanned (__init__ :94-103)
    out["link_scan_pose"] = scan.numpy()
    out["link_scan_matrix"] = _FakePose(scan).to_transformation_matrix().numpy()
    K = torch.tensor(consts.rs_d435i_rgb_k)
    W, H = 640, 480

    real_tensor = torch.tensor

    def tensor_on_cpu(*a, **kw):
        if kw.get("device", None) == "cuda":
            kw["device"] = "cpu"
        return real_tensor(*a, **kw)

    for E in (1, 3):
        now = (scan[None] + rigid(E * len(link_names), 0.08, 0.03).reshape(E, len(link_names), 7) * torch.tensor([1, 1, 1, 0, 1, 1, 1.0])).contiguous()
        actors = torch.cat((rigid(E * len(actor_names), 0.5, 0.0),
                            torch.randn(E * len(actor_names), 6, generator=gen)), 1).reshape(E, len(actor_names), 13)
        actors[:, :, :3] = torch.rand(E, len(actor_names), 3, generator=gen) * torch.tensor([0.4, 0.4, 0.1]) + \
            torch.tensor([0.2, -0.2, 0.03])  # somewhere on the table
        cam2world = {"right_cam": torch.tensor(consts.xarm_right2base), "wrist_cam": torch.tensor(consts.xarm_wrist2base)}
        cam2world["right_cam"][2, 3] += 0.03  # robot root pose (0, 0, 0.03), align.py:181-183
        extr = {k: torch.linalg.inv(v)[:3, :4].repeat(E, 1, 1) for k, v in cam2world.items()}

        class FakeLink:
            def __init__(self, name, raw):
                self.name, self.pose = name, _FakePose(raw)

        robot = types.SimpleNamespace(get_links=lambda: [FakeLink(n, now[:, k]) for k, n in enumerate(link_names)],
                                      name="xarm6_uf_gripper")
        base = types.SimpleNamespace(
            agent=types.SimpleNamespace(robot=robot, uid="xarm6_uf_gripper"),
            get_state_dict=lambda: {"actors": {n: actors[:, k] for k, n in enumerate(actor_names)}},
            get_sensor_params=lambda: {k: {"extrinsic_cv": extr[k], "intrinsic_cv": K.repeat(E, 1, 1)} for k in extr},
            get_sensor_images=lambda: {k: {"rgb": torch.zeros(E, H, W, 3, dtype=torch.uint8)} for k in extr})
        base.unwrapped = base
        base.base_env = base
        w = object.__new__(gw.GSWorldWrapper)
        w.env = base
        import argparse

        w.num_envs, w.device, w.robot_pipe = E, "cpu", gsu.PipelineParams(argparse.ArgumentParser())
        w.gs_semantics = dict(consts.xarm_gs_semantics)  # (:53; a copy: transform_gs_perlink adds the actors, :162)
        w.sim2gs_arm_trans = torch.tensor(consts.sim2gs_xarm_trans, dtype=torch.float32)  # (:54, :65)
        w.obj_gs_semantics = consts.obj_gs_semantics
        w.rigid_sim2real, w.scale_sim2real, _, _ = pcd.extract_rigid_transform(w.sim2gs_arm_trans)  # (:70)
        w.initial_merger_robot = model
        w.gs_link_pose_mats = [_FakePose(scan[k:k + 1]).to_transformation_matrix() for k in range(len(link_names))]
        w.gs_movable_pts = dict()

        calls, renders = [], []
        real_tg = gsu.transform_gaussians

        def tg_hook(gaussians, selected_indices, scale, rot_mat, translation, new_opacity):
            calls.append(dict(selected=selected_indices.clone(), scale=None if scale is None else scale.clone(),
                              rot_mat=rot_mat.clone(), translation=translation.clone()))
            return real_tg(gaussians, selected_indices=selected_indices, scale=scale, rot_mat=rot_mat,
                           translation=translation, new_opacity=new_opacity)

        def render_hook(cam, gs, pipe, bg, use_trained_exp=False, separate_sh=False):
            renders.append((cam, copy.deepcopy(gs), bg.clone(), separate_sh))
            return {"render": torch.zeros(3, cam.image_height, cam.image_width)}

        gw.transform_gaussians, gw.render = tg_hook, render_hook
        torch.tensor = tensor_on_cpu
        try:
            w.transform_gs_perlink(w.initial_merger_robot)
            frames = w._render_gsworld()
        finally:
            torch.tensor = real_tensor
            gw.transform_gaussians = real_tg
        p = f"E{E}."
        out[p + "link_pose"] = now.numpy()
        out[p + "actor_state"] = actors.numpy()
        out[p + "link_pose_matrix"] = torch.stack(
            [_FakePose(now[:, k]).to_transformation_matrix() for k in range(len(link_names))], 1).numpy()
        out[p + "actor_pose_matrix"] = torch.stack(
            [_FakePose(actors[:, k, :7]).to_transformation_matrix() for k in range(len(actor_names))], 1).numpy()
        part_names = list(w.gs_movable_pts.keys())
        assert part_names == link_names + actor_names[:2] and len(calls) == len(part_names)
        out[p + "part_names"] = np.array(part_names)
        for name, c in zip(part_names, calls):
            out[p + f"call.{name}.selected"] = c["selected"].numpy()
            out[p + f"call.{name}.rot_mat"] = c["rot_mat"].numpy()
            out[p + f"call.{name}.translation"] = c["translation"].numpy()
            if c["scale"] is not None:
                out[p + f"call.{name}.scale"] = c["scale"].numpy()
            for attr, v in zip(("xyz", "scaling", "rotation", "opacity"), w.gs_movable_pts[name]):
                out[p + f"moved.{name}.{attr}"] = v.numpy()
        assert len(renders) == 2 * E and list(frames.keys()) == ["right_cam", "wrist_cam"]
        assert all(f.shape == (E, H, W, 3) and f.dtype == torch.uint8 for f in frames.values())
        for ci, cname in enumerate(frames):
            cam = renders[ci * E][0]
            out[p + f"cam.{cname}.extrinsic_cv"] = extr[cname][0].numpy()
            out[p + f"cam.{cname}.R"] = np.asarray(cam.R)
            out[p + f"cam.{cname}.T"] = np.asarray(cam.T)
            out[p + f"cam.{cname}.fov"] = np.array([cam.FoVx, cam.FoVy], dtype=np.float64)
            out[p + f"cam.{cname}.size"] = np.array([cam.image_width, cam.image_height])
            for e in range(E):
                cam_e, gs, bg, sep = renders[ci * E + e]
                assert sep is False and float(bg.abs().max()) == 0.0 and cam_e.image_name == cname
                for a in ("_features_dc", "_features_rest", "_semantics"):
                    assert torch.equal(getattr(gs, a), getattr(model, a))  # never written back
                for a in ("_xyz", "_scaling", "_rotation", "_opacity"):
                    key = p + f"gs4render.env{e}.{a}"
                    if ci == 0:
                        out[key] = getattr(gs, a).numpy()
                    else:  # the second camera is handed the same model (only the camera differs)
                        assert np.array_equal(out[key], getattr(gs, a).numpy())
    out["intrinsic_k"] = K.numpy()
    out["sim2gs_arm_trans"] = np.asarray(consts.sim2gs_xarm_trans, dtype=np.float32)
    out["rigid_sim2real"] = w.rigid_sim2real.numpy()
    out["scale_sim2real"] = np.asarray(w.scale_sim2real.numpy())
    out["object_offset.xarm_arm"] = np.array(consts.object_offset["xarm_arm"], dtype=np.float32)
    for n in actor_names[:2]:
        out[f"object_offset.{n}"] = np.array(consts.object_offset[n], dtype=np.float32)
        out[f"object_scale.{n}"] = np.array(consts.object_scale[n], dtype=np.float32)
        out[f"sim2gs_object.{n}"] = np.asarray(consts.sim2gs_object_transforms[n], dtype=np.float32)
        out[f"label.{n}"] = np.array(consts.obj_gs_semantics[n])
    np.savez_compressed(os.path.join(OUT, "wrapper_glue.npz"), **out)


if __name__ == "__main__":
    main()
