"""Seeded synthetic Gaussian scenes and cameras for the configurations BASELINE.json names.

The real GSWorld assets (``xarm6.ply`` etc., /root/reference/configs/xarm6_align.json:1-22) live in an
external dataset that is not available offline, so every configuration is reproduced synthetically with the
distributions fixed in SURVEY.md section 8d.  Calibration constants below are *data* quoted from
/root/reference/gsworld/constants.py (sim2gs matrices :30-42, intrinsics :514-518, camera poses :520-532);
tests/golden/reference_constants.npz holds the values captured from the reference itself.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

from .camera import ViewParams, cam_maniskill2gs, extract_rigid_transform, look_at_view

# /root/reference/gsworld/constants.py:30-35
SIM2GS_ARM_TRANS = np.array([
    [0.65203872, 0.70075277, 0.03073432, -0.08619287],
    [0.03194594, 0.01225097, -0.95706996, -0.75944751],
    [-0.70069858, 0.65264769, -0.01503433, 0.25320947],
    [0.0, 0.0, 0.0, 1.0]], dtype=np.float32)
# /root/reference/gsworld/constants.py:37-42
SIM2GS_XARM_TRANS = np.array([
    [-0.97002696, 0.2247966, 0.10835464, 0.32787871],
    [0.05080531, 0.60369423, -0.7976206, 0.37823396],
    [-0.24432164, -0.76697216, -0.59605971, 0.45637834],
    [0.0, 0.0, 0.0, 1.0]], dtype=np.float32)
# /root/reference/gsworld/constants.py:514-518
RS_D435I_RGB_K = np.array([
    [606.12145996, 0.0, 318.3548584],
    [0.0, 605.1428833, 242.92498779],
    [0.0, 0.0, 1.0]], dtype=np.float32)
# /root/reference/gsworld/constants.py:520-525
RIGHT2BASE = np.array([
    [-0.025185470710454363, 0.9003537485256276, -0.43442930331751733, 0.8003658631290567],
    [0.9990845637502204, 0.007637667199582072, -0.04209157297821219, 0.014761293894194942],
    [-0.034579279071787865, -0.4350917070636938, -0.8997218903101533, 0.8497237283025128],
    [0.0, 0.0, 0.0, 1.0]], dtype=np.float32)
# /root/reference/gsworld/constants.py:527-532
XARM_RIGHT2BASE = np.array([
    [-0.99815940, 0.02312000, 0.05609515, 0.38209513],
    [-0.00610404, 0.88159275, -0.47197380, 0.40018010],
    [-0.06036488, -0.47144645, -0.87982790, 0.46095666],
    [0.0, 0.0, 0.0, 1.0]], dtype=np.float32)

# scene names of config 4 = /root/reference/configs/*.json
SCENE_NAMES = ["xarm6_align", "xarm6_rot_banana", "xarm6_spoon2board", "fr3_align", "fr3_pnp_box",
               "fr3_pour", "fr3_stack", "fr3_no_objs"]
XARM6_ALIGN_NUM_GAUSSIANS = 1_468_850  # /root/reference/gsworld/utils/pcd_utils.py:68


@dataclass
class RawGaussians:
    """Raw (pre-activation) parameters, laid out as ``Semantic3DGSWrapper.load_ply`` produces them
    (/root/reference/gsworld/mani_skill/utils/wrappers/semantic_3dgs_wrapper.py:151-157)."""

    xyz: torch.Tensor  # (N,3)
    features_dc: torch.Tensor  # (N,1,3)
    features_rest: torch.Tensor  # (N,15,3)
    opacity: torch.Tensor  # (N,1) logits
    scaling: torch.Tensor  # (N,3) log-scales
    rotation: torch.Tensor  # (N,4) un-normalised (r,x,y,z)
    semantics: torch.Tensor | None = None  # (N,1) float labels

    @property
    def num(self) -> int:
        return self.xyz.shape[0]

    def to(self, device):
        return RawGaussians(*[None if t is None else t.to(device) for t in (
            self.xyz, self.features_dc, self.features_rest, self.opacity, self.scaling, self.rotation,
            self.semantics)])

    def activated(self):
        """The per-frame activations of upstream ``render()``: returns
        means3D, shs (N,16,3), opacities (N,1), scales, rotations ready for the rasterizer."""
        shs = torch.cat((self.features_dc, self.features_rest), dim=1).contiguous()
        return (self.xyz.contiguous(), shs, torch.sigmoid(self.opacity), torch.exp(self.scaling),
                torch.nn.functional.normalize(self.rotation))


def _matrix_to_quat_wxyz(R: np.ndarray) -> np.ndarray:
    m = R.astype(np.float64)
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    if tr > 0:
        s = math.sqrt(tr + 1.0) * 2
        q = [0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s]
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = math.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = [(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s]
    elif m[1, 1] > m[2, 2]:
        s = math.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = [(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s]
    else:
        s = math.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = [(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s]
    return np.asarray(q, dtype=np.float64)


def _quat_mul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw), -1)


def _sh(gen, n):
    dc = torch.randn(n, 1, 3, generator=gen) * 0.5
    rest = torch.randn(n, 15, 3, generator=gen) * 0.05
    return dc, rest


def random_scene_camera_frame(n: int = 100_000, seed: int = 0, near_fraction: float = 0.01) -> RawGaussians:
    """Config 1 / config 5 distribution: Gaussians in the camera frame of an identity view,
    means ~ U([-1,1]^2 x [1.5,4.5]); ``near_fraction`` of them at z in (0.05, 0.2) so GSWorld's 0.05f cull
    (vs stock 0.2f) is exercised."""
    gen = torch.Generator().manual_seed(seed)
    xyz = torch.rand(n, 3, generator=gen)
    xyz[:, :2] = xyz[:, :2] * 2 - 1
    xyz[:, 2] = xyz[:, 2] * 3 + 1.5
    n_near = int(n * near_fraction)
    if n_near:
        xyz[:n_near, 2] = 0.05 + torch.rand(n_near, generator=gen) * 0.15
        xyz[:n_near, :2] *= 0.1
    lo, hi = math.log(0.005), math.log(0.05)
    scaling = torch.rand(n, 3, generator=gen) * (hi - lo) + lo
    rotation = torch.randn(n, 4, generator=gen)
    opacity = torch.rand(n, 1, generator=gen) * 6 - 2
    dc, rest = _sh(gen, n)
    return RawGaussians(xyz, dc, rest, opacity, scaling, rotation)


def identity_camera(width: int = 256, height: int = 256, fov_deg: float = 60.0) -> ViewParams:
    from .camera import view_params

    f = math.radians(fov_deg)
    return view_params(np.eye(3), np.zeros(3), f, f, width, height)


def tabletop_scene(name: str = "xarm6_align", n: int = XARM6_ALIGN_NUM_GAUSSIANS, seed: int = 1) -> RawGaussians:
    """Config 2 / 4: a synthetic "xarm6_align-like" table-top scan in the GS (real) frame.

    70 % table/background splats near the plane z~0 (flat: one axis x0.1), 25 % robot/object splats in 20
    clusters above the table, 5 % far floaters; opacity logits from a two-mode mixture; positions and
    orientations mapped sim -> GS by the robot's calibration matrix.
    """
    gen = torch.Generator().manual_seed(seed)
    n_tab = int(n * 0.70)
    n_rob = int(n * 0.25)
    n_flo = n - n_tab - n_rob
    # table / background
    xyz_t = torch.rand(n_tab, 3, generator=gen)
    xyz_t[:, 0] = xyz_t[:, 0] * 2.0 - 0.5
    xyz_t[:, 1] = xyz_t[:, 1] * 2.0 - 1.0
    xyz_t[:, 2] = torch.randn(n_tab, generator=gen) * 0.003
    ls_t = (math.log(0.01) + 0.5 * torch.randn(n_tab, 1, generator=gen)).repeat(1, 3)
    ls_t[:, 2] += math.log(0.1)
    q_t = torch.zeros(n_tab, 4)
    q_t[:, 0] = 1.0
    q_t += 0.05 * torch.randn(n_tab, 4, generator=gen)
    # robot / objects
    centers = torch.rand(20, 3, generator=gen) * torch.tensor([0.8, 0.8, 0.8]) + torch.tensor([0.0, -0.4, 0.0])
    which = torch.randint(0, 20, (n_rob,), generator=gen)
    xyz_r = centers[which] + torch.randn(n_rob, 3, generator=gen) * 0.04
    ls_r = math.log(0.004) + 0.4 * torch.randn(n_rob, 3, generator=gen)
    q_r = torch.randn(n_rob, 4, generator=gen)
    # floaters
    d = torch.randn(n_flo, 3, generator=gen)
    d = d / d.norm(dim=1, keepdim=True)
    xyz_f = d * (2 + 4 * torch.rand(n_flo, 1, generator=gen))
    ls_f = math.log(0.05) + 0.5 * torch.randn(n_flo, 3, generator=gen)
    q_f = torch.randn(n_flo, 4, generator=gen)

    xyz = torch.cat((xyz_t, xyz_r, xyz_f))
    scaling = torch.cat((ls_t, ls_r, ls_f))
    rotation = torch.cat((q_t, q_r, q_f))
    mode = torch.rand(n, 1, generator=gen) < 0.6
    opacity = torch.where(mode, 3 + torch.randn(n, 1, generator=gen), -2 + torch.randn(n, 1, generator=gen))
    dc, rest = _sh(gen, n)
    semantics = torch.cat((torch.zeros(n_tab, 1), (which[:, None] + 1).float(), torch.zeros(n_flo, 1)))
    # interleave the three populations so that index order carries no spatial structure
    perm = torch.randperm(n, generator=gen)
    xyz, scaling, rotation, opacity, dc, rest, semantics = (
        t[perm].contiguous() for t in (xyz, scaling, rotation, opacity, dc, rest, semantics))

    sim2gs = torch.tensor(SIM2GS_XARM_TRANS if name.startswith("xarm") else SIM2GS_ARM_TRANS)
    rigid, scale, R, t = extract_rigid_transform(sim2gs)
    xyz = (xyz @ sim2gs[:3, :3].T + sim2gs[:3, 3]).contiguous()
    qR = torch.tensor(_matrix_to_quat_wxyz(R.numpy()), dtype=torch.float32)
    rotation = _quat_mul(qR.expand_as(rotation), rotation).contiguous()
    scaling = scaling + math.log(float(scale))
    return RawGaussians(xyz, dc, rest, opacity, scaling.contiguous(), rotation, semantics)


def arm_tabletop_scene(link_scan, labels, n: int = XARM6_ALIGN_NUM_GAUSSIANS, seed: int = 1, robot_fraction: float = 0.08):
    """A second table-top surrogate in which the robot LOOKS like a robot: the same table / background and floaters as
    :func:`tabletop_scene` (xarm6_align), but the robot's Gaussians sit ON the xarm6's links -- capsules between consecutive
    link frames of the reference's URDF at the scan pose (``link_scan`` (L,4,4) sim-frame poses and their ``labels``, e.g. from
    :func:`gsworld_amd.rollouts.xarm6_rollout`: ``tests/golden/xarm6_rollout.npz``) -- and two graspable objects stand on the
    table (labels 17, 18).  :func:`tabletop_scene` scatters its 20 part clusters over a 0.8 m cube that fills the sensor
    camera's view, and a forward-kinematics rollout then swings those clusters through the whole frame; here an arm moves the way
    an arm does.  ``robot_fraction`` of the Gaussians belong to the links (a real GSWorld asset: the robot scan is a small part
    of the scene).  Not a BASELINE configuration: a datum beside configs[2]'s surrogate (bench.py ``closed_loop.arm_shaped``)."""
    gen = torch.Generator().manual_seed(seed)
    link_scan = torch.as_tensor(link_scan, dtype=torch.float32)
    L = link_scan.shape[0]
    n_rob = int(n * robot_fraction)
    n_obj = int(n * 0.01)
    n_flo = int(n * 0.03)
    n_tab = n - n_rob - 2 * n_obj - n_flo
    # table / background (as tabletop_scene)
    xyz_t = torch.rand(n_tab, 3, generator=gen)
    xyz_t[:, 0] = xyz_t[:, 0] * 2.0 - 0.5
    xyz_t[:, 1] = xyz_t[:, 1] * 2.0 - 1.0
    xyz_t[:, 2] = torch.randn(n_tab, generator=gen) * 0.003
    ls_t = (math.log(0.01) + 0.5 * torch.randn(n_tab, 1, generator=gen)).repeat(1, 3)
    ls_t[:, 2] += math.log(0.1)
    q_t = torch.zeros(n_tab, 4)
    q_t[:, 0] = 1.0
    q_t += 0.05 * torch.randn(n_tab, 4, generator=gen)
    # robot: link k's Gaussians along the segment from its frame to the next link's, 2.5 cm around it
    org = link_scan[:, :3, 3]
    nxt = torch.cat((org[1:], org[-1:] + torch.tensor([[0.0, 0.0, -0.03]])))
    seg_len = (nxt - org).norm(dim=1) + 0.05
    share = seg_len / seg_len.sum()
    which = torch.multinomial(share, n_rob, replacement=True, generator=gen)
    t = torch.rand(n_rob, 1, generator=gen)
    xyz_r = org[which] * (1.0 - t) + nxt[which] * t + torch.randn(n_rob, 3, generator=gen) * 0.025
    ls_r = math.log(0.004) + 0.4 * torch.randn(n_rob, 3, generator=gen)
    q_r = torch.randn(n_rob, 4, generator=gen)
    lab = torch.as_tensor(labels)
    if lab.dim() == 2:  # (a link with two labels: its Gaussians carry either)
        pick = torch.randint(0, lab.shape[1], (n_rob,), generator=gen)
        lab_r = lab[which, pick]
        lab_r = torch.where(lab_r < 0, lab[which, 0], lab_r)
    else:
        lab_r = lab[which]
    # two objects on the table
    obj_c = torch.tensor([[0.45, -0.15, 0.03], [0.45, 0.15, 0.03]])
    xyz_o = obj_c.repeat_interleave(n_obj, 0) + torch.randn(2 * n_obj, 3, generator=gen) * 0.02
    ls_o = math.log(0.003) + 0.3 * torch.randn(2 * n_obj, 3, generator=gen)
    q_o = torch.randn(2 * n_obj, 4, generator=gen)
    lab_o = torch.tensor([17.0, 18.0]).repeat_interleave(n_obj)
    # floaters
    d = torch.randn(n_flo, 3, generator=gen)
    d = d / d.norm(dim=1, keepdim=True)
    xyz_f = d * (2 + 4 * torch.rand(n_flo, 1, generator=gen))
    ls_f = math.log(0.05) + 0.5 * torch.randn(n_flo, 3, generator=gen)
    q_f = torch.randn(n_flo, 4, generator=gen)

    xyz = torch.cat((xyz_t, xyz_r, xyz_o, xyz_f))
    scaling = torch.cat((ls_t, ls_r, ls_o, ls_f))
    rotation = torch.cat((q_t, q_r, q_o, q_f))
    mode = torch.rand(n, 1, generator=gen) < 0.6
    opacity = torch.where(mode, 3 + torch.randn(n, 1, generator=gen), -2 + torch.randn(n, 1, generator=gen))
    dc, rest = _sh(gen, n)
    semantics = torch.cat((torch.zeros(n_tab), lab_r.float(), lab_o, torch.zeros(n_flo)))[:, None]
    perm = torch.randperm(n, generator=gen)
    xyz, scaling, rotation, opacity, dc, rest, semantics = (
        v[perm].contiguous() for v in (xyz, scaling, rotation, opacity, dc, rest, semantics))
    sim2gs = torch.tensor(SIM2GS_XARM_TRANS)
    rigid, scale, R, _ = extract_rigid_transform(sim2gs)
    xyz = (xyz @ sim2gs[:3, :3].T + sim2gs[:3, 3]).contiguous()
    qR = torch.tensor(_matrix_to_quat_wxyz(R.numpy()), dtype=torch.float32)
    rotation = _quat_mul(qR.expand_as(rotation), rotation).contiguous()
    scaling = scaling + math.log(float(scale))
    return RawGaussians(xyz, dc, rest, opacity, scaling.contiguous(), rotation, semantics)


def sensor_camera(name: str = "xarm6_align", width: int = 640, height: int = 480) -> ViewParams:
    """The ``right_cam`` sensor of the scene's env, pushed through the wrapper's camera conversion.

    Pose = robot root pose (0,0,0.03) (/root/reference/gsworld/mani_skill/envs/tasks/tabletop/xarm6/align.py:181-183)
    composed with the calibrated camera-to-base matrix (real_xarm_env.py:104-110); intrinsics
    ``rs_d435i_rgb_k`` (real_xarm_env.py:111-134).  FoVx = 0.9715089, FoVy = 0.7551448 at 640x480.
    """
    xarm = name.startswith("xarm")
    cam2base = XARM_RIGHT2BASE if xarm else RIGHT2BASE
    root = np.eye(4, dtype=np.float64)
    root[2, 3] = 0.03 if xarm else 0.0
    cam2world = root @ cam2base.astype(np.float64)
    extrinsic_cv = torch.tensor(np.linalg.inv(cam2world)[:3, :4], dtype=torch.float32)
    sim2gs = torch.tensor(SIM2GS_XARM_TRANS if xarm else SIM2GS_ARM_TRANS)
    rigid, scale, _, _ = extract_rigid_transform(sim2gs)
    return cam_maniskill2gs(extrinsic_cv, torch.tensor(RS_D435I_RGB_K), width, height, rigid, scale)


def dense_view_camera(name: str = "xarm6_align", width: int = 640, height: int = 480, height_m: float = 1.5) -> ViewParams:
    """A second view of the same scene in which most of it is on screen: straight down on the table centre from
    ``height_m`` metres (sim frame, mapped by the scene's sim2gs), same intrinsics as the sensor camera.  At 1.5 m the
    xarm6_align-like scene has V = 0.60 N visible Gaussians -- the ratio SURVEY.md 8d's worked example assumes -- against
    0.12 N from ``right_cam``, whose frustum holds only the part of the table in front of the robot."""
    M = (SIM2GS_XARM_TRANS if name.startswith("xarm") else SIM2GS_ARM_TRANS).astype(np.float64)
    to_gs = lambda p: M[:3, :3] @ np.asarray(p, dtype=np.float64) + M[:3, 3]  # noqa: E731
    return look_at_view(to_gs([0.5, 0.0, height_m]), to_gs([0.5, 0.0, 0.0]), M[:3, :3] @ np.array([1.0, 0.0, 0.0]),
                        0.9715089, 0.7551448, width, height)


def training_camera(width: int = 800, height: int = 800, fov_deg: float = 60.0) -> ViewParams:
    return identity_camera(width, height, fov_deg)


__all__ = ["RawGaussians", "random_scene_camera_frame", "identity_camera", "tabletop_scene", "arm_tabletop_scene", "sensor_camera",
           "dense_view_camera",
           "training_camera", "SCENE_NAMES", "XARM6_ALIGN_NUM_GAUSSIANS", "look_at_view"]
