"""Camera math on the caller side of the rasterizer.

Restates (a) the camera conversion GSWorld performs per frame,
``GSWorldWrapper.cam_maniskill2gs`` (/root/reference/gsworld/mani_skill/utils/wrappers/gs_world_wrapper.py:277-325),
and (b) the view/projection construction of the un-vendored 3DGS python layer it feeds
(``utils/graphics_utils.py`` getWorld2View2 / getProjectionMatrix and ``scene/cameras.py`` Camera;
SURVEY.md Appendix B.1).  Pure numpy/torch host code: runs once per camera, not on the hot path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

ZNEAR = 0.01
ZFAR = 100.0


def get_world2view2(R: np.ndarray, t: np.ndarray, translate=np.array([0.0, 0.0, 0.0]), scale: float = 1.0):
    """graphics_utils.getWorld2View2: ``R`` is stored transposed (caller passes world2cam[:3,:3].T)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    cam_center = C2W[:3, 3]
    cam_center = (cam_center + translate) * scale
    C2W[:3, 3] = cam_center
    Rt = np.linalg.inv(C2W)
    return np.float32(Rt)


def get_projection_matrix(znear: float, zfar: float, fovX: float, fovY: float) -> torch.Tensor:
    """graphics_utils.getProjectionMatrix: symmetric frustum, z mapped to [0,1], w = +z."""
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = torch.zeros(4, 4)
    z_sign = 1.0
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


@dataclass
class ViewParams:
    """What the rasterizer consumes from a 3DGS ``Camera`` (all float32 CPU tensors until ``.to``)."""

    image_width: int
    image_height: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor  # (4,4) = W2C^T
    full_proj_transform: torch.Tensor  # (4,4) = (P W2C)^T
    camera_center: torch.Tensor  # (3,)

    @property
    def tanfovx(self) -> float:
        return math.tan(self.FoVx * 0.5)

    @property
    def tanfovy(self) -> float:
        return math.tan(self.FoVy * 0.5)

    def to(self, device):
        return ViewParams(self.image_width, self.image_height, self.FoVx, self.FoVy,
                          self.world_view_transform.to(device), self.full_proj_transform.to(device),
                          self.camera_center.to(device))


def view_params(R: np.ndarray, T: np.ndarray, FoVx: float, FoVy: float, width: int, height: int,
                trans=np.array([0.0, 0.0, 0.0]), scale: float = 1.0) -> ViewParams:
    """scene/cameras.py Camera.__init__ matrix part (znear=0.01, zfar=100)."""
    wvt = torch.tensor(get_world2view2(R, T, trans, scale)).transpose(0, 1)
    proj = get_projection_matrix(ZNEAR, ZFAR, FoVx, FoVy).transpose(0, 1)
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
    center = wvt.inverse()[3, :3]
    return ViewParams(width, height, float(FoVx), float(FoVy), wvt.contiguous(), full.contiguous(),
                      center.contiguous())


def extract_rigid_transform(M: torch.Tensor):
    """Polar decomposition of a 4x4 affine into rigid 4x4 + uniform scale.

    Mirrors /root/reference/gsworld/utils/pcd_utils.py:224-252 (same return tuple).
    """
    single = M.ndim == 2
    if single:
        if M.shape != (4, 4):
            raise ValueError("Input must be (4,4) or (N,4,4)")
        M = M.unsqueeze(0)
    elif not (M.ndim == 3 and M.shape[1:] == (4, 4)):
        raise ValueError("Input must be (4,4) or (N,4,4)")
    A, t = M[:, :3, :3], M[:, :3, 3]
    U, S, Vh = torch.linalg.svd(A)
    scales = S.mean(dim=1)
    R_rigid = U @ Vh
    M_rigid = torch.eye(4, device=M.device, dtype=M.dtype).repeat(M.shape[0], 1, 1)
    M_rigid[:, :3, :3] = R_rigid
    M_rigid[:, :3, 3] = t
    if single:
        return M_rigid[0], scales[0], R_rigid[0], t[0]
    return M_rigid, scales, R_rigid, t


def cam_maniskill2gs(extrinsic_cv: torch.Tensor, intrinsic_k: torch.Tensor, img_w: int, img_h: int,
                     rigid_sim2real: torch.Tensor, scale_sim2real) -> ViewParams:
    """gs_world_wrapper.py:280-301: OpenCV (3,4) extrinsic + K -> (R, T, FoVx, FoVy) -> 3DGS camera.

    The principal point is ignored (``:293-294``), the camera position is scaled by ``scale_sim2real`` and
    the pose is left-multiplied by ``rigid_sim2real`` (``:297-301``).
    """
    ext = extrinsic_cv.to(torch.float32)
    sim_world2cam = torch.vstack([ext, torch.tensor([0, 0, 0, 1], dtype=ext.dtype)])
    sim_cam2world = torch.linalg.inv(sim_world2cam)
    fx, fy = intrinsic_k[0, 0], intrinsic_k[1, 1]
    fovx = 2 * torch.arctan(img_w / (2 * fx))
    fovy = 2 * torch.arctan(img_h / (2 * fy))
    real_cam2world = sim_cam2world
    real_cam2world[:3, 3] = real_cam2world[:3, 3] * scale_sim2real
    real_world2cam = torch.linalg.inv(rigid_sim2real.to(torch.float32) @ real_cam2world)
    R = real_world2cam[:3, :3].T
    T = real_world2cam[:3, 3]
    return view_params(R.cpu().numpy(), T.cpu().numpy(), float(fovx), float(fovy), img_w, img_h)


def look_at_view(eye, target, up, fov_x: float, fov_y: float, width: int, height: int) -> ViewParams:
    """Convenience: OpenCV-convention (x right, y down, z forward) look-at camera."""
    eye, target, up = (np.asarray(v, dtype=np.float64) for v in (eye, target, up))
    z = target - eye
    z /= np.linalg.norm(z)
    x = np.cross(z, up)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    c2w_R = np.stack([x, y, z], axis=1)  # columns = camera axes in world
    w2c_R = c2w_R.T
    T = -w2c_R @ eye
    return view_params(w2c_R.T, T, fov_x, fov_y, width, height)
