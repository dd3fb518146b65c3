"""Per-step rigid transform of labelled Gaussians -- the step immediately before the rasterizer in GSWorld
(SURVEY.md 8f-1).

Host-side mirror of ``transform_gaussians`` (/root/reference/gsworld/utils/gs_utils.py:283-385: scale -> rotate
-> translate -> opacity on a selected index set, with its exact output shapes) and of the two PyTorch3D-derived
helpers it imports from ManiSkill (``matrix_to_quaternion``, ``quaternion_multiply``; real-first ``wxyz``).
The fused HIP operator that replaces the wrapper's per-link mask / gather / scatter passes
(gs_world_wrapper.py:110-162, 244-265) lives in ``gsworld_amd/csrc/transform.hip`` and is checked against this
file's semantics.
"""
from __future__ import annotations

import torch


def _sqrt_positive_part(x: torch.Tensor) -> torch.Tensor:
    ret = torch.zeros_like(x)
    pos = x > 0
    ret[pos] = torch.sqrt(x[pos])
    return ret


def standardize_quaternion(q: torch.Tensor) -> torch.Tensor:
    return torch.where(q[..., 0:1] < 0, -q, q)


def matrix_to_quaternion(matrix: torch.Tensor) -> torch.Tensor:
    """Rotation matrices (...,3,3) -> quaternions (...,4), real part first, real part >= 0."""
    if matrix.size(-1) != 3 or matrix.size(-2) != 3:
        raise ValueError(f"Invalid rotation matrix shape {matrix.shape}.")
    batch = matrix.shape[:-2]
    m = matrix.reshape(batch + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(m, dim=-1)
    q_abs = _sqrt_positive_part(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22,
                                             1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1))
    # candidates: each row is the quaternion multiplied by one of r, i, j, k
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1)], dim=-2)
    floor = torch.tensor(0.1).to(dtype=q_abs.dtype, device=q_abs.device)
    cand = cand / (2.0 * q_abs[..., None].max(floor))
    # pick the best-conditioned candidate (largest denominator)
    best = torch.nn.functional.one_hot(q_abs.argmax(dim=-1), num_classes=4) > 0.5
    out = cand[best, :].reshape(batch + (4,))
    return standardize_quaternion(out)


def quaternion_raw_multiply(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    ow = aw * bw - ax * bx - ay * by - az * bz
    ox = aw * bx + ax * bw + ay * bz - az * by
    oy = aw * by - ax * bz + ay * bw + az * bx
    oz = aw * bz + ax * by - ay * bx + az * bw
    return torch.stack((ow, ox, oy, oz), -1)


def quaternion_multiply(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Hamilton product, standardised to a non-negative real part."""
    return standardize_quaternion(quaternion_raw_multiply(a, b))


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def _compose_rotation(quat_r: torch.Tensor, r: torch.Tensor) -> torch.Tensor:
    """gs_utils.py:242-249: rotate a (possibly un-normalised) Gaussian quaternion, keeping its norm."""
    norm = r.norm(dim=-1, keepdim=True)
    return quaternion_multiply(quat_r, r / norm) * norm


class FusedPartTransform:
    """One-pass replacement of ``GSWorldWrapper.transform_gs_perlink`` + the write-back loop of
    ``_render_gsworld`` (gs_world_wrapper.py:110-162, 244-265) for ``num_envs = 1``.

    ``part_labels`` maps a part name (robot link or tracked actor) to its semantic label(s), as
    ``xarm_gs_semantics`` / ``obj_gs_semantics`` do (/root/reference/gsworld/constants.py:402-505).  Per step the
    caller passes one 4x4 per part -- for a link ``sim2gs @ link_now @ inv(link_scan) @ inv(sim2gs)``
    (gs_world_wrapper.py:120), for an actor the rigid part of ``sim2gs @ pose @ inv(sim2gs_obj)`` plus its uniform
    scale (``:146-156``) -- and gets the transformed ``xyz`` / ``rotation`` buffers the rasterizer should read.
    What the wrapper writes back at ``num_envs = 1`` is exactly xyz and rotation (scaling and opacity keep their
    shapes and fail its ``shape[0] == num_envs`` test), which is what this op produces.
    """

    def __init__(self, part_labels: dict, semantics: torch.Tensor, lut_size: int = 2048):
        self.names = list(part_labels.keys())
        lut = torch.full((lut_size,), -1, dtype=torch.int32)
        for k, name in enumerate(self.names):
            labels = part_labels[name]
            for lab in (labels if isinstance(labels, (list, tuple)) else [labels]):
                if not 0 <= int(lab) < lut_size:
                    raise ValueError(f"label {lab} of part {name!r} outside the LUT (size {lut_size})")
                lut[int(lab)] = k
        self.device = semantics.device
        self.lut = lut.to(self.device)
        self.semantics = semantics.reshape(-1).to(torch.float32).contiguous()
        self._xyz_out = None
        self._rot_out = None
        self._table = None

    def pack(self, matrices: torch.Tensor, scales: torch.Tensor | None = None) -> torch.Tensor:
        """(K,4,4) rigid matrices (+ optional (K,) uniform scales) -> (K,17) transform table (host math, K ~ 18)."""
        M = matrices.detach().to("cpu", torch.float32)
        K = M.shape[0]
        if K != len(self.names):
            raise ValueError(f"expected {len(self.names)} matrices, got {K}")
        q = matrix_to_quaternion(M[:, :3, :3])
        s = torch.ones(K) if scales is None else scales.detach().to("cpu", torch.float32).reshape(K)
        return torch.cat((M[:, :3, :3].reshape(K, 9), M[:, :3, 3], s[:, None], q), dim=1).contiguous()

    def pack_on_device(self, matrices: torch.Tensor, scales: torch.Tensor | None = None) -> torch.Tensor:
        """Same table as :meth:`pack`, built by ``gsr_pack_part_transforms`` from DEVICE matrices: no host
        round trip, hipGraph-capturable (the table buffer is persistent)."""
        import ctypes as C

        from ._lib import check, lib

        L = lib()
        if not getattr(L, "_xfp_bound", False):
            L.gsr_pack_part_transforms.restype = C.c_int
            L.gsr_pack_part_transforms.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            L._xfp_bound = True
        K = len(self.names)
        if tuple(matrices.shape) != (K, 4, 4) or matrices.dtype != torch.float32 or not matrices.is_contiguous():
            raise ValueError(f"expected contiguous float32 ({K},4,4) matrices, got {tuple(matrices.shape)}")
        if scales is not None and (tuple(scales.shape) != (K,) or scales.dtype != torch.float32):
            raise ValueError(f"expected float32 ({K},) scales")
        if self._table is None:
            self._table = torch.empty((K, 17), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(L.gsr_pack_part_transforms(
                K, C.c_void_p(matrices.data_ptr()), C.c_void_p(scales.data_ptr() if scales is not None else 0),
                C.c_void_p(self._table.data_ptr()), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return self._table

    def apply(self, xyz: torch.Tensor, rotation: torch.Tensor, matrices: torch.Tensor, scales=None):
        """-> (xyz', rotation') in persistent output buffers (overwritten by the next call).  ``matrices`` (K,4,4)
        (+ ``scales`` (K,)) may live on the host (packed with torch, one small H2D copy) or on the device (packed by a
        kernel: no synchronisation, capturable)."""
        import ctypes as C

        from ._lib import check, lib

        if not xyz.is_cuda:
            raise RuntimeError("FusedPartTransform.apply: tensors must live on a HIP device (no CPU path)")
        L = lib()
        if not getattr(L, "_xf_bound", False):
            L.gsr_transform_gaussians.restype = C.c_int
            L.gsr_transform_gaussians.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                                  C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
            L._xf_bound = True
        xyz = xyz.detach().to(torch.float32).contiguous()
        rotation = rotation.detach().to(torch.float32).contiguous()
        P = xyz.shape[0]
        if self._xyz_out is None or self._xyz_out.shape[0] != P:
            self._xyz_out = torch.empty_like(xyz)
            self._rot_out = torch.empty_like(rotation)
        if matrices.is_cuda:
            table = self.pack_on_device(matrices, scales)
        else:
            table = self.pack(matrices, scales).to(self.device, non_blocking=True)
        with torch.cuda.device(self.device):
            check(L.gsr_transform_gaussians(
                P, C.c_void_p(xyz.data_ptr()), C.c_void_p(rotation.data_ptr()), C.c_void_p(self.semantics.data_ptr()),
                C.c_void_p(self.lut.data_ptr()), self.lut.numel(), C.c_void_p(table.data_ptr()), table.shape[0],
                C.c_void_p(self._xyz_out.data_ptr()), C.c_void_p(self._rot_out.data_ptr()),
                C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return self._xyz_out, self._rot_out


def transform_gaussians(gaussians, selected_indices, scale=None, rot_mat=None, translation=None, new_opacity=None):
    """Same contract as the reference function (gs_utils.py:283-385), including its output shapes:
    one rotation ``(1,3,3)`` keeps xyz ``(N,3)`` but yields rotations ``(1,N,4)``; a ``(B,3)`` translation
    promotes xyz to ``(B,N,3)`` -- the shapes GSWorldWrapper's ``shape[0] == num_envs`` tests rely on
    (gs_world_wrapper.py:246-265)."""
    xyz = gaussians._xyz[selected_indices]
    scaling = gaussians._scaling[selected_indices]
    rotation = gaussians._rotation[selected_indices]
    opacities = gaussians._opacity[selected_indices]

    if scale is not None:
        if scale.dim() == 0:
            xyz = xyz * scale
            scaling = inverse_sigmoid(torch.exp(scaling) * scale)
        elif scale.dim() == 1:
            s = scale[:, None, None]
            xyz = xyz.unsqueeze(0) * s
            scaling = inverse_sigmoid(torch.exp(scaling.unsqueeze(0)) * s)
        else:
            raise ValueError(f"Unexpected scale shape {scale.shape}")

    if rot_mat is not None:
        quat_r = matrix_to_quaternion(rot_mat)
        nrot = rot_mat.size(0)
        if nrot == 1:
            xyz = xyz @ rot_mat[0].T if xyz.dim() == 2 else torch.matmul(xyz, rot_mat[0].T)
        elif nrot == xyz.size(0) and xyz.dim() == 2:
            xyz = torch.einsum("nij,nj->ni", rot_mat, xyz)
        else:
            pts = xyz if xyz.dim() == 3 else xyz.unsqueeze(0).expand(nrot, xyz.size(-2), 3)
            xyz = torch.einsum("bij,bnj->bni", rot_mat, pts)
        if rotation.numel() > 0:
            if quat_r.size(0) == rotation.size(0) and xyz.dim() == 2:
                rotation = _compose_rotation(quat_r, rotation)
            else:
                B, N = quat_r.size(0), rotation.size(0)
                rotation = _compose_rotation(quat_r[:, None, :].expand(B, N, 4).reshape(B * N, 4),
                                             rotation[None].expand(B, N, 4).reshape(B * N, 4)).view(B, N, 4)

    if translation is not None:
        if translation.dim() == 1:
            xyz = xyz + translation
        elif translation.dim() == 2:
            xyz = (xyz.unsqueeze(0) if xyz.dim() == 2 else xyz) + translation[:, None, :]
        else:
            raise ValueError(f"Unexpected translation shape {translation.shape}")

    if new_opacity is not None:
        mask = opacities < opacities.mean() * 5
        if new_opacity.dim() == 0:
            result = opacities.clone()
            result[mask] = new_opacity
        elif new_opacity.dim() == 1:
            B, N = new_opacity.size(0), opacities.size(0)
            result = opacities[None, :].expand(B, N).clone()
            mask_b = mask[None, :].expand(B, N)
            result[mask_b] = new_opacity[:, None].expand(B, N)[mask_b]
        else:
            raise ValueError(f"Unexpected new_opacity shape {new_opacity.shape}")
        opacities = result

    return xyz, scaling, rotation, opacities
