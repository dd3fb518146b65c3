"""CPU restatement of the reference's ``transform_gaussians`` -- TEST INFRASTRUCTURE, not product code.

Follows /root/reference/gsworld/utils/gs_utils.py:283-385 (scale -> rotate -> translate -> opacity on a selected index
set, with its exact output shapes), :242-249 (quaternion composition that keeps the norm) and :169 (``inverse_sigmoid``),
plus the two PyTorch3D-derived helpers the reference imports from ManiSkill
(``mani_skill.utils.geometry.rotation_conversions.matrix_to_quaternion`` / ``quaternion_multiply``; real-first
``wxyz``), which are NOT in the reference tree -- the versions below are the stub ``tools/make_golden.py`` hands to the
imported reference, so the golden rotations are pinned only up to this stub.

Pinned by ``tests/golden/transform_gaussians.npz`` (outputs of the imported reference function itself,
``tests/test_host_cpu.py``).  Only ``tests/`` and ``tools/make_golden.py`` import this module; the product's fused HIP
operator (``gsworld_amd/csrc/transform.hip``) is checked against it and against the golden vectors directly.
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
