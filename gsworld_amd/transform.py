"""Per-step rigid transform of labelled Gaussians -- the step immediately before the rasterizer in GSWorld
(SURVEY.md 8f-1).

:class:`FusedPartTransform` drives the fused HIP operator (``gsworld_amd/csrc/transform.hip``) that replaces the
wrapper's per-link mask / gather / ``transform_gaussians`` / scatter passes
(/root/reference/gsworld/mani_skill/utils/wrappers/gs_world_wrapper.py:110-162, 244-265;
gsworld/utils/gs_utils.py:283-385).  The quaternion helpers are the host side of the pose-table packing
(``matrix_to_quaternion`` as ManiSkill / PyTorch3D define it, real part first).  The CPU restatement of
``transform_gaussians`` itself is checker code and lives in ``oracle/transform_ref.py``.
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


class FusedPartTransform:
    """One-pass replacement of ``GSWorldWrapper.transform_gs_perlink`` + the write-back loop of
    ``_render_gsworld`` (gs_world_wrapper.py:110-162, 244-265), for one environment or a batch of ``E``.

    ``part_labels`` maps a part name (robot link or tracked actor) to its semantic label(s), as
    ``xarm_gs_semantics`` / ``obj_gs_semantics`` do (/root/reference/gsworld/constants.py:402-505).  Per step the
    caller passes one 4x4 per part (and environment) -- for a link ``sim2gs @ link_now @ inv(link_scan) @ inv(sim2gs)``
    (gs_world_wrapper.py:120), for an actor the rigid part of ``sim2gs @ pose @ inv(sim2gs_obj)`` plus its uniform
    scale (``:146-156``) -- and gets the transformed ``xyz`` / ``rotation`` buffers the rasterizer should read:
    ``(P,3)`` / ``(P,4)`` for ``(K,4,4)`` matrices, ``(E,P,3)`` / ``(E,P,4)`` for ``(E,K,4,4)`` -- environment ``e`` is
    slice ``e``, exactly what the wrapper's ``gs_movable_pts[key][j][i]`` write-back produces for ``i = e``.

    ``scaled_parts``: names of the parts that are given a per-environment scale VECTOR by the wrapper (the tracked
    actors: ``scale * object_scale[actor_key]`` with ``scale`` of shape ``(num_envs,)``).  For those the reference also
    rewrites the log-scale parameter (``inverse_sigmoid(exp(scaling) * scale)``, gs_utils.py:296-304) and the wrapper
    writes it back (its ``shape[0] == num_envs`` test passes for a ``(num_envs, n, 3)`` tensor); pass ``scaling=`` to
    :meth:`apply` to get that third buffer.  Links (``scale=None``) keep their scaling; opacity is never changed
    (``new_opacity=None`` at both call sites).
    """

    def __init__(self, part_labels: dict, semantics: torch.Tensor, lut_size: int = 2048, scaled_parts=()):
        self.names = list(part_labels.keys())
        lut = torch.full((lut_size,), -1, dtype=torch.int32)
        for k, name in enumerate(self.names):
            labels = part_labels[name]
            for lab in (labels if isinstance(labels, (list, tuple)) else [labels]):
                if not 0 <= int(lab) < lut_size:
                    raise ValueError(f"label {lab} of part {name!r} outside the LUT (size {lut_size})")
                lut[int(lab)] = k
        scaled_parts = tuple(scaled_parts)
        unknown = [n for n in scaled_parts if n not in part_labels]
        if unknown:
            raise ValueError(f"scaled_parts names unknown parts: {unknown}")
        self.device = semantics.device
        self.lut = lut.to(self.device)
        self.semantics = semantics.reshape(-1).to(torch.float32).contiguous()
        self.rescale = torch.tensor([1 if n in set(scaled_parts) else 0 for n in self.names],
                                    dtype=torch.uint8).to(self.device)
        self._any_rescale = len(tuple(scaled_parts)) > 0
        self._out = {}
        self._table = None

    def pack(self, matrices: torch.Tensor, scales: torch.Tensor | None = None) -> torch.Tensor:
        """(K,4,4) or (E,K,4,4) rigid matrices (+ optional uniform scales of shape (K,) / (E,K)) -> (K,17) / (E,K,17)
        transform table (host math, K ~ 18)."""
        M = matrices.detach().to("cpu", torch.float32)
        lead = M.shape[:-2]
        if M.shape[-3] != len(self.names):
            raise ValueError(f"expected {len(self.names)} matrices, got {M.shape[-3]}")
        q = matrix_to_quaternion(M[..., :3, :3])
        s = torch.ones(lead) if scales is None else scales.detach().to("cpu", torch.float32).reshape(lead)
        return torch.cat((M[..., :3, :3].reshape(lead + (9,)), M[..., :3, 3], s[..., None], q), dim=-1).contiguous()

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
        lead = tuple(matrices.shape[:-2])
        if lead[-1:] != (K,) or tuple(matrices.shape[-2:]) != (4, 4) or len(lead) > 2 or \
                matrices.dtype != torch.float32 or not matrices.is_contiguous():
            raise ValueError(f"expected contiguous float32 ({K},4,4) or (E,{K},4,4) matrices, got {tuple(matrices.shape)}")
        if scales is not None and (tuple(scales.shape) != lead or scales.dtype != torch.float32 or
                                   not scales.is_contiguous()):
            raise ValueError(f"expected contiguous float32 {lead} scales")
        if self._table is None or tuple(self._table.shape[:-1]) != lead:
            self._table = torch.empty(lead + (17,), dtype=torch.float32, device=self.device)
        n = self._table.numel() // 17
        with torch.cuda.device(self.device):
            check(L.gsr_pack_part_transforms(
                n, C.c_void_p(matrices.data_ptr()), C.c_void_p(scales.data_ptr() if scales is not None else 0),
                C.c_void_p(self._table.data_ptr()), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return self._table

    def parts(self, matrices: torch.Tensor, scales=None, env: int | None = None):
        """-> the ``parts=`` tuple of :meth:`gsworld_amd.renderer.FrameRenderer.render` for this step's poses: the
        transform is then applied inside preprocess and no transformed copy of the model is written.  ``matrices``
        (K,4,4) or (E,K,4,4) on the DEVICE (+ ``scales``); for a batch, ``env`` selects the environment (None: a list with
        one tuple per environment).  The pose table lives in a persistent buffer (capturable)."""
        table = self.pack_on_device(matrices, scales) if matrices.is_cuda else \
            self.pack(matrices, scales).to(self.device, non_blocking=True)
        return self.parts_of_table(table, env)

    def parts_of_table(self, table: torch.Tensor, env: int | None = None):
        """The ``parts=`` tuple(s) over a pose table that is already packed ((K,17) or (E,K,17) on the device)."""
        rescale = self.rescale if self._any_rescale else None
        if table.dim() == 2:
            return (self.semantics, self.lut, table, rescale)
        if env is not None:
            return (self.semantics, self.lut, table[env], rescale)
        return [(self.semantics, self.lut, table[e], rescale) for e in range(table.shape[0])]

    def _buffer(self, name, shape, like):
        t = self._out.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = self._out[name] = torch.empty(shape, dtype=torch.float32, device=like.device)
        return t

    def apply(self, xyz: torch.Tensor, rotation: torch.Tensor, matrices: torch.Tensor, scales=None, scaling=None):
        """-> (xyz', rotation') [, scaling'] in persistent output buffers (overwritten by the next call).  ``matrices``
        (K,4,4) or (E,K,4,4) (+ ``scales`` (K,) / (E,K)) may live on the host (packed with torch, one small H2D copy) or
        on the device (packed by a kernel: no synchronisation, capturable).  ``scaling``: the (P,3) log-scale parameter;
        when given, a third buffer comes back in which the ``scaled_parts`` carry the reference's rewritten values."""
        import ctypes as C

        from ._lib import check, lib

        if not xyz.is_cuda:
            raise RuntimeError("FusedPartTransform.apply: tensors must live on a HIP device (no CPU path)")
        L = lib()
        if not getattr(L, "_xf_bound", False):
            L.gsr_transform_gaussians_batch.restype = C.c_int
            L.gsr_transform_gaussians_batch.argtypes = [
                C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            L._xf_bound = True
        xyz = xyz.detach().to(torch.float32).contiguous()
        rotation = rotation.detach().to(torch.float32).contiguous()
        P = xyz.shape[0]
        if tuple(xyz.shape) != (P, 3) or tuple(rotation.shape) != (P, 4) or self.semantics.numel() != P:
            raise ValueError("expected xyz (P,3), rotation (P,4) and one label per Gaussian")
        if matrices.dim() not in (3, 4):
            raise ValueError(f"expected (K,4,4) or (E,K,4,4) matrices, got {tuple(matrices.shape)}")
        batched = matrices.dim() == 4
        E = matrices.shape[0] if batched else 1
        lead = (E,) if batched else ()
        xyz_out = self._buffer("xyz", lead + (P, 3), xyz)
        rot_out = self._buffer("rot", lead + (P, 4), xyz)
        scaling_out = None
        if scaling is not None:
            scaling = scaling.detach().to(torch.float32).contiguous()
            if tuple(scaling.shape) != (P, 3):
                raise ValueError("expected scaling (P,3)")
            scaling_out = self._buffer("scaling", lead + (P, 3), xyz)
        if matrices.is_cuda:
            table = self.pack_on_device(matrices, scales)
        else:
            table = self.pack(matrices, scales).to(self.device, non_blocking=True)
        with torch.cuda.device(self.device):
            check(L.gsr_transform_gaussians_batch(
                P, E, C.c_void_p(xyz.data_ptr()), C.c_void_p(rotation.data_ptr()),
                C.c_void_p(scaling.data_ptr()) if scaling is not None else None, C.c_void_p(self.semantics.data_ptr()),
                C.c_void_p(self.lut.data_ptr()), self.lut.numel(), C.c_void_p(table.data_ptr()), len(self.names),
                C.c_void_p(self.rescale.data_ptr()), C.c_void_p(xyz_out.data_ptr()), C.c_void_p(rot_out.data_ptr()),
                C.c_void_p(scaling_out.data_ptr()) if scaling_out is not None else None,
                C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        if scaling_out is not None:
            return xyz_out, rot_out, scaling_out
        return xyz_out, rot_out
