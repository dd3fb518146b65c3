"""Load-time layout of a Gaussian model for view-frustum culling (``GsrInputs.cull_blocks`` / ``orig_index``, include/gsr.h).

The reference rasterizer projects every Gaussian in front of the near plane and finds out at the very end of its
per-Gaussian stage that the tile rect is empty (upstream forward.cu ``preprocessCUDA`` -> ``getRect``; SURVEY.md 8a
rows A3/A4): a sensor camera of a table-top scene sees an eighth of the model, the other seven eighths pay the whole
covariance chain.  A GSWorld scene does not change between frames except for rigid poses of labelled parts
(``gs_world_wrapper.py:110-162``), so the model can be laid out ONCE so that whole workgroups of the per-Gaussian kernel
are dismissed by one box test:

* :func:`morton_order` -- a permutation that sorts the Gaussians by (part label, 30-bit Morton code of the centre):
  256 consecutive Gaussians then sit in one small box and share a pose;
* :func:`build_cull_blocks` -- per block of 256 consecutive Gaussians the box of the centres, a bound ``rho`` of the
  largest 3-sigma axis and the common label.  Valid for ANY order of the model (an unsorted model just has boxes that
  span the scene and are never culled);
* :class:`SceneLayout` -- the permuted copy of a model's arrays, its ``orig_index`` and its blocks, ready for
  ``FrameRenderer.render(layout=...)``.

Nothing here changes what is rendered: a block is only skipped when every Gaussian in it has ``radii == 0`` in the
reference as well (csrc/preprocess.hip ``prep_block_culled`` states the bound), and a permuted model keeps the original
numbering in every output, in the lists and in the order of depth ties.  ``tests/test_layout_gpu.py`` holds frames,
radii and point lists bit-identical with and without a layout; ``tests/test_layout_cpu.py`` checks a numpy restatement
of the block test against the oracle's radii.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from ._lib import RAW_ROTATIONS, RAW_SCALES

BLOCK = 256  # = GSR_BLOCK, the workgroup of preprocess_kernel: one block record per workgroup


def _part1by2(v: torch.Tensor) -> torch.Tensor:
    """Spreads the low 10 bits of v so that two zero bits separate consecutive bits."""
    v = v & 0x3FF
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


SIZE_CLASS_QUANTILES = (0.5, 0.8, 0.95, 0.99)  # of the per-Gaussian bound rho: see morton_order


def morton_order(means3D: torch.Tensor, labels: torch.Tensor | None = None, rho: torch.Tensor | None = None) -> torch.Tensor:
    """Permutation (int64, ``new position -> original number``) that sorts by (label, size class, Morton code of the
    centre).  Stable: Gaussians with an equal key keep their original relative order.  ``rho`` (per-Gaussian largest
    scale): a block's radius bound is that of its LARGEST member, and splat sizes spread over two decades -- at configs[1]
    the median Gaussian has 0.9 cm, the largest of a random 256 3.8 cm -- so the Gaussians are first split into five
    size classes (quantiles 0.5 / 0.8 / 0.95 / 0.99) and sorted in space inside each: 61 % -> 70 % of the blocks culled
    from the sensor camera, 10 % -> 20 % from the dense view (79 % / 27 % hold no visible Gaussian)."""
    m = means3D.detach().to(torch.float32)
    finite = torch.isfinite(m).all(1)
    lo = torch.where(finite[:, None], m, torch.full_like(m, float("inf"))).amin(0)
    hi = torch.where(finite[:, None], m, torch.full_like(m, float("-inf"))).amax(0)
    # (robust to a handful of far floaters: quantise the 0.1 % .. 99.9 % range, clamp the rest)
    if m.shape[0] > 4096:
        sub = m[finite][:: max(1, int(finite.sum()) // 200_000)]
        lo = torch.quantile(sub, 0.001, dim=0)
        hi = torch.quantile(sub, 0.999, dim=0)
    span = (hi - lo).clamp_min(1e-12)
    q = ((m - lo) / span * 1023.0).nan_to_num(0.0).clamp(0, 1023).to(torch.int64)
    code = _part1by2(q[:, 0]) | (_part1by2(q[:, 1]) << 1) | (_part1by2(q[:, 2]) << 2)
    if rho is not None and m.shape[0] > 4096:
        r = rho.detach().reshape(-1).to(torch.float32).nan_to_num(0.0)
        qs = torch.quantile(r[:: max(1, r.shape[0] // 200_000)], torch.tensor(SIZE_CLASS_QUANTILES, device=r.device))
        code = code | (torch.bucketize(r, qs) << 30)
    if labels is not None:
        lab = labels.detach().reshape(-1).to(torch.float64).nan_to_num(-1.0).to(torch.int64)  # (.long(): truncation)
        code = code | ((lab - lab.min()) << 33)
    return torch.sort(code, stable=True).indices


def _rotation_norm_bound(rot: torch.Tensor) -> torch.Tensor:
    """|R(q)|_2 <= max(1, 2 |q|^2 - 1): R(q) = (1 - n) I + n R(q / |q|) for the kernel's quaternion -> matrix formula."""
    n = (rot.to(torch.float64) ** 2).sum(1)
    return torch.clamp(2.0 * n - 1.0, min=1.0)


def build_cull_blocks(means3D: torch.Tensor, scales: torch.Tensor | None, rotations: torch.Tensor | None,
                      labels: torch.Tensor | None = None, param_space: int = 0,
                      cov3D_precomp: torch.Tensor | None = None) -> torch.Tensor:
    """``(ceil(P / 256), 8)`` float32: lo.xyz, hi.xyz, rho, label per block of 256 consecutive Gaussians.
    ``scales`` / ``rotations`` as they are handed to the rasterizer (``param_space``: OR of ``_lib.RAW_*`` when they are
    the raw parameters); ``cov3D_precomp`` instead when the covariances are given directly."""
    P = means3D.shape[0]
    dev = means3D.device
    nb = (P + BLOCK - 1) // BLOCK
    if P == 0:
        return torch.zeros((0, 8), dtype=torch.float32, device=dev)
    m = means3D.detach().to(torch.float32)
    if cov3D_precomp is not None:
        c = cov3D_precomp.detach().to(torch.float64)
        rho = torch.sqrt((c[:, 0] + c[:, 3] + c[:, 5]).clamp_min(0.0))  # lambda_max <= trace
    else:
        s = scales.detach().to(torch.float64)
        if param_space & RAW_SCALES:
            s = torch.exp(s)
        r = rotations.detach().to(torch.float64)
        if param_space & RAW_ROTATIONS:
            r = r / r.norm(dim=1, keepdim=True).clamp_min(1e-12)
        rho = s.abs().amax(1) * _rotation_norm_bound(r)
    rho = (rho * (1.0 + 1e-4)).to(torch.float32)
    rho = torch.where(torch.isfinite(rho), rho, torch.full_like(rho, float("inf")))
    pad = nb * BLOCK - P

    def blocks(t):  # (the last block is padded with copies of the last Gaussian: bounds and labels do not change)
        if pad:
            t = torch.cat([t, t[-1:].expand(pad, *t.shape[1:])])
        return t.reshape(nb, BLOCK, *t.shape[1:])

    mb = blocks(m)
    out = torch.empty((nb, 8), dtype=torch.float32, device=dev)
    out[:, 0:3] = mb.amin(1)  # (amin / amax propagate NaN: such a block is never culled)
    out[:, 3:6] = mb.amax(1)
    out[:, 6] = blocks(rho).amax(1)
    if labels is not None:
        lb = blocks(labels.detach().reshape(-1).to(torch.float32))
        same = (lb.to(torch.int64) == lb[:, :1].to(torch.int64)).all(1) & torch.isfinite(lb).all(1)
        out[:, 7] = torch.where(same, lb[:, 0], torch.full_like(lb[:, 0], float("nan")))
    else:
        # no labels: NaN = "members have no common label" -- should this layout ever be rendered WITH a part transform,
        # no block is then tested under a pose its members do not share (preprocess.hip prep_block_culled; the field is
        # not read at all while GsrInputs.part_labels is NULL)
        out[:, 7] = float("nan")
    return out.contiguous()


@dataclass
class SceneLayout:
    """A model laid out for culling: the arrays in Morton order, their original numbering and the block bounds.

    ``arrays`` holds the permuted tensors under the names they were given; ``layout`` is the argument of
    ``FrameRenderer.render(layout=...)`` / ``_C.forward_raw(layout=...)``."""
    arrays: dict
    perm: torch.Tensor        # (P,) int64: position in the permuted arrays -> original number
    orig_index: torch.Tensor | None  # (P,) int32, the same for the kernels; None: the model kept its order
    cull_blocks: torch.Tensor  # (ceil(P/256), 8) float32

    @property
    def layout(self):
        return (self.cull_blocks, self.orig_index)

    @classmethod
    def build(cls, means3D: torch.Tensor, scales: torch.Tensor, rotations: torch.Tensor, labels: torch.Tensor | None = None,
              param_space: int = 0, reorder: bool = True, **per_gaussian) -> "SceneLayout":
        """``per_gaussian``: every other (P, ...) tensor of the model (opacities, shs, features_dc, features_rest, ...):
        permuted alongside.  ``reorder=False`` keeps the model's order (bounds only: correct, rarely tight)."""
        P = means3D.shape[0]
        dev = means3D.device
        if reorder:
            from ._lib import plan_query

            # a permuted model is only taken by inference frames on the default sort / placement path; a model beyond the
            # sample sort (8 388 608 Gaussians) never gets one, whatever the camera: keep its order, hand over bounds only
            if plan_query(640, 480, P, permuted=True, forward_only=True, tuned=False) is None:
                import warnings

                warnings.warn(f"SceneLayout: a model of {P} Gaussians is rendered through the LSD radix depth sort, which "
                              "takes no permuted model: the layout keeps the caller's order (block bounds only)", stacklevel=2)
                reorder = False
        if reorder:
            s = scales.detach().to(torch.float32)
            perm = morton_order(means3D, labels, (torch.exp(s) if param_space & RAW_SCALES else s).abs().amax(1))
        else:
            perm = torch.arange(P, device=dev)
        arrays = {"means3D": means3D, "scales": scales, "rotations": rotations, **per_gaussian}
        if labels is not None:
            arrays["labels"] = labels
        out = {}
        for k, t in arrays.items():
            if t is None:
                out[k] = None
                continue
            if t.shape[0] != P:
                raise ValueError(f"{k}: first dimension {t.shape[0]} != {P} Gaussians")
            out[k] = t.detach()[perm].contiguous() if reorder else t.detach().contiguous()
        blocks = build_cull_blocks(out["means3D"], out["scales"], out["rotations"], out.get("labels"), param_space)
        return cls(out, perm, perm.to(torch.int32).contiguous() if reorder else None, blocks)
