"""Restatement of GSWorldWrapper's per-step render glue, op for op in torch -- TEST INFRASTRUCTURE (the checker of
``gsworld_amd.closed_loop``), never imported by the product.

Follows /root/reference/gsworld/mani_skill/utils/wrappers/gs_world_wrapper.py:
  ``transform_gs_perlink`` (:110-162)  deep copy of the model; per robot link an ``isin`` mask over all labels,
                                       ``transform_gaussians(scale=None, rot_mat=(E,3,3), translation=(E,3))``; per
                                       tracked actor ``transform_gaussians(scale=(E,) vector, ...)`` -- a scale VECTOR,
                                       so the reference also rewrites the actor's log-scales;
  ``_render_gsworld``      (:232-275)  per camera, per environment: deep copy, then for every moved part and each of
                                       (xyz, scaling, rotation, opacity) the masked write-back guarded by
                                       ``value.shape[0] == num_envs``; upstream ``render()`` (sigmoid / exp / normalize /
                                       cat -> rasterizer); ``(x * 255).clamp(0, 255).to(uint8)`` of the HWC frame.
``transform_gaussians`` itself is ``oracle/transform_ref.py`` (pinned by the reference's own outputs).  The rasterizer
is passed in by the caller (the HIP ``GaussianRasterizer`` in exact mode, or the CPU oracle).
"""
from __future__ import annotations

import copy

import torch

from . import transform_ref


def transform_parts(model, part_labels: dict, matrices: torch.Tensor, scales: torch.Tensor, actors=()):
    """-> {part: (xyz, scaling, rotation, opacity)} as ``self.gs_movable_pts`` holds them.  ``matrices`` (E,K,4,4),
    ``scales`` (E,K); links are called with ``scale=None``, actors with their (E,) scale vector."""
    splats = copy.deepcopy(model)  # :112
    moved = {}
    labels = splats._semantics.long().squeeze(-1)
    for k, (name, lab) in enumerate(part_labels.items()):
        target = torch.tensor(lab if isinstance(lab, (list, tuple)) else [lab], device=labels.device).long()
        mask = torch.isin(labels, target)
        scale = scales[:, k].contiguous() if name in actors else None
        moved[name] = transform_ref.transform_gaussians(
            splats, torch.where(mask)[0], scale=scale, rot_mat=matrices[:, k, :3, :3],
            translation=matrices[:, k, :3, 3], new_opacity=None)
    return moved


def assemble_env(model, part_labels: dict, moved: dict, env: int, num_envs: int):
    """The ``gs4render`` of environment ``env`` (:244-265)."""
    gs = copy.deepcopy(model)
    labels = gs._semantics.long().squeeze(-1)
    for name, lab in part_labels.items():
        target = torch.tensor(lab if isinstance(lab, (list, tuple)) else [lab], device=labels.device).long()
        for attr, val in zip(("_xyz", "_scaling", "_rotation", "_opacity"), moved[name]):
            if val.shape[0] == num_envs:
                getattr(gs, attr)[torch.isin(labels, target)] = val[env]
    return gs


def render_step(model, part_labels: dict, cameras: dict, matrices, scales, rasterize, actors=(), background=None):
    """One ``step()`` of the wrapper's render side.  ``rasterize(view, means3D, shs, opacities, scales, rotations, bg)``
    -> (3,H,W) float image.  Returns {camera: uint8 (E,H,W,3)}."""
    if matrices.dim() == 3:
        matrices, scales = matrices[None], scales[None]
    E = matrices.shape[0]
    dev = model._xyz.device
    bg = torch.zeros(3, device=dev) if background is None else background
    moved = transform_parts(model, part_labels, matrices.to(dev), scales.to(dev), actors)
    out = {}
    for cam_name, view in cameras.items():
        frames = []
        for e in range(E):
            frames.append(render_model(assemble_env(model, part_labels, moved, e, E), view, rasterize, bg))
        out[cam_name] = torch.vstack(frames)
    return out


def render_model(gs, view, rasterize, bg):
    """What the wrapper does with one ``gs4render`` (:266-270): upstream ``render()`` (activations, SH concat,
    rasterizer, clamp to [0, 1]) and the uint8 conversion.  -> uint8 (1,H,W,3)."""
    n = gs._xyz.shape[0]
    color = rasterize(view, gs._xyz, torch.cat((gs._features_dc, gs._features_rest), dim=1),
                      torch.sigmoid(gs._opacity).reshape(n, 1), torch.exp(gs._scaling),
                      torch.nn.functional.normalize(gs._rotation), bg)
    img = color.clamp(0, 1).permute(1, 2, 0).unsqueeze(0)  # render()["render"] is clamped to [0, 1]
    return (img * 255).clamp(0, 255).to(torch.uint8)  # :268-270
