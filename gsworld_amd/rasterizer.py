"""Python API of the drop-in ``diff_gaussian_rasterization`` package.

Mirrors the un-vendored upstream ``diff_gaussian_rasterization/__init__.py`` that GSWorld renders through
(``gaussian_renderer.render`` -> ``GaussianRasterizer``; call site
/root/reference/gsworld/mani_skill/utils/wrappers/gs_world_wrapper.py:266-267): same class and field names,
same argument meaning, same exceptions, same return tuple ``(color, radii, invdepth)``.

``SparseGaussianAdam`` is deliberately NOT exported: GSWorld probes for it
(gs_world_wrapper.py:22-26) and its absence selects the ``separate_sh=False`` call path.
"""
from __future__ import annotations

from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


def cpu_deep_copy_tuple(input_tuple):
    copied = [item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple]
    return tuple(copied)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    antialiasing: bool


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        rs = raster_settings
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
                rs.sh_degree, rs.campos, rs.prefiltered, rs.antialiasing, rs.debug)
        if rs.debug:
            cpu_args = cpu_deep_copy_tuple(args)  # snapshot before anything can corrupt the inputs
            try:
                num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, invdepths = \
                    _C.rasterize_gaussians(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            cap = _C.nosync_capacity(means3D.size(0), rs.image_height, rs.image_width,
                                     device=means3D.device if means3D.is_cuda else None)
            done = False
            if cap is not None and not rs.prefiltered and means3D.dim() == 2 and means3D.size(1) == 3:
                # (the Function keeps num_rendered to itself: it need not be read back in mid-frame when the instance
                #  list is sized by the bound P x tiles -- _C.rasterize_gaussians_nosync; same kernels, same state.
                #  The list is held until the backward: several forwards kept alive at once, or a shared device, may
                #  not have the room -- then upstream's exact sizing is what runs)
                try:
                    num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, invdepths = \
                        _C.rasterize_gaussians_nosync(
                            cap, rs.bg, means3D, opacities, scales, rotations, rs.scale_modifier, rs.viewmatrix,
                            rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree,
                            rs.campos, rs.antialiasing, rs.debug, colors=colors_precomp, cov3D_precomp=cov3Ds_precomp)
                    done = True
                except torch.cuda.OutOfMemoryError:
                    torch.cuda.empty_cache()
            if not done:
                num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, invdepths = \
                    _C.rasterize_gaussians(*args)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        # (an output the loss does not use arrives as None in backward, not as a zero image autograd had to fill:
        #  GSWorld's losses never touch invdepths)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities,
                              geomBuffer, binningBuffer, imgBuffer)
        return color, radii, invdepths

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii, grad_out_depth):
        num_rendered = ctx.num_rendered
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities, geomBuffer,
         binningBuffer, imgBuffer) = ctx.saved_tensors
        if grad_out_color is None:
            grad_out_color = torch.zeros((3, rs.image_height, rs.image_width), device=means3D.device)
        args = (rs.bg, means3D, radii, colors_precomp, opacities, scales, rotations, rs.scale_modifier,
                cov3Ds_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color,
                grad_out_depth, sh, rs.sh_degree, rs.campos, geomBuffer, num_rendered, binningBuffer, imgBuffer,
                rs.antialiasing, rs.debug)
        if rs.debug:
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                grads_raw = _C.rasterize_gaussians_backward(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise ex
        else:
            grads_raw = _C.rasterize_gaussians_backward(*args)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
         grad_scales, grad_rotations) = grads_raw
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales,
                grad_rotations, grad_cov3Ds_precomp, None)


class _RasterizeGaussiansFused(torch.autograd.Function):
    """Extension of the upstream Function: the model's parameters go in AS STORED -- features_dc and
    features_rest separately (no per-step ``cat``), and, per ``param_space`` bit, opacity logits / log scales /
    un-normalised quaternions, whose sigmoid / exp / normalize run inside preprocess and whose chain rule runs
    inside the backward kernel.  Gradients come back in the same parameter space."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh_dc, sh_rest, opacities, scales, rotations, raster_settings, param_space):
        rs = raster_settings
        empty = torch.empty(0, device=means3D.device)
        cap = _C.nosync_capacity(means3D.size(0), rs.image_height, rs.image_width,
                                 device=means3D.device if means3D.is_cuda else None)
        done = False
        if cap is not None and not rs.prefiltered:
            # no host read of num_rendered in mid-frame: the instance list is sized by a bound no frame can exceed
            # (room permitting -- see _RasterizeGaussians.forward)
            try:
                num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, invdepths = _C.rasterize_gaussians_nosync(
                    cap, rs.bg, means3D, opacities, scales, rotations, rs.scale_modifier, rs.viewmatrix, rs.projmatrix,
                    rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh_dc, rs.sh_degree, rs.campos,
                    rs.antialiasing, rs.debug, sh_rest=sh_rest, param_space=param_space)
                done = True
            except torch.cuda.OutOfMemoryError:
                torch.cuda.empty_cache()
        if not done:
            num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, invdepths = _C.rasterize_gaussians(
                rs.bg, means3D, empty, opacities, scales, rotations, rs.scale_modifier, empty, rs.viewmatrix,
                rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh_dc, rs.sh_degree, rs.campos,
                rs.prefiltered, rs.antialiasing, rs.debug, sh_rest=sh_rest, param_space=param_space)
        ctx.raster_settings, ctx.num_rendered, ctx.param_space = rs, num_rendered, param_space
        ctx.set_materialize_grads(False)  # (see _RasterizeGaussians.forward)
        ctx.save_for_backward(means3D, scales, rotations, radii, sh_dc, sh_rest, opacities, geomBuffer, binningBuffer,
                              imgBuffer)
        return color, radii, invdepths

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii, grad_out_depth):
        rs = ctx.raster_settings
        (means3D, scales, rotations, radii, sh_dc, sh_rest, opacities, geomBuffer, binningBuffer,
         imgBuffer) = ctx.saved_tensors
        empty = torch.empty(0, device=means3D.device)
        if grad_out_color is None:
            grad_out_color = torch.zeros((3, rs.image_height, rs.image_width), device=means3D.device)
        (grad_means2D, _gc, grad_opacities, grad_means3D, _gcov, grad_dc, grad_scales, grad_rotations,
         grad_rest) = _C.rasterize_gaussians_backward(
            rs.bg, means3D, radii, empty, opacities, scales, rotations, rs.scale_modifier, empty, rs.viewmatrix,
            rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, grad_out_depth, sh_dc, rs.sh_degree, rs.campos,
            geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, rs.antialiasing, rs.debug, sh_rest=sh_rest,
            param_space=ctx.param_space)
        return (grad_means3D, grad_means2D, grad_dc, grad_rest, grad_opacities.view_as(opacities), grad_scales,
                grad_rotations, None, None)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, shs_rest=None, param_space=0):
        """Upstream's signature; ``shs_rest`` / ``param_space`` are this library's extension (see
        :class:`_RasterizeGaussiansFused`): ``shs`` is then features_dc (P,1,3)."""
        rs = self.raster_settings
        if shs_rest is not None or param_space:
            if shs is None or shs_rest is None or colors_precomp is not None or cov3D_precomp is not None or \
                    scales is None or rotations is None:
                raise Exception("shs_rest / param_space need shs (dc) + shs_rest, scales and rotations")
            return _RasterizeGaussiansFused.apply(means3D, means2D, shs, shs_rest, opacities, scales, rotations, rs,
                                                  int(param_space))
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        empty = torch.Tensor([])
        if shs is None:
            shs = empty
        if colors_precomp is None:
            colors_precomp = empty
        if scales is None:
            scales = empty
        if rotations is None:
            rotations = empty
        if cov3D_precomp is None:
            cov3D_precomp = empty
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, rs)


__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "cpu_deep_copy_tuple"]
