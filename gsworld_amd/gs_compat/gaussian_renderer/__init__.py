"""``gaussian_renderer.render`` of the 3DGS python layer -- the function GSWorld calls
per camera per env (/root/reference/gsworld/mani_skill/utils/wrappers/gs_world_wrapper.py:266-267:
``render(cam_param, gs4render, self.robot_pipe, background, use_trained_exp=False, separate_sh=...)["render"]``).

Packs the rasterizer arguments exactly as upstream does (activations sigmoid / exp / normalize through the model's
getters, ``cat(dc, rest)`` -> (N,16,3) SH, tan(FoV/2), transposed 4x4s) and returns the same dict.
"""
import math

import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, separate_sh=False,
           override_color=None, use_trained_exp=False):
    # gradient carrier for the 2D means (densification statistics read its .grad)
    screenspace_points = torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True,
                                          device=pc.get_xyz.device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:  # noqa: BLE001
        pass

    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5),
        tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center,
        prefiltered=False,
        debug=getattr(pipe, "debug", False),
        antialiasing=getattr(pipe, "antialiasing", False),
    )
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    means3D = pc.get_xyz
    means2D = screenspace_points
    opacity = pc.get_opacity

    scales = rotations = cov3D_precomp = None
    if getattr(pipe, "compute_cov3D_python", False):
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales = pc.get_scaling
        rotations = pc.get_rotation

    shs = colors_precomp = None
    if override_color is None:
        if getattr(pipe, "convert_SHs_python", False):
            from gsworld_amd.sh import eval_sh

            shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = pc.get_xyz - viewpoint_camera.camera_center.repeat(pc.get_features.shape[0], 1)
            dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp_normalized) + 0.5, 0.0)
        else:
            # separate_sh (3dgs_accel signature) is never requested: SparseGaussianAdam is not exported
            shs = pc.get_features
    else:
        colors_precomp = override_color

    rendered_image, radii, depth_image = rasterizer(
        means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp, opacities=opacity, scales=scales,
        rotations=rotations, cov3D_precomp=cov3D_precomp)

    if use_trained_exp:
        exposure = pc.get_exposure_from_name(viewpoint_camera.image_name)
        rendered_image = torch.matmul(rendered_image.permute(1, 2, 0), exposure[:3, :3]).permute(2, 0, 1) + \
            exposure[:3, 3, None, None]

    rendered_image = rendered_image.clamp(0, 1)
    return {
        "render": rendered_image,
        "viewspace_points": screenspace_points,
        "visibility_filter": (radii > 0).nonzero(),
        "radii": radii,
        "depth": depth_image,
    }
