"""``gaussian_renderer.render`` of the 3DGS python layer -- the function GSWorld calls
per camera per env (/root/reference/gsworld/mani_skill/utils/wrappers/gs_world_wrapper.py:266-267:
``render(cam_param, gs4render, self.robot_pipe, background, use_trained_exp=False, separate_sh=...)["render"]``).

Packs the rasterizer arguments exactly as upstream does (activations sigmoid / exp / normalize through the model's
getters, ``cat(dc, rest)`` -> (N,16,3) SH, tan(FoV/2), transposed 4x4s) and returns the same dict.
"""
import math

import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer


_GETTERS = ("get_xyz", "get_opacity", "get_scaling", "get_rotation", "get_features", "get_features_dc",
            "get_features_rest")


def _stock_getters(pc) -> bool:
    """True when ``pc`` renders exactly its stored ``_xyz`` ... through the base GaussianModel's getters."""
    try:
        from scene.gaussian_model import GaussianModel as Base
    except ImportError:  # (package imported as gsworld_amd.gs_compat... rather than through sys.path)
        from ..scene.gaussian_model import GaussianModel as Base
    cls = type(pc)
    if not isinstance(pc, Base):
        from gsworld_amd.gs_compat.scene.gaussian_model import GaussianModel as Base2
        if not isinstance(pc, Base2):
            return False
        Base = Base2
    return all(getattr(cls, n, None) is getattr(Base, n) for n in _GETTERS)


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, separate_sh=False,
           override_color=None, use_trained_exp=False):
    # gradient carrier for the 2D means (densification statistics read its .grad); the no-grad shortcut below creates it
    # only when somebody reads it from the result
    def make_screenspace_points():
        pts = torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True, device=pc.get_xyz.device) + 0
        try:
            pts.retain_grad()
        except Exception:  # noqa: BLE001
            pass
        return pts

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

    # Inference fast path: when no gradient can flow -- GSWorld renders a detached deepcopy of a
    # frozen model -- the two SH parameters are handed to the rasterizer as they are stored instead of being
    # concatenated into a fresh (N,16,3) tensor for every frame (upstream `shs = pc.get_features`: 564 MB of
    # traffic at 1.47 M Gaussians, as much as the whole frame).  Same image bit for bit (tests/test_dropin_gpu.py).
    raw = tuple(getattr(pc, n, None) for n in ("_xyz", "_opacity", "_scaling", "_rotation", "_features_dc",
                                              "_features_rest"))
    # The shortcut reads the stored tensors, not the getters: it is only taken for a model whose getters are the stock
    # ones (a subclass that overrides get_xyz / get_features / ... -- deformation, pose or scale optimisation -- may
    # render something else than its raw attributes, and may be differentiable while they are frozen).
    no_grad = _stock_getters(pc) and (not torch.is_grad_enabled()
                                      or not any(t is not None and t.requires_grad for t in raw))
    fast = (no_grad and override_color is None and not getattr(pipe, "convert_SHs_python", False)
            and raw[5] is not None and raw[5].shape[1] > 0 and raw[4].is_contiguous() and raw[5].is_contiguous())
    # opt-in on top of it (pipe.fused_activations): hand over the RAW parameters and let preprocess apply sigmoid /
    # exp / normalize in its canonical float32 order -- three fewer passes over the model per frame, but exp is then
    # this library's, not torch's (last-ulp differences in the image), so it is not the default
    fused_ok = (getattr(pipe, "fused_activations", False) and not getattr(pipe, "compute_cov3D_python", False)
                and getattr(pc, "opacity_activation", None) is torch.sigmoid
                and getattr(pc, "scaling_activation", None) is torch.exp
                and getattr(pc, "rotation_activation", None) is torch.nn.functional.normalize)
    fused = fast and fused_ok
    screenspace_points = None if fast else make_screenspace_points()
    means2D = screenspace_points
    # the same opt-in while TRAINING: raw parameters and the two SH tensors go through the autograd Function as they
    # are stored; activations and their chain rule run inside the preprocess kernels (no sigmoid / exp / normalize /
    # cat passes and none of their backward passes per step)
    if (fused_ok and not no_grad and override_color is None and not getattr(pipe, "convert_SHs_python", False)
            and raw[5] is not None and raw[5].shape[1] > 0):
        from gsworld_amd._lib import RAW_OPACITY, RAW_ROTATIONS, RAW_SCALES

        rendered_image, radii, depth_image = rasterizer(
            means3D=means3D, means2D=means2D, opacities=pc._opacity, shs=pc._features_dc.contiguous(),
            shs_rest=pc._features_rest.contiguous(), scales=pc._scaling, rotations=pc._rotation,
            param_space=RAW_OPACITY | RAW_SCALES | RAW_ROTATIONS)
        return _finish(rendered_image, radii, depth_image, screenspace_points, viewpoint_camera, pc, use_trained_exp)

    scales = rotations = cov3D_precomp = None
    if fused:
        opacity, scales, rotations = pc._opacity, pc._scaling, pc._rotation
    else:
        opacity = pc.get_opacity
        if getattr(pipe, "compute_cov3D_python", False):
            cov3D_precomp = pc.get_covariance(scaling_modifier)
        else:
            scales = pc.get_scaling
            rotations = pc.get_rotation

    if fast:
        from gsworld_amd._lib import RAW_OPACITY, RAW_ROTATIONS, RAW_SCALES

        # ... and the frame goes through a renderer that is kept per device: its state buffers and the instance
        # capacity of the previous frame are reused, so nothing is allocated and the host is not consulted in the
        # middle of the frame (upstream's contract -- fresh byte tensors sized from a D2H read of num_rendered --
        # is only needed when a backward will follow).  The instance list is sized by the bound no frame can exceed
        # (P x tiles) when that fits the memory budget -- then the call returns without ever synchronising with the
        # device --; otherwise by the previous frame's count with head-room, the capacity flag is checked once the frame
        # is enqueued, and an overflow (the scene grew by more than the head-room) re-renders exactly.
        rs = raster_settings
        renderer = _frame_renderer(means3D.device)
        view = _View(rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, rs.viewmatrix, rs.projmatrix, rs.campos)

        P = means3D.shape[0]
        f32 = dict(dtype=torch.float32, device=means3D.device)
        # the frame is written straight into tensors the caller keeps (no copies out of renderer-owned buffers)
        outs = (torch.empty((3, rs.image_height, rs.image_width), **f32),
                torch.empty((1, rs.image_height, rs.image_width), **f32),
                torch.empty((P,), dtype=torch.int32, device=means3D.device))

        def enqueue():
            return renderer.render(
                view, means3D, opacity, shs=pc._features_dc, shs_rest=pc._features_rest, scales=scales,
                rotations=rotations, cov3D_precomp=cov3D_precomp, bg=rs.bg, sh_degree=rs.sh_degree,
                scale_modifier=rs.scale_modifier, antialiasing=rs.antialiasing, debug=rs.debug,
                param_space=(RAW_OPACITY | RAW_SCALES | RAW_ROTATIONS) if fused else 0, outputs=outs)

        with torch.no_grad():
            if P == 0:  # upstream: no launch, zero image
                rendered_image = torch.zeros((3, rs.image_height, rs.image_width), device=means3D.device)
                radii = torch.zeros((0,), dtype=torch.int32, device=means3D.device)
                depth_image = torch.zeros((1, rs.image_height, rs.image_width), device=means3D.device)
            else:
                rendered_image, radii, depth_image = enqueue()
                if not renderer.bounded:  # (a list sized by P x tiles cannot overflow: nothing to read back)
                    renderer.ensure_valid(enqueue)
        return _finish(rendered_image, radii, depth_image, make_screenspace_points, viewpoint_camera, pc,
                       use_trained_exp, lazy=True)

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
    return _finish(rendered_image, radii, depth_image, screenspace_points, viewpoint_camera, pc, use_trained_exp)


class _View:
    """What FrameRenderer reads from a camera (gsworld_amd.camera.ViewParams duck type)."""
    __slots__ = ("image_height", "image_width", "tanfovx", "tanfovy", "world_view_transform", "full_proj_transform",
                 "camera_center")

    def __init__(self, h, w, tanfovx, tanfovy, view, proj, campos):
        self.image_height, self.image_width, self.tanfovx, self.tanfovy = h, w, tanfovx, tanfovy
        self.world_view_transform, self.full_proj_transform, self.camera_center = view, proj, campos


_RENDERERS = {}


def _frame_renderer(device):
    from gsworld_amd.renderer import FrameRenderer

    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    r = _RENDERERS.get(key)
    if r is None:
        # frozen parameters = inference: forward_only frames (bit-identical image and radii; nothing kept for a backward)
        r = _RENDERERS[key] = FrameRenderer(torch.device(key[0], key[1]), forward_only=True, bound_capacity=True)
    return r


class _LazyResult(dict):
    """upstream's result dict whose rarely read entries are computed on first access: GSWorld reads ["render"] only, and
    `(radii > 0).nonzero()` alone is a compaction kernel plus a host synchronisation per frame.  Any access that is not a
    plain lookup of an eager key (iteration, ``in``, ``get``, ``len``, ...) materialises everything first, so the object
    is indistinguishable from the eager dict."""

    def __init__(self, eager: dict, lazy: dict):
        super().__init__(eager)
        self._lazy = lazy

    def _all(self):
        for k in list(self._lazy):
            dict.__setitem__(self, k, self._lazy.pop(k)())

    def __getitem__(self, k):
        if k in self._lazy:
            dict.__setitem__(self, k, self._lazy.pop(k)())
        return dict.__getitem__(self, k)

    def _materialised(name):  # noqa: N805
        def f(self, *a, **kw):
            self._all()
            return getattr(dict, name)(self, *a, **kw)
        return f

    for _n in ("__contains__", "__iter__", "__len__", "__repr__", "__eq__", "get", "keys", "values", "items", "copy",
               "pop", "setdefault", "update", "__reduce_ex__"):
        locals()[_n] = _materialised(_n)
    del _n, _materialised


def _finish(rendered_image, radii, depth_image, screenspace_points, viewpoint_camera, pc, use_trained_exp, lazy=False):
    if use_trained_exp:
        exposure = pc.get_exposure_from_name(viewpoint_camera.image_name)
        rendered_image = torch.matmul(rendered_image.permute(1, 2, 0), exposure[:3, :3]).permute(2, 0, 1) + \
            exposure[:3, 3, None, None]

    rendered_image = rendered_image.clamp(0, 1)
    if lazy:  # (`screenspace_points` is then the function that makes them)
        return _LazyResult({"render": rendered_image, "radii": radii, "depth": depth_image},
                           {"viewspace_points": screenspace_points,
                            "visibility_filter": lambda: (radii > 0).nonzero()})
    return {
        "render": rendered_image,
        "viewspace_points": screenspace_points,
        "visibility_filter": (radii > 0).nonzero(),
        "radii": radii,
        "depth": depth_image,
    }
