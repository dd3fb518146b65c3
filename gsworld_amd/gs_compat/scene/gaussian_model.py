"""``scene.gaussian_model.GaussianModel`` -- the parameter container GSWorld subclasses
(``Semantic3DGSWrapper(GaussianModel)``, /root/reference/gsworld/mani_skill/utils/wrappers/semantic_3dgs_wrapper.py:34-184)
and merges (/root/reference/gsworld/utils/gaussian_merger.py:213-274).

Restates the layout and activations of the 3DGS python layer: raw parameters
``_xyz (N,3)``, ``_features_dc (N,1,3)``, ``_features_rest (N,15,3)``, ``_scaling (N,3)`` (log), ``_rotation (N,4)``
(r,x,y,z), ``_opacity (N,1)`` (logit); getters apply exp / normalize / sigmoid / cat.  The optimiser-side
methods (training_setup, densification) are kept to what the subclass hooks call.
"""
import os

import numpy as np
import torch
from torch import nn

from utils.general_utils import build_scaling_rotation, inverse_sigmoid, strip_symmetric
from utils.system_utils import mkdir_p
from gsworld_amd.sh import RGB2SH


class GaussianModel:
    def setup_functions(self):
        def build_covariance_from_scaling_rotation(scaling, scaling_modifier, rotation):
            L = build_scaling_rotation(scaling_modifier * scaling, rotation)
            return strip_symmetric(L @ L.transpose(1, 2))

        self.scaling_activation = torch.exp
        self.scaling_inverse_activation = torch.log
        self.covariance_activation = build_covariance_from_scaling_rotation
        self.opacity_activation = torch.sigmoid
        self.inverse_opacity_activation = inverse_sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    def __init__(self, sh_degree, optimizer_type="default"):
        self.active_sh_degree = 0
        self.optimizer_type = optimizer_type
        self.max_sh_degree = sh_degree
        self._xyz = torch.empty(0)
        self._features_dc = torch.empty(0)
        self._features_rest = torch.empty(0)
        self._scaling = torch.empty(0)
        self._rotation = torch.empty(0)
        self._opacity = torch.empty(0)
        self.max_radii2D = torch.empty(0)
        self.xyz_gradient_accum = torch.empty(0)
        self.denom = torch.empty(0)
        self.optimizer = None
        self.percent_dense = 0
        self.spatial_lr_scale = 0
        self.pretrained_exposures = None
        self.setup_functions()

    # ---- state ---------------------------------------------------------------------------------------------
    def capture(self):
        return (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling,
                self._rotation, self._opacity, self.max_radii2D, self.xyz_gradient_accum, self.denom,
                self.optimizer.state_dict() if self.optimizer is not None else None, self.spatial_lr_scale)

    def restore(self, model_args, training_args):
        (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation,
         self._opacity, self.max_radii2D, xyz_gradient_accum, denom, opt_dict, self.spatial_lr_scale) = model_args
        if training_args is not None:
            self.training_setup(training_args)
            if opt_dict is not None:
                self.optimizer.load_state_dict(opt_dict)
        self.xyz_gradient_accum = xyz_gradient_accum
        self.denom = denom

    # ---- activated views (what render() passes to the rasterizer) ----------------------------------------
    @property
    def get_scaling(self):
        return self.scaling_activation(self._scaling)

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_features_dc(self):
        return self._features_dc

    @property
    def get_features_rest(self):
        return self._features_rest

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    @property
    def get_exposure(self):
        return getattr(self, "_exposure", None)

    def get_exposure_from_name(self, image_name):
        if self.pretrained_exposures is None:
            return self._exposure[self.exposure_mapping[image_name]]
        return self.pretrained_exposures[image_name]

    def get_covariance(self, scaling_modifier=1):
        return self.covariance_activation(self.get_scaling, scaling_modifier, self._rotation)

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ---- initialisation from a point cloud: the one caller of simple_knn.distCUDA2 ------------------------
    def create_from_pcd(self, pcd, cam_infos, spatial_lr_scale):
        from simple_knn._C import distCUDA2

        self.spatial_lr_scale = spatial_lr_scale
        pts = torch.tensor(np.asarray(pcd.points)).float().cuda()
        color = RGB2SH(torch.tensor(np.asarray(pcd.colors)).float().cuda())
        features = torch.zeros((color.shape[0], 3, (self.max_sh_degree + 1) ** 2)).float().cuda()
        features[:, :3, 0] = color
        print("Number of points at initialisation : ", pts.shape[0])
        dist2 = torch.clamp_min(distCUDA2(pts), 0.0000001)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
        rots = torch.zeros((pts.shape[0], 4), device="cuda")
        rots[:, 0] = 1
        opacities = self.inverse_opacity_activation(0.1 * torch.ones((pts.shape[0], 1), dtype=torch.float,
                                                                     device="cuda"))
        self._xyz = nn.Parameter(pts.requires_grad_(True))
        self._features_dc = nn.Parameter(features[:, :, 0:1].transpose(1, 2).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(features[:, :, 1:].transpose(1, 2).contiguous().requires_grad_(True))
        self._scaling = nn.Parameter(scales.requires_grad_(True))
        self._rotation = nn.Parameter(rots.requires_grad_(True))
        self._opacity = nn.Parameter(opacities.requires_grad_(True))
        self.max_radii2D = torch.zeros((self.get_xyz.shape[0]), device="cuda")
        if cam_infos:
            self.exposure_mapping = {cam_info.image_name: idx for idx, cam_info in enumerate(cam_infos)}
            exposure = torch.eye(3, 4, device="cuda")[None].repeat(len(cam_infos), 1, 1)
            self._exposure = nn.Parameter(exposure.requires_grad_(True))

    # ---- optimiser plumbing (what prune / densify hooks of the subclass rely on) -------------------------
    def training_setup(self, training_args):
        self.percent_dense = training_args.percent_dense
        n = self.get_xyz.shape[0]
        dev = self._xyz.device
        self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        self.denom = torch.zeros((n, 1), device=dev)
        groups = [
            {"params": [self._xyz], "lr": training_args.position_lr_init * self.spatial_lr_scale, "name": "xyz"},
            {"params": [self._features_dc], "lr": training_args.feature_lr, "name": "f_dc"},
            {"params": [self._features_rest], "lr": training_args.feature_lr / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": training_args.opacity_lr, "name": "opacity"},
            {"params": [self._scaling], "lr": training_args.scaling_lr, "name": "scaling"},
            {"params": [self._rotation], "lr": training_args.rotation_lr, "name": "rotation"},
        ]
        self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)

    def _swap_group_tensor(self, group, new_tensor, state_map):
        old = group["params"][0]
        state = self.optimizer.state.pop(old, None)
        group["params"][0] = nn.Parameter(new_tensor.requires_grad_(True))
        if state is not None:
            state["exp_avg"] = state_map(state["exp_avg"])
            state["exp_avg_sq"] = state_map(state["exp_avg_sq"])
            self.optimizer.state[group["params"][0]] = state
        return group["params"][0]

    def _rebind(self, tensors):
        self._xyz, self._features_dc, self._features_rest = tensors["xyz"], tensors["f_dc"], tensors["f_rest"]
        self._opacity, self._scaling, self._rotation = tensors["opacity"], tensors["scaling"], tensors["rotation"]

    def _current(self):
        return {"xyz": self._xyz, "f_dc": self._features_dc, "f_rest": self._features_rest,
                "opacity": self._opacity, "scaling": self._scaling, "rotation": self._rotation}

    def prune_points(self, mask):
        keep = ~mask
        if self.optimizer is not None:
            out = {}
            for group in self.optimizer.param_groups:
                out[group["name"]] = self._swap_group_tensor(group, group["params"][0][keep],
                                                             lambda s: s[keep])
            self._rebind(out)
        else:
            self._rebind({k: (nn.Parameter(v[keep]) if isinstance(v, nn.Parameter) else v[keep])
                          for k, v in self._current().items()})
        if self.xyz_gradient_accum.numel():
            self.xyz_gradient_accum = self.xyz_gradient_accum[keep]
            self.denom = self.denom[keep]
        if self.max_radii2D.numel():
            self.max_radii2D = self.max_radii2D[keep]
        if getattr(self, "tmp_radii", None) is not None:
            self.tmp_radii = self.tmp_radii[keep]

    def densification_postfix(self, new_xyz, new_features_dc, new_features_rest, new_opacities, new_scaling,
                              new_rotation, new_tmp_radii):
        extra = {"xyz": new_xyz, "f_dc": new_features_dc, "f_rest": new_features_rest, "opacity": new_opacities,
                 "scaling": new_scaling, "rotation": new_rotation}
        if self.optimizer is not None:
            out = {}
            for group in self.optimizer.param_groups:
                add = extra[group["name"]]
                out[group["name"]] = self._swap_group_tensor(
                    group, torch.cat((group["params"][0], add), dim=0),
                    lambda s, add=add: torch.cat((s, torch.zeros_like(add)), dim=0))
            self._rebind(out)
        else:
            self._rebind({k: torch.cat((v, extra[k]), dim=0) for k, v in self._current().items()})
        if getattr(self, "tmp_radii", None) is not None and new_tmp_radii is not None:
            self.tmp_radii = torch.cat((self.tmp_radii, new_tmp_radii))
        n = self.get_xyz.shape[0]
        dev = self._xyz.device
        self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        self.denom = torch.zeros((n, 1), device=dev)
        self.max_radii2D = torch.zeros((n,), device=dev)

    # ---- PLY interchange (property order of SURVEY.md 8f-3) -------------------------------------------------
    def construct_list_of_attributes(self):
        names = ["x", "y", "z", "nx", "ny", "nz"]
        names += [f"f_dc_{i}" for i in range(self._features_dc.shape[1] * self._features_dc.shape[2])]
        names += [f"f_rest_{i}" for i in range(self._features_rest.shape[1] * self._features_rest.shape[2])]
        names.append("opacity")
        names += [f"scale_{i}" for i in range(self._scaling.shape[1])]
        names += [f"rot_{i}" for i in range(self._rotation.shape[1])]
        return names

    def save_ply(self, path):
        from gsworld_amd.ply import write_gaussian_ply

        mkdir_p(os.path.dirname(path))
        write_gaussian_ply(path, self)

    def load_ply(self, path, use_train_test_exp=False):
        from gsworld_amd.ply import read_gaussian_ply

        # the stock loader: trainable nn.Parameters, _opacity (N,1).  GSWorld's Semantic3DGSWrapper overrides load_ply
        # (frozen tensors, (N,1,1) opacity, semantics) -- that flavour is read_gaussian_ply's default.
        read_gaussian_ply(path, self, upstream=True)
        self.active_sh_degree = self.max_sh_degree
