"""``scene.cameras.Camera`` as GSWorld constructs it per frame
(/root/reference/gsworld/mani_skill/utils/wrappers/gs_world_wrapper.py:309-322): keyword arguments
``resolution, colmap_id, R, T, FoVx, FoVy, depth_params, image, invdepthmap, image_name, uid, data_device``.
Exposes what ``gaussian_renderer.render`` reads: image_width/height, FoVx/FoVy, world_view_transform,
full_proj_transform, camera_center (znear = 0.01, zfar = 100).

The fabricated PIL image GSWorld passes only carries the resolution; it is converted lazily so that building a
camera does not cost a 640x480 host->device copy per frame.
"""
import numpy as np
import torch
from torch import nn

from gsworld_amd.camera import ZFAR, ZNEAR, view_params


class Camera(nn.Module):
    def __init__(self, resolution, colmap_id, R, T, FoVx, FoVy, depth_params, image, invdepthmap, image_name, uid,
                 trans=np.array([0.0, 0.0, 0.0]), scale=1.0, data_device="cuda", train_test_exp=False,
                 is_test_dataset=False, is_test_view=False):
        super().__init__()
        self.uid = uid
        self.colmap_id = colmap_id
        self.R = R
        self.T = T
        self.FoVx = FoVx
        self.FoVy = FoVy
        self.image_name = image_name
        try:
            self.data_device = torch.device(data_device)
        except Exception as e:  # noqa: BLE001
            print(e)
            print(f"[Warning] Custom device {data_device} failed, fallback to default cuda device")
            self.data_device = torch.device("cuda")
        self.image_width, self.image_height = int(resolution[0]), int(resolution[1])
        self._image = image
        self._resolution = resolution
        self.alpha_mask = None
        self.invdepthmap = None
        self.depth_reliable = False
        if invdepthmap is not None:
            self.invdepthmap = torch.as_tensor(np.asarray(invdepthmap), dtype=torch.float32)[None].to(self.data_device)
            self.depth_reliable = depth_params is not None
        self.zfar = ZFAR
        self.znear = ZNEAR
        self.trans = trans
        self.scale = scale
        vp = view_params(np.asarray(R), np.asarray(T), FoVx, FoVy, self.image_width, self.image_height, trans, scale)
        # upstream puts the three matrices on the GPU whatever `data_device` says (`...transpose(0, 1).cuda()`): they
        # are rasterizer arguments, only the images follow data_device (CPU-only hosts, i.e. the unit tests, keep them
        # on the CPU)
        dev = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        self.world_view_transform = vp.world_view_transform.to(dev)
        self.projection_matrix = (self.world_view_transform.new_zeros(4, 4))
        from gsworld_amd.camera import get_projection_matrix

        self.projection_matrix = get_projection_matrix(ZNEAR, ZFAR, FoVx, FoVy).transpose(0, 1).to(dev)
        self.full_proj_transform = vp.full_proj_transform.to(dev)
        self.camera_center = vp.camera_center.to(dev)

    @property
    def original_image(self):
        """(3,H,W) float image in [0,1] on the data device (upstream converts eagerly; here on first use)."""
        if isinstance(self._image, torch.Tensor):
            return self._image
        from utils.general_utils import PILtoTorch

        img = PILtoTorch(self._image, self._resolution)[:3, ...].clamp(0.0, 1.0).to(self.data_device)
        self._image = img
        return img


class MiniCam:
    def __init__(self, width, height, fovy, fovx, znear, zfar, world_view_transform, full_proj_transform):
        self.image_width = width
        self.image_height = height
        self.FoVy = fovy
        self.FoVx = fovx
        self.znear = znear
        self.zfar = zfar
        self.world_view_transform = world_view_transform
        self.full_proj_transform = full_proj_transform
        view_inv = torch.inverse(self.world_view_transform)
        self.camera_center = view_inv[3][:3]
