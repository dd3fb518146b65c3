"""Persistent-state forward renderer: the inference path for closed-loop rendering.

``GaussianRasterizer`` (rasterizer.py) keeps upstream's contract: fresh state tensors per call and one host
read-back of ``num_rendered`` in the middle of the frame.  A closed-loop simulator renders the same scene from
the same cameras thousands of times (GSWorld: 402 frames per 200-step episode, gs_world_wrapper.py:176-198,
239-242), so :class:`FrameRenderer` instead owns the three state buffers and the output images, sizes the
binning state from the previous frames (capacity = growth x last R) and never synchronises inside a frame;
overflow of the capacity is detected from the on-device frame header and the frame is re-rendered in exact mode.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from . import _C
from ._lib import GsrFrameStats, GsrSettings, check, lib


@dataclass
class FrameStats:
    num_gaussians: int
    num_visible: int
    num_rendered: int
    overflow: bool

    def algorithmic_bytes(self, width: int, height: int) -> int:
        """B_alg of SURVEY.md 8d: 48 N + 280 V + 64 R + 16 W H."""
        return 48 * self.num_gaussians + 280 * self.num_visible + 64 * self.num_rendered + 16 * width * height


class FrameRenderer:
    def __init__(self, device="cuda", growth: float = 1.25, near_plane: float | None = None):
        self.device = torch.device(device)
        self.growth = growth
        self.near_plane = _C.NEAR_PLANE if near_plane is None else near_plane
        u8 = dict(dtype=torch.uint8, device=self.device)
        self.geom = torch.empty(0, **u8)
        self.binning = torch.empty(0, **u8)
        self.image = torch.empty(0, **u8)
        self.r_capacity = 0
        self._out = None
        self._P = 0

    def _outputs(self, P, H, W):
        if self._out is None or self._out[0].shape != (3, H, W) or self._out[2].shape[0] != P:
            f32 = dict(dtype=torch.float32, device=self.device)
            self._out = (torch.zeros((3, H, W), **f32), torch.zeros((1, H, W), **f32),
                         torch.zeros((P,), dtype=torch.int32, device=self.device))
        return self._out

    def render(self, view, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
               cov3D_precomp=None, bg=None, sh_degree: int = 3, scale_modifier: float = 1.0,
               antialiasing: bool = False, debug: bool = False, exact: bool = False):
        """Enqueue one frame; returns (color (3,H,W), radii (P,), invdepth (1,H,W)) -- tensors owned by the
        renderer and overwritten by the next call.  ``view`` is a :class:`gsworld_amd.camera.ViewParams` on device."""
        dev = self.device
        P = means3D.shape[0]
        H, W = view.image_height, view.image_width
        color, invd, radii = self._outputs(P, H, W)
        self._P = P
        if bg is None:
            bg = torch.zeros(3, device=dev)
        empty = torch.empty(0, device=dev)
        M = shs.shape[1] if shs is not None else 0
        st = GsrSettings(H, W, view.tanfovx, view.tanfovy, float(scale_modifier), int(sh_degree), int(M), 0,
                         int(antialiasing), int(debug), float(self.near_plane))
        cap = 0 if (exact or self.r_capacity == 0) else self.r_capacity
        stats = _C.forward_raw(
            st, bg, means3D, colors_precomp if colors_precomp is not None else empty, opacities,
            scales if scales is not None else empty, rotations if rotations is not None else empty,
            cov3D_precomp if cov3D_precomp is not None else empty, view.world_view_transform,
            view.full_proj_transform, shs if shs is not None else empty, view.camera_center, color, invd, radii,
            self.geom, self.binning, self.image, r_capacity=cap, want_stats=(cap == 0))
        if cap == 0:
            self.r_capacity = max(int(stats.num_rendered * self.growth), 1 << 16)
        return color, radii, invd

    def pack_rgb8(self, color: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """(3,H,W) float image -> (H,W,3) uint8 exactly as GSWorld converts frames
        (gs_world_wrapper.py:268-270: ``(x * 255).clamp(0, 255).to(torch.uint8)``)."""
        _, H, W = color.shape
        if out is None:
            out = torch.empty((H, W, 3), dtype=torch.uint8, device=color.device)
        with torch.cuda.device(color.device):
            check(lib().gsr_pack_rgb8(C.c_void_p(color.data_ptr()), W, H, C.c_void_p(out.data_ptr()),
                                      C.c_void_p(torch.cuda.current_stream(color.device).cuda_stream)))
        return out

    def stats(self) -> FrameStats:
        """V / R / overflow of the last frame (synchronises the current stream)."""
        s = GsrFrameStats()
        with torch.cuda.device(self.device):
            code = lib().gsr_frame_stats(C.c_void_p(self.geom.data_ptr()), C.byref(s),
                                         C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if code not in (0, -4):
            check(code)
        return FrameStats(self._P, int(s.num_visible), int(s.num_rendered), bool(s.overflow))

    def ensure_valid(self, rerender) -> FrameStats:
        """Checks the last frame for capacity overflow; if it overflowed, grows the capacity and calls
        ``rerender()`` (which must call :meth:`render` again with the same arguments)."""
        s = self.stats()
        if s.overflow:
            self.r_capacity = int(s.num_rendered * self.growth)
            rerender()
            s = self.stats()
        return s
