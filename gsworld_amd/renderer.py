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
    overflow_frames: int = 0  # frames on this renderer's state that overflowed since its buffers were allocated
    truncated: bool = False   # a cooperative quadrant of this frame's compositor timed out (GSR_E_TRUNCATED): pixels are wrong
    coop_timeouts: int = 0    # such quadrants in all frames on this state

    def algorithmic_bytes(self, width: int, height: int) -> int:
        """B_alg of SURVEY.md 8d: 48 N + 280 V + 64 R + 16 W H."""
        return 48 * self.num_gaussians + 280 * self.num_visible + 64 * self.num_rendered + 16 * width * height


class FrameRenderer:
    def __init__(self, device="cuda", growth: float = 1.25, near_plane: float | None = None,
                 forward_only: bool = False, want_radii: bool = True, min_capacity: int = 1 << 16,
                 bound_capacity: bool = False, overflow_mirror: bool = False, want_float: bool = True):
        """``forward_only``: inference frames (GsrSettings.forward_only, include/gsr.h): the image is bit-identical, but
        nothing a backward would read is written and the instances are binned per super-tile of 2 x 1 tiles -- the state buffers
        are then no input for ``gsr_backward`` and :meth:`stats` counts super-tile instances.  ``want_radii=False``
        (forward_only only): the (P,) radii array is not written either; :meth:`render` returns ``None`` for it.
        ``growth`` / ``min_capacity``: the binning capacity of the no-sync frames is
        ``max(growth x R of the frame that sized it, min_capacity)`` instances (4 B each).
        ``bound_capacity``: size the instance list by the bound NO frame can exceed -- P x tiles -- whenever that fits
        the budget of :func:`gsworld_amd._C.nosync_capacity` (8 GiB and a quarter of the free memory; 7 GB at 1.47 M
        Gaussians, 640 x 480): no exact-mode first frame, no overflow, hence no flag to read back --
        :attr:`bounded` then says that :meth:`ensure_valid` (a host synchronisation) is not needed.  For a single
        renderer that serves call after call (the drop-in ``render()``); not for dozens of lanes.
        ``want_float=False`` (forward_only only; every call must then pass ``rgb8_out``): the float colour / inverse-depth
        images are not written at all -- a caller that keeps GSWorld's uint8 frame only (gs_world_wrapper.py:266-270) saves
        16 bytes per pixel of stores nobody reads; :meth:`render` returns ``None`` for both.
        ``overflow_mirror``: after every frame the header's overflow count is copied (8 bytes, asynchronously, also
        inside a captured graph) into pinned host memory: :meth:`overflows_seen` then tells WITHOUT a synchronisation
        how many frames on this state have exceeded their capacity so far, as of the last copy that has landed --
        what lets a loop that never synchronises notice an overflowed frame a few steps later instead of at its end."""
        self.device = torch.device(device)
        if self.device.index is None and self.device.type == "cuda":
            # an unindexed device never equals a tensor's `cuda:0`: resolve it once (multi-GPU processes: the CURRENT
            # device at construction is this renderer's device from then on)
            self.device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        self.forward_only = bool(forward_only)
        self.want_radii = bool(want_radii) or not self.forward_only
        self.want_float = bool(want_float) or not self.forward_only
        self.growth = growth
        self.min_capacity = int(min_capacity)
        self.bound_capacity = bool(bound_capacity)
        self.bounded = False   # the current capacity is P x tiles: the last frame cannot have overflowed
        self._bound_for = None
        self.near_plane = _C.NEAR_PLANE if near_plane is None else near_plane
        u8 = dict(dtype=torch.uint8, device=self.device)
        self.geom = torch.empty(0, **u8)
        self.binning = torch.empty(0, **u8)
        self.image = torch.empty(0, **u8)
        self.r_capacity = 0
        self._out = None
        self._P = 0
        self._mirror = torch.zeros(2, dtype=torch.int32).pin_memory() if overflow_mirror and torch.cuda.is_available() else None
        # The frame's capacity check writes the mirror itself (GsrOutputs.overflow_mirror: the pinned words' device-visible
        # address); round 5 copied the header's last 8 bytes behind every frame -- two copies per closed-loop step, ~6 us of
        # its stream by the kernel trace.  (No device-visible address: the copy it is.)
        self._mirror_np = self._mirror.numpy() if self._mirror is not None else None
        self._mirror_dev = None
        if self._mirror is not None:
            import ctypes as C

            from ._lib import lib
            d = C.c_void_p()
            if lib().gsr_pinned_device_address(C.c_void_p(self._mirror.data_ptr()), C.byref(d)) == 0 and d.value:
                self._mirror_dev = int(d.value)
        self.overflows_handled = 0  # (callers that re-render on overflow keep their own tally against overflows_seen)

    def _capacity_for(self, num_rendered: int) -> int:
        return max(int(num_rendered * self.growth), self.min_capacity, 1 << 16)

    def _outputs(self, P, H, W):
        if self._out is None or self._out[0].shape != (3, H, W) or self._out[2].shape[0] != P:
            f32 = dict(dtype=torch.float32, device=self.device)
            self._out = (torch.zeros((3, H, W), **f32), torch.zeros((1, H, W), **f32),
                         torch.zeros((P,), dtype=torch.int32, device=self.device))
        return self._out

    def render(self, view, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
               cov3D_precomp=None, bg=None, sh_degree: int = 3, scale_modifier: float = 1.0,
               antialiasing: bool = False, debug: bool = False, exact: bool = False, shs_rest=None,
               param_space: int = 0, rgb8_out=None, parts=None, outputs=None, layout=None):
        """Enqueue one frame; returns (color (3,H,W), radii (P,), invdepth (1,H,W)) -- tensors owned by the
        renderer and overwritten by the next call.  ``view`` is a :class:`gsworld_amd.camera.ViewParams` on device.
        ``shs_rest``: pass the model's two SH parameters as they are stored, ``shs=features_dc`` (P,1,3) and
        ``shs_rest=features_rest`` (P,M-1,3), instead of concatenating them for every frame.
        ``param_space``: OR of ``gsworld_amd._lib.RAW_OPACITY / RAW_SCALES / RAW_ROTATIONS`` -- the corresponding
        arguments are the model's RAW parameters (logits, log scales, un-normalised quaternions) and are activated
        inside preprocess (no sigmoid / exp / normalize passes per frame).
        ``rgb8_out``: optional (H,W,3) uint8 tensor that receives GSWorld's uint8 frame conversion directly from the
        compositing kernel (same bytes as :meth:`pack_rgb8` of the returned colour image).
        ``parts``: ``(labels (P,) float32, lut int32, table (K,17) float32, rescale (K,) uint8 | None)`` -- this frame's
        rigid transform of the labelled Gaussians, applied inside preprocess: ``means3D`` / ``rotations`` / ``scales``
        are then the BASE model and no transformed copy is ever written (:class:`gsworld_amd.transform.FusedPartTransform`
        builds the tuple; bit-identical to transforming first).
        ``layout``: ``(cull_blocks, orig_index | None)`` of :class:`gsworld_amd.layout.SceneLayout` -- block bounds by
        which preprocess skips the blocks of the model no tile can see, and, for a model stored in the layout's Morton
        order, the original numbering (radii, lists and depth ties stay those of the original model; bit-identical frames).
        ``outputs``: optional caller-owned ``(color (3,H,W) f32, invdepth (1,H,W) f32, radii (P,) i32)`` on this device,
        written instead of the renderer's own buffers (every element is written)."""
        call, color, radii, invd = self._prepare(
            view, means3D, opacities, shs=shs, colors_precomp=colors_precomp, scales=scales, rotations=rotations,
            cov3D_precomp=cov3D_precomp, bg=bg, sh_degree=sh_degree, scale_modifier=scale_modifier,
            antialiasing=antialiasing, debug=debug, exact=exact, shs_rest=shs_rest, param_space=param_space,
            rgb8_out=rgb8_out, parts=parts, outputs=outputs, layout=layout)
        cap = call["r_capacity"]
        stats = _C.forward_raw(want_stats=(cap == 0), **call)
        self._finish(cap, stats)
        return color, radii, invd

    def _prepare(self, view, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                 cov3D_precomp=None, bg=None, sh_degree: int = 3, scale_modifier: float = 1.0,
                 antialiasing: bool = False, debug: bool = False, exact: bool = False, shs_rest=None,
                 param_space: int = 0, rgb8_out=None, parts=None, outputs=None, layout=None):
        """Everything of :meth:`render` before the call into the library (same arguments): tensors normalised, outputs
        and capacity chosen.  -> (keyword arguments of ``_C.forward_raw`` for this frame, color, radii, invdepth)."""
        dev = self.device

        def norm(t, what, allow_none=True):
            # the kernels read raw pointers: float32, on this device, dense.  A no-op for tensors that already are (the
            # closed-loop case); anything else -- a transposed view matrix that kept its strides (upstream's
            # `.transpose(0, 1).cuda()`), a camera built with data_device="cpu", float64 parameters -- is converted
            # here instead of being read as garbage.
            if t is None:
                if allow_none:
                    return None
                raise ValueError(f"{what} is required")
            if not isinstance(t, torch.Tensor):
                raise TypeError(f"{what} must be a tensor, got {type(t).__name__}")
            if t.dtype != torch.float32:
                if not t.dtype.is_floating_point:
                    raise TypeError(f"{what} must be a floating-point tensor, got {t.dtype}")
                t = t.to(torch.float32)
            if t.device != dev:
                t = t.to(dev)
            return t if t.is_contiguous() else t.contiguous()

        means3D = norm(means3D, "means3D", False)
        opacities = norm(opacities, "opacities", False)
        shs, shs_rest = norm(shs, "shs"), norm(shs_rest, "shs_rest")
        colors_precomp, cov3D_precomp = norm(colors_precomp, "colors_precomp"), norm(cov3D_precomp, "cov3D_precomp")
        scales, rotations, bg = norm(scales, "scales"), norm(rotations, "rotations"), norm(bg, "bg")
        view_m = norm(view.world_view_transform, "view.world_view_transform", False)
        proj_m = norm(view.full_proj_transform, "view.full_proj_transform", False)
        campos = norm(view.camera_center, "view.camera_center", False)
        if means3D.dim() != 2 or means3D.shape[1] != 3:
            raise ValueError("means3D must have dimensions (num_points, 3)")
        P = means3D.shape[0]
        H, W = view.image_height, view.image_width
        if outputs is not None:
            color, invd, radii = outputs
            rad = radii if radii is not None else torch.empty((P,), dtype=torch.int32, device=dev)
            if radii is None and self.want_radii:
                raise ValueError("outputs: radii may only be None on a renderer built with want_radii=False")
            if (color.shape != (3, H, W) or invd.shape != (1, H, W) or rad.shape != (P,) or color.dtype != torch.float32
                    or invd.dtype != torch.float32 or rad.dtype != torch.int32
                    or not (color.is_contiguous() and invd.is_contiguous() and rad.is_contiguous())
                    or color.device != dev or invd.device != dev or rad.device != dev):
                raise ValueError("outputs must be dense (3,H,W) float32, (1,H,W) float32, (P,) int32 tensors on the "
                                 "renderer's device")
        elif not self.want_float:
            if rgb8_out is None:
                raise ValueError("a renderer built with want_float=False writes the uint8 frame only: pass rgb8_out")
            if self._out is None or self._out[2].shape[0] != P:
                none = torch.empty(0, dtype=torch.float32, device=dev)
                self._out = (none, none, torch.zeros((P if self.want_radii else 0,), dtype=torch.int32, device=dev))
            color, invd, radii = self._out
        else:
            color, invd, radii = self._outputs(P, H, W)
        if not self.want_radii:
            radii = None
        self._P = P
        if bg is None:
            bg = torch.zeros(3, device=dev)
        empty = torch.empty(0, device=dev)
        M = shs.shape[1] if shs is not None else 0
        if shs_rest is not None:
            if shs is None or shs.dim() != 3 or shs.shape[1] != 1 or shs_rest.dim() != 3 or shs_rest.shape[0] != P:
                raise ValueError("shs_rest needs shs = features_dc of shape (P,1,3) and shs_rest (P,M-1,3)")
            M = 1 + shs_rest.shape[1]
        st = GsrSettings(H, W, view.tanfovx, view.tanfovy, float(scale_modifier), int(sh_degree), int(M), 0,
                         int(antialiasing), int(debug), float(self.near_plane))
        if self.bound_capacity and self._bound_for != (P, H, W):
            bound = _C.nosync_capacity(P, H, W, device=dev)
            self._bound_for, self.bounded = (P, H, W), bound is not None
            self.r_capacity = bound if bound is not None else 0
        cap = self.r_capacity if self.bounded else (0 if (exact or self.r_capacity == 0) else self.r_capacity)
        call = dict(
            settings=st, background=bg, means3D=means3D, colors=colors_precomp if colors_precomp is not None else empty,
            opacity=opacities, scales=scales if scales is not None else empty,
            rotations=rotations if rotations is not None else empty,
            cov3D_precomp=cov3D_precomp if cov3D_precomp is not None else empty, viewmatrix=view_m, projmatrix=proj_m,
            sh=shs if shs is not None else empty, campos=campos, out_color=color, out_invdepth=invd, radii=radii,
            geomBuffer=self.geom, binningBuffer=self.binning, imgBuffer=self.image, r_capacity=cap, sh_rest=shs_rest,
            param_space=param_space, rgb8_out=rgb8_out, parts=parts, forward_only=self.forward_only, layout=layout,
            overflow_mirror=(self._mirror_dev or 0) if not self.bounded else 0)
        if not self.want_float and outputs is None:
            color = invd = None  # (what the caller gets back; the call holds the empty tensors = NULL images)
        return call, color, radii, invd

    def _finish(self, cap: int, stats=None):
        """Bookkeeping behind a frame: capacity from an exact frame's count, the overflow mirror copy."""
        if cap == 0:
            self.r_capacity = self._capacity_for(stats.num_rendered)
        if self._mirror is not None and self._mirror_dev is None and not self.bounded and self.geom.numel() >= 256:
            # (of_magic, overflow_frames): the last two words of the 256-byte frame header (csrc/gsr_internal.h GsrHeader)
            self._mirror.copy_(self.geom[248:256].view(torch.int32), non_blocking=True)

    def overflows_seen(self) -> int:
        """Frames on this state whose instance count exceeded the capacity, as far as the host has been told (no
        synchronisation; needs ``overflow_mirror=True``): the count the device had written when the most recent
        mirror copy that has COMPLETED was taken."""
        if self._mirror is None:
            raise RuntimeError("FrameRenderer was built without overflow_mirror=True")
        magic, count = int(self._mirror_np[0]), int(self._mirror_np[1])
        return count if (magic & 0xFFFFFFFF) == 0x0F10F10F else 0

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
        if code not in (0, -4, -5):
            check(code)
        return FrameStats(self._P, int(s.num_visible), int(s.num_rendered), bool(s.overflow), int(s.overflow_frames),
                          bool(s.truncated), int(s.coop_timeouts))

    def ensure_valid(self, rerender) -> FrameStats:
        """Checks the last frame for capacity overflow; if it overflowed, grows the capacity and calls
        ``rerender()`` (which must call :meth:`render` again with the same arguments)."""
        s = self.stats()
        if s.overflow or s.truncated:
            if s.overflow:
                self.r_capacity = self._capacity_for(s.num_rendered)
            rerender()
            s = self.stats()
            _raise_if_truncated([s])
        return s


def _raise_if_truncated(stats) -> None:
    """A frame whose compositor reported a timed-out cooperative quadrant (``GsrFrameStats.truncated``) shows wrong pixels;
    the validity checks re-render it once -- the hand-off cannot time out by construction, so a second time is a fault of the
    library or the device, never something to render on with."""
    if any(s.truncated for s in stats):
        raise RuntimeError("libgsr_hip: a cooperative quadrant of the compositor timed out again on the re-rendered frame "
                           f"(GSR_E_TRUNCATED; {sum(s.coop_timeouts for s in stats)} such quadrants on these states): the "
                           "frame is invalid.  GSWORLD_AMD_TUNING=render_split=3 renders without cooperative quadrants")


class MultiCameraRenderer:
    """All cameras of one simulation step rendered concurrently.

    GSWorld renders its sensor cameras one after the other inside ``_render_gsworld``
    (gs_world_wrapper.py:239-267: ``for cam_name, cam_param in self.camera_params.items()``).  At 640x480 most kernels
    of a frame are launch/latency-bound (the 10-kernel binning chain keeps ~1/4 of the chip busy), so two cameras of
    the same Gaussians overlap almost perfectly when each gets its own HIP stream and its own renderer state: the
    step's render latency is ~one frame, not ``num_cameras`` frames.  Results are bit-identical to sequential
    :class:`FrameRenderer` calls (same kernels, disjoint state).

    The Gaussian tensors are only READ on the side streams; they must stay alive until the caller's stream has
    passed the join at the end of :meth:`render` (true for the per-step buffers of a closed loop).
    """

    def __init__(self, num_cameras: int, device="cuda", batched: bool = True, **renderer_kw):
        """``batched`` (default): the frames of a step go through ``gsr_forward_batch`` -- ONE set of launches on the
        caller's stream whose grids span the frames (include/gsr.h; up to 8 frames per set) -- instead of one complete
        pipeline per frame on its own HIP stream.  Same kernels, disjoint state: the frames are bit-identical either
        way.  Frames that cannot share launches (the exact-mode frame that sizes a lane, A/B selectors) run one after
        the other on the caller's stream.  ``False``: the stream-per-frame path of rounds 2-4.
        Attributes a caller may set before the first :meth:`render`: ``set_frames`` (frames per set of launches, default and
        maximum 8 = ``GSR_MAX_FRAMES_PER_LAUNCH``) and ``max_set_streams`` (sets of one image size in flight at once, default
        4): a step with more frames than ``set_frames`` goes as several sets on streams of their own, forked from and joined
        to the caller's stream (capturable); ``max_set_streams = 1`` runs them one after the other in one call."""
        self.device = torch.device(device)
        self.batched = bool(batched)
        self.lanes = [FrameRenderer(self.device, **renderer_kw) for _ in range(num_cameras)]
        self.streams = [torch.cuda.Stream(self.device) for _ in range(num_cameras)] if not self.batched else []
        self._set_streams: list = []  # (batched: one per image size beyond the first, created when a step has several)
        self.max_set_streams = 4      # sets of ONE image size in flight at once (16 environments: 16.9 k with three, 17.6 k with four)
        self.set_frames = _C.MAX_FRAMES_PER_LAUNCH  # frames per set of launches
        self.last_pack = None

    def rerun(self, pack_caps) -> None:
        """Enqueues a step again from the argument pack :meth:`render` left in ``last_pack`` -- same tensors, same settings,
        same state: what a loop whose per-step values live in device buffers does every step, without the per-frame Python
        of :meth:`render` (normalising tensors, building settings) in front of the launches."""
        pack, caps = pack_caps
        with torch.cuda.device(self.device):
            _C.run_packed_batch(pack, self.device)
        for lane, cap in caps:
            lane._finish(cap)

    def render(self, views, means3D, opacities, rgb8_out=None, per_lane=None, **render_kw):
        """``views``: one :class:`gsworld_amd.camera.ViewParams` per camera.  Returns ``[(color, radii, invdepth)]``
        per camera (renderer-owned tensors, overwritten by the next call).  ``rgb8_out``: optional list of (H,W,3)
        uint8 tensors that receive GSWorld's uint8 frame conversion on the same stream as the frame.
        ``per_lane``: optional list of keyword overrides per lane (``means3D``, ``scales``, ``rotations`` ...): the
        frames of one step over several ENVIRONMENTS read environment-specific geometry (slice e of the batched fused
        transform) but share everything the step does not move (SH coefficients, opacity)."""
        if len(views) != len(self.lanes) or (per_lane is not None and len(per_lane) != len(self.lanes)):
            raise ValueError(f"expected {len(self.lanes)} cameras, got {len(views)}")
        if self.batched:
            return self._render_batched(views, means3D, opacities, rgb8_out, per_lane, render_kw)
        cur = torch.cuda.current_stream(self.device)
        outs = []
        for k, (lane, stream, view) in enumerate(zip(self.lanes, self.streams, views)):
            kw = render_kw if per_lane is None else {**render_kw, **per_lane[k]}
            m3d = kw.pop("means3D", means3D) if per_lane is not None else means3D
            stream.wait_stream(cur)  # the step's transformed Gaussians are ready
            with torch.cuda.stream(stream):
                color, radii, invd = lane.render(view, m3d, opacities,
                                                 rgb8_out=rgb8_out[k] if rgb8_out is not None else None, **kw)
            outs.append((color, radii, invd))
        for stream in self.streams:
            cur.wait_stream(stream)  # join: the caller's stream sees every frame
        return outs

    def _render_batched(self, views, means3D, opacities, rgb8_out, per_lane, render_kw):
        self.last_pack = None  # (the argument pack of this step, when ALL its frames went through one gsr_forward_batch call)
        outs, calls, caps = [], [], []
        for k, (lane, view) in enumerate(zip(self.lanes, views)):
            kw = render_kw if per_lane is None else {**render_kw, **per_lane[k]}
            m3d = kw.pop("means3D", means3D) if per_lane is not None else means3D
            call, color, radii, invd = lane._prepare(view, m3d, opacities,
                                                     rgb8_out=rgb8_out[k] if rgb8_out is not None else None, **kw)
            outs.append((color, radii, invd))
            cap = call["r_capacity"]
            if cap == 0:
                # exact mode: the frame reads its instance count back in the middle and sizes the lane from it
                lane._finish(0, _C.forward_raw(want_stats=True, **call))
            else:
                calls.append(call)
                caps.append((lane, cap))
        if calls:
            # frames of ONE image size share their launches; a rig with cameras of several sizes (a wrist camera beside a
            # sensor camera) is several such sets, which have nothing to wait for in each other: each set on a stream of its
            # own, forked from and joined to the caller's (rounds 2-4 overlapped such frames lane by lane; round 5's batched
            # path ran them one after the other on the caller's stream)
            sets: dict = {}
            for call in calls:
                sets.setdefault((call["settings"].image_height, call["settings"].image_width), []).append(call)
            # ... and more frames of one size than one set of launches takes (include/gsr.h GSR_MAX_FRAMES_PER_LAUNCH = 8: five or
            # more environments with two cameras) are several sets as well: gsr_forward_batch would run them one after the other
            # on one stream, where two or three in flight fill the latency-bound stages of each other (the headline's
            # arrangement: one stream 12.8 k, three 15.0 k frames/s).  Up to `max_set_streams` streams (four), sets dealt in turn.
            chunks = []
            for group in sets.values():
                for i in range(0, len(group), self.set_frames):
                    chunks.append(group[i:i + self.set_frames])
            with torch.cuda.device(self.device):
                if len(chunks) == 1 or (len(sets) == 1 and self.max_set_streams <= 1):
                    pack = _C.pack_batch(calls)
                    _C.run_packed_batch(pack, self.device)
                    if len(calls) == len(self.lanes):
                        self.last_pack = (pack, caps)
                else:
                    cur = torch.cuda.current_stream(self.device)
                    n_st = min(len(chunks), max(1, self.max_set_streams)) if len(sets) == 1 else len(chunks)
                    while len(self._set_streams) < n_st:
                        self._set_streams.append(torch.cuda.Stream(self.device))
                    used = self._set_streams[:n_st]
                    for st in used:
                        st.wait_stream(cur)
                    for i, group in enumerate(chunks):
                        with torch.cuda.stream(used[i % n_st]):
                            _C.forward_batch_raw(group, device=self.device)
                    for st in used:
                        cur.wait_stream(st)
            for lane, cap in caps:
                lane._finish(cap)
        return outs

    def ensure_valid(self, rerender) -> list:
        """Overflow check of every lane's last frame (synchronises); ``rerender()`` must repeat the step's
        :meth:`render` call.  Returns the per-camera :class:`FrameStats`."""
        stats = [lane.stats() for lane in self.lanes]
        if any(s.overflow or s.truncated for s in stats):
            for lane, s in zip(self.lanes, stats):
                if s.overflow:
                    lane.r_capacity = lane._capacity_for(s.num_rendered)
            rerender()
            stats = [lane.stats() for lane in self.lanes]
            _raise_if_truncated(stats)
        return stats
