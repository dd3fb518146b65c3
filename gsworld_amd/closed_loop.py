"""Render side of GSWorld's closed loop (BASELINE.json configs[2]) without the wrapper's per-step torch glue.

``GSWorldWrapper.step`` / ``reset`` (/root/reference/gsworld/mani_skill/utils/wrappers/gs_world_wrapper.py:176-198)
do two things after the physics step: ``transform_gs_perlink`` (``:110-162``: one ``transform_gaussians`` per robot link
and tracked actor) and ``_render_gsworld`` (``:232-275``: per camera and per environment a deep copy of the model, the
masked write-back of every moved part, ``render()``, uint8 conversion).  :class:`ClosedLoopRenderer` is that pair for a
scene held on one MI355X:

* the pose table of the step is packed on the device (``gsr_pack_part_transforms``), one fused pass moves every labelled
  Gaussian (``gsr_transform_gaussians_batch``; ``E`` environments at once),
* all ``E x C`` frames of the step are enqueued on their own HIP streams with their own renderer state
  (:class:`gsworld_amd.renderer.MultiCameraRenderer`), the compositor writes GSWorld's uint8 frames itself,
* the whole GPU side of a step can be captured once and replayed as one hipGraph.

Returns what ``_render_gsworld`` returns: ``{camera name: uint8 (num_envs, H, W, 3)}``.

SAPIEN / PhysX do not run on a headless GPU box, so when the loop is measured (``bench.py``,
``tools/closed_loop_surrogate.py``) :mod:`gsworld_amd.rollouts` stands in for the simulator (link poses of a seeded
random-action rollout from forward kinematics of the reference's xarm6 URDF, pushed through :func:`part_poses_from_sim`).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _C
from . import transform as tf
from ._lib import FRAME_KEPT, RAW_ROTATIONS, RAW_SCALES, model_version
from .renderer import MultiCameraRenderer


def _to_mirror(dst: torch.Tensor, dst_np, src: torch.Tensor) -> None:
    """``dst.copy_(src.to(float32).reshape(dst.shape))`` for the host mirror; a float32 host tensor that needs no
    conversion goes through numpy (same memory, a fraction of a torch op's dispatch)."""
    if src.dtype == torch.float32 and src.device.type == "cpu" and not src.requires_grad and src.is_contiguous() and \
            src.numel() == dst_np.size:
        np.copyto(dst_np, src.numpy().reshape(dst_np.shape))
    else:
        dst.copy_(src.detach().to("cpu", torch.float32).reshape(dst.shape))


class ClosedLoopRenderer:
    def __init__(self, raw, part_labels: dict, cameras: dict, scaled_parts=(), num_envs: int = 1, device="cuda",
                 background=None, fuse_transform: bool = True, growth: float = 2.0, bound_capacity="auto",
                 layout: bool = True, min_capacity: int | None = None, share_model_of=None, batched: bool = True,
                 keep_float: bool = False, block_cache: bool = True, tile_reuse: bool = True):
        """``raw``: :class:`gsworld_amd.scenes.RawGaussians` (or any object with the same raw parameter tensors, e.g. a
        merged semantic model's ``_xyz`` ... under those names); ``part_labels``: part name -> semantic label(s), in
        the order the pose matrices will arrive; ``cameras``: name -> :class:`gsworld_amd.camera.ViewParams`;
        ``scaled_parts``: the tracked actors (their log-scales are rewritten per step as the reference does).
        ``fuse_transform`` (default): the step's rigid transforms are applied INSIDE each frame's preprocess
        (``GsrInputs.part_*``): every (environment, camera) frame reads the one base model plus that environment's pose
        table, and no transformed copy of the model is written -- per step that saves a 60 B/Gaussian pass per
        environment (88 MB at 1.47 M) and, for ``num_envs`` > 1, the (E,P,.) buffers themselves.  ``False``: one
        ``gsr_transform_gaussians_batch`` pass per step, frames read its outputs (same bytes out: tests).
        ``growth``: every lane's binning capacity is ``max(growth x R, 2 N)`` instances, ``R`` the instance count of
        the frame that sized it (:meth:`reset`, or the re-render after an overflow).  An arm sweeping past a wrist
        camera changes that count far more than a fixed camera ever sees (0.39 M .. 1.46 M over the xarm6 random-action
        rollout, against 0.71 M at reset): the margin costs a few tens of MB per lane and keeps that rollout clear of
        the limit; :meth:`overflow_frames` says, without a per-step sync, whether a rollout stayed clear.
        ``bound_capacity`` (``"auto"`` / True / False): size every lane's list by the bound no frame can exceed, N x
        tiles (7 GB per lane at 1.47 M Gaussians, 640 x 480), so that a frame CANNOT overflow whatever the arm does.
        ``"auto"``: when all lanes together stay below a quarter of the device's free memory and 64 GiB -- one
        environment with two or three cameras on a 288 GB MI355X; larger batches keep the ``growth`` rule.
        ``layout`` (with ``fuse_transform``): the loop keeps its OWN copy of the model in Morton order per part and size
        class, with block bounds (:mod:`gsworld_amd.layout`): every frame's per-Gaussian pass skips the blocks of the
        model that the step's poses put outside the camera's frustum (two thirds of them from a sensor camera).  Same
        frames, bit for bit (tests/test_layout_gpu.py, tests/test_closed_loop_gpu.py).
        ``share_model_of``: another loop over the SAME model on the same device whose (laid-out) model tensors this one
        reads instead of making its own copy (:class:`PipelinedClosedLoop`); everything per step -- poses, cameras,
        renderer states, frames, graph -- stays its own.
        ``keep_float``: the loop returns GSWorld's uint8 frames and, by default, writes nothing else -- the float colour /
        inverse-depth images of its lanes (16 bytes per pixel nobody reads) are left out (``GsrOutputs``: NULL images for an
        inference frame with ``out_rgb8``).  True keeps them in ``multi.lanes[k]._out`` (tests that compare float colour).
        ``tile_reuse`` (with ``block_cache``; default): ``self.frames`` are the loop's own buffers and nobody else writes
        them (READ the returned frames, copy them if you want to draw into them): a 16 x 16 tile that no recomputed Gaussian
        touches under an unchanged camera and background is left as the previous step composited it (include/gsr.h
        GSR_FRAME_KEPT; csrc/render.hip "tile reuse") -- from a fixed sensor camera everything but the tiles the robot and the
        tracked objects cover, now or a step ago.  Same frames, bit for bit.  ``False`` for a caller that writes into them.
        ``batched`` (default): the E x C frames of a step go through ``gsr_forward_batch`` -- one set of launches on the
        step's stream whose grids span the frames -- instead of one pipeline per frame on its own stream
        (:class:`gsworld_amd.renderer.MultiCameraRenderer`); same frames bit for bit (tests/test_batch_gpu.py)."""
        self.device = torch.device(device)
        dev = self.device
        self.num_envs = int(num_envs)
        self.names = list(cameras.keys())
        self._given_cameras = [cameras[n] for n in self.names]
        self.cameras = self._given_cameras  # (image sizes; replaced by device-resident copies once the staging vector exists)
        g = lambda a, b: getattr(raw, a) if hasattr(raw, a) else getattr(raw, b)  # noqa: E731
        other = share_model_of
        if other is not None:
            if other.device != dev or bool(other.layout is not None) != bool(layout and fuse_transform):
                raise ValueError("share_model_of: the other loop must live on the same device and use the same layout option")
            self.xyz, self.rotation, self.scaling = other.xyz, other.rotation, other.scaling
            self.features_dc, self.features_rest, self.opacity = other.features_dc, other.features_rest, other.opacity
            self.layout, semantics = other.layout, other._semantics
            if other.layout is not None:
                self.perm = other.perm
        else:
            self.xyz = g("xyz", "_xyz").detach().to(dev, torch.float32).contiguous()
            self.rotation = g("rotation", "_rotation").detach().to(dev, torch.float32).contiguous()
            self.scaling = g("scaling", "_scaling").detach().to(dev, torch.float32).contiguous()
            self.features_dc = g("features_dc", "_features_dc").detach().to(dev, torch.float32).contiguous()
            self.features_rest = g("features_rest", "_features_rest").detach().to(dev, torch.float32).contiguous()
            # opacity is never moved (new_opacity=None at both call sites of the wrapper): activate it once
            self.opacity = torch.sigmoid(g("opacity", "_opacity").detach().to(dev, torch.float32).reshape(-1, 1)).contiguous()
            semantics = g("semantics", "_semantics")
            self.layout = None
        if other is None and layout and fuse_transform:
            from .layout import SceneLayout

            # A permuted copy of the model (orig_index) is only taken by inference frames on the default sort / placement
            # path (gsr_forward rejects it elsewhere: api.hip make_plan): tile grids up to 16384 tiles and 256 tiles
            # wide, no A/B selector that leaves that path.  Where a camera cannot have that, the loop keeps the model in
            # the caller's order and hands over the block bounds alone (valid for any order, any path; rarely tight).
            reorder = all(self._takes_permuted_model(c, int(self.xyz.shape[0])) for c in self.cameras)
            L = SceneLayout.build(self.xyz, self.scaling, self.rotation,
                                  labels=semantics.detach().to(dev, torch.float32).reshape(-1),
                                  param_space=RAW_SCALES | RAW_ROTATIONS, reorder=reorder, features_dc=self.features_dc,
                                  features_rest=self.features_rest, opacity=self.opacity)
            a = L.arrays
            self.xyz, self.scaling, self.rotation = a["means3D"], a["scales"], a["rotations"]
            self.features_dc, self.features_rest, self.opacity = a["features_dc"], a["features_rest"], a["opacity"]
            semantics = a["labels"]
            self.layout = L.layout
            self.perm = L.perm  # (position in the loop's arrays -> number in the caller's model)
        self._semantics = semantics
        self.op = tf.FusedPartTransform(part_labels, semantics.to(dev), scaled_parts=scaled_parts)
        self.rescaled = len(tuple(scaled_parts)) > 0
        if self.num_envs > 1:
            # A part with EXACTLY num_envs Gaussians takes a degenerate branch in the reference (gs_utils.py:331
            # `rot_mat.size(0) == xyz.size(0)`: Gaussian g is rotated by environment g's matrix; the wrapper's
            # `shape[0] == num_envs` write-back then broadcasts row i over the part -- recorded from the reference
            # itself in tests/golden/wrapper_glue.npz).  That is an accident of shapes, not a behaviour anyone relies
            # on; it is not mirrored here: every part moves rigidly under its environment's pose.  Say so, loudly.
            lab = semantics.reshape(-1).long()
            for name, labels in part_labels.items():
                ids = torch.as_tensor(labels if isinstance(labels, (list, tuple)) else [labels], device=lab.device)
                if int(torch.isin(lab, ids.long()).sum()) == self.num_envs:
                    import warnings

                    warnings.warn(f"part {name!r} has exactly num_envs = {self.num_envs} Gaussians: the reference's "
                                  "shape tests mis-broadcast such a part; it is moved rigidly here", stacklevel=2)
        self.fuse_transform = bool(fuse_transform)
        self.K = len(self.op.names)
        H, W = self.cameras[0].image_height, self.cameras[0].image_width
        self.frames = {n: torch.zeros((self.num_envs, c.image_height, c.image_width, 3), dtype=torch.uint8, device=dev)
                       for n, c in zip(self.names, self.cameras)}
        self.bg = torch.zeros(3, device=dev) if background is None else background.to(dev, torch.float32)
        # the loop never differentiates a frame: inference frames (GsrSettings.forward_only), no radii array
        lanes = self.num_envs * len(self.cameras)
        if bound_capacity == "auto":
            per_lane = max(4 * int(self.xyz.shape[0]) * ((c.image_width + 15) // 16) * ((c.image_height + 15) // 16)
                           for c in self.cameras)
            free = torch.cuda.mem_get_info(dev)[0] if dev.type == "cuda" and torch.cuda.is_available() else 0
            bound_capacity = lanes * per_lane <= min(free // 4, 64 << 30)
        # (... and keeps the uint8 frames only: where the frames take the compositor that can leave the float images out --
        #  inference frames on the default path, the library's own answer -- they are not written either)
        from ._lib import plan_query

        plans = [plan_query(c.image_width, c.image_height, int(self.xyz.shape[0]), forward_only=True) for c in self.cameras]
        no_float = not keep_float and all(p is not None and p["super"] for p in plans)
        self.multi = MultiCameraRenderer(lanes, dev, forward_only=True, want_radii=False, want_float=not no_float, growth=growth,
                                         min_capacity=(2 * int(self.xyz.shape[0]) if min_capacity is None else int(min_capacity)),
                                         bound_capacity=bool(bound_capacity), overflow_mirror=True, batched=batched)
        # Six frames and more per step (three environments with two cameras) go as at least TWO sets of launches on streams of
        # their own (MultiCameraRenderer: up to three in flight), at most eight frames each: the latency-bound stages of one
        # set are filled by the other's.  Same box, configs[2]'s surrogate / the arm-shaped one, frames/s enqueued ahead:
        # 4 environments 14.0 -> 14.6 k / 17.8 -> 18.0 k (policy in the loop 13.5 -> 14.1 / 16.8 -> 17.5), 8 environments
        # 14.2 -> 16.5 k / 17.9 -> 21.4 k, 16 environments 14.1 -> 16.9 k / 17.3 -> 22.0 k.  Fewer frames stay one set (two
        # environments as 2 + 2: 11.6 -> 11.4 k).
        if lanes >= 6:  # (three environments as 3 + 3 frames: 12.95 -> 13.2 k, policy in the loop 12.4 -> 12.8 k)
            self.multi.set_frames = min(8, (lanes + 1) // 2)
        self.recovered_steps = 0      # steps re-rendered because a lane had overflowed (see step())
        self.late_overflow_frames = 0  # overflowed frames that were only noticed after their step had been returned
        lead = (self.num_envs, self.K) if self.num_envs > 1 else (self.K,)
        # Everything a step reads that changes from step to step -- part matrices, uniform scales, camera matrices --
        # lives in ONE device vector: `matrices`, `scales` and the cameras' tensors are views into it.  Values handed over
        # as HOST tensors (the usual case: a CPU simulator, a test) are collected in a host mirror and go up in one copy
        # per step (round 4: five small copies, 5 us each on the step's stream, 8 % of a step); device tensors (ManiSkill
        # link poses on the GPU) are copied view by view, device to device.
        nm, ns = self.num_envs * self.K * 16, self.num_envs * self.K
        self._seg = {"poses": (0, nm + (ns + 3) // 4 * 4)}
        off = self._seg["poses"][1]
        for n in self.names:
            self._seg[n] = (off, off + 36)  # world_view (16) | full_proj (16) | centre (3) + pad
            off += 36
        self._stage = torch.zeros(off, device=dev)
        self._host = torch.zeros(off)
        pin = dev.type == "cuda" and torch.cuda.is_available()
        self._ring = [torch.zeros(off).pin_memory() if pin else torch.zeros(off) for _ in range(8)]
        self._ring_ev, self._ring_k, self._ring_waited = [None] * len(self._ring), 0, -1
        self._on_gpu = pin
        self._dirty, self._stale = set(), set()  # segments changed on the host / last written from a device tensor

        def views(buf):
            v = {"matrices": buf[:nm].view(*lead, 4, 4), "scales": buf[nm:nm + ns].view(lead)}
            for n in self.names:
                o = self._seg[n][0]
                v[n] = (buf[o:o + 16].view(4, 4), buf[o + 16:o + 32].view(4, 4), buf[o + 32:o + 35])
            return v

        self._hv, dv = views(self._host), views(self._stage)
        # (the same host memory as numpy arrays: what a step writes into the mirror and the ring are a dozen copies of a
        #  few hundred bytes, ~0.4 us each through numpy against 1.5-2 us through a torch op -- with the policy in the loop
        #  the host's share of a step is on the critical path: round 6, tools/host_step_profile.py)
        self._hv_np = {k: (tuple(x.numpy() for x in v) if isinstance(v, tuple) else v.numpy()) for k, v in self._hv.items()}
        self._host_np = self._host.numpy()
        self._ring_np = [r.numpy() for r in self._ring]
        self._hv["matrices"].copy_(torch.eye(4).expand(*lead, 4, 4))
        self._hv["scales"].fill_(1.0)
        from .camera import ViewParams

        cams_dev = []
        for n, cam in zip(self.names, self._given_cameras):
            for dst, src in zip(self._hv[n], (cam.world_view_transform, cam.full_proj_transform, cam.camera_center)):
                dst.copy_(src.detach().to("cpu", torch.float32))
            cams_dev.append(ViewParams(cam.image_width, cam.image_height, cam.FoVx, cam.FoVy, *dv[n]))
        self.cameras = cams_dev
        self._stage.copy_(self._host)
        # device-resident pose buffers: what a GPU simulator hands over (ManiSkill link poses are device tensors)
        self.matrices, self.scales = dv["matrices"], dv["scales"]
        # The 17-float pose table(s) the frames read (fuse_transform) live in one persistent buffer and are packed WHERE THE
        # POSES ARRIVE, not inside the step: host poses by the kernel that brings the step's host values to the device
        # (gsr_stage_step reads the pinned slot directly: one launch instead of an H2D copy + the pack kernel), device
        # poses by gsr_pack_part_transforms right behind their device-to-device copy (set_poses).
        self._table = None
        self._ring_dev = None  # device-visible addresses of the pinned slots (gsr_pinned_device_address), resolved once
        if self.fuse_transform and pin:
            import ctypes as C

            from ._lib import lib

            self._table = self.op.pack_on_device(self.matrices, self.scales)
            addr = []
            for slot in self._ring:
                d = C.c_void_p()
                if lib().gsr_pinned_device_address(C.c_void_p(slot.data_ptr()), C.byref(d)) != 0:
                    addr = None  # (not device-visible: the steps' host values go up by copies, the table by its own kernel)
                    break
                addr.append(d.value)
            self._ring_dev = addr
        # The model is this object's own copy (laid out at construction, never written afterwards): the frames may keep the
        # blocks of 256 Gaussians whose camera and part pose are the previous frame's bit for bit (include/gsr.h
        # GSR_MODEL_VERSION; csrc/preprocess.hip prep_block_cached) -- under a fixed sensor camera everything that is not a
        # robot link or a tracked object.  A random version per object: a state buffer the allocator hands from one loop to
        # another never vouches for the other's model.  ``block_cache=False``: every block recomputed every frame.
        import random

        self._model_version = model_version(random.getrandbits(24) | 1) if (block_cache and self.fuse_transform and
                                                                            self.layout is not None) else 0
        if self._model_version != 0 and tile_reuse and not keep_float:
            self._model_version |= FRAME_KEPT
        self._graph = None    # the captured step (host values staged OUTSIDE it: a copy / a launch between two replays) ...
        self._graphs = None   # ... or one captured step per ring slot, each staging its slot itself (see capture())
        self._pack = None     # the step's argument pack (MultiCameraRenderer.last_pack): eager steps without the Python
        self.eager_when_ahead = True  # step(ensure=False) issues the launches one by one instead of replaying the graph
        self._stage_fn = getattr(_C._ext, "stage_host_values", None) if _C._ext is not None else None
        self._sync_fn = getattr(_C._ext, "sync_current_stream", None) if (_C._ext is not None and dev.type == "cuda") else None
        self._dev_index = dev.index if dev.index is not None else (torch.cuda.current_device() if dev.type == "cuda" else 0)
        self._mat_shape = tuple(self.matrices.shape)
        self._cam_meta = {n: (self._seg[n][0], c.image_width, c.image_height, c.FoVx, c.FoVy)
                          for n, c in zip(self.names, self.cameras)}
        self.eager_when_waited = False  # ... step(ensure=True) too (A/B: the graph's one submission against eleven launches)
        self.image_size = (H, W)

    @staticmethod
    def _takes_permuted_model(cam, P: int) -> bool:
        """Will a frame of this camera over a model of ``P`` Gaussians be an inference frame on the default sort /
        placement path (the only frames that take ``GsrInputs.orig_index``)?  The library's own decision
        (``gsr_plan_query`` = csrc/api.hip make_plan, asked on the host): tile grids up to 16384 tiles and 256 tiles wide,
        no A/B selector that leaves the path, and a model the sample sort takes (up to 8 388 608 Gaussians -- beyond that
        the frames fall back to the LSD radix depth sort, which knows nothing of a permutation)."""
        from ._lib import plan_query

        plan = plan_query(cam.image_width, cam.image_height, P, permuted=True, forward_only=True)
        return plan is not None and bool(plan["infer"])

    # ---- one step ------------------------------------------------------------------------------------------------
    def _gpu_step(self):
        """Everything the GPU does per step: pose table, fused transform, E x C frames (reads self.matrices / scales)."""
        E, C = self.num_envs, len(self.cameras)
        if not (self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            self._flush()
        if self.fuse_transform:
            # (the pose table was packed when the poses arrived: _flush / set_poses)
            parts = self.op.parts_of_table(self._table) if self._table is not None else \
                self.op.parts(self.matrices, self.scales)
            views, outs, per_lane = [], [], []
            for e in range(E):
                for c in range(C):
                    views.append(self.cameras[c])
                    outs.append(self.frames[self.names[c]][e])
                    per_lane.append(dict(parts=parts if E == 1 else parts[e]))
            self.multi.render(views, self.xyz, self.opacity, rgb8_out=outs, shs=self.features_dc,
                              shs_rest=self.features_rest, scales=self.scaling, rotations=self.rotation,
                              param_space=RAW_SCALES | RAW_ROTATIONS | self._model_version, bg=self.bg, per_lane=per_lane,
                              layout=self.layout)
            return
        if self.rescaled:
            xyz, rot, scaling = self.op.apply(self.xyz, self.rotation, self.matrices, self.scales, scaling=self.scaling)
        else:
            xyz, rot = self.op.apply(self.xyz, self.rotation, self.matrices, self.scales)
            scaling = self.scaling
        # lane e * C + c renders environment e from camera c; the transformed quaternions keep their norm (reference
        # semantics) and the scales stay logs: preprocess activates both on load
        views, outs = [], []
        for e in range(E):
            for c in range(C):
                views.append(self.cameras[c])
                outs.append(self.frames[self.names[c]][e])
        if E == 1:
            self.multi.render(views, xyz, self.opacity, rgb8_out=outs, shs=self.features_dc,
                              shs_rest=self.features_rest, scales=scaling, rotations=rot,
                              param_space=RAW_SCALES | RAW_ROTATIONS, bg=self.bg)
        else:
            per_lane = [dict(means3D=xyz[e], scales=(scaling[e] if scaling.dim() == 3 else scaling), rotations=rot[e])
                        for e in range(E) for _ in range(C)]
            self.multi.render(views, None, self.opacity, rgb8_out=outs, shs=self.features_dc,
                              shs_rest=self.features_rest, param_space=RAW_SCALES | RAW_ROTATIONS, bg=self.bg,
                              per_lane=per_lane)

    def _flush(self):
        """Host-side changes since the last flush go up in as few copies as possible: one contiguous run of the staging
        vector from the first to the last changed segment, unless a segment in between was last written from a DEVICE
        tensor (its host mirror is stale: the run is cut around it).  Every copy reads from its own pinned slot of a
        ring, so the caller may overwrite its tensors -- and this object its mirror -- right away."""
        if not self._dirty:
            return
        order = ["poses"] + self.names
        runs, cur = [], None
        first = min(order.index(n) for n in self._dirty)
        last = max(order.index(n) for n in self._dirty)
        for n in order[first:last + 1]:
            if n in self._stale and n not in self._dirty:
                cur = None
                continue
            lo, hi = self._seg[n]
            if cur is None:
                cur = [lo, hi]
                runs.append(cur)
            else:
                cur[1] = hi
        for lo, hi in runs:
            k = self._slot_acquire()
            slot = self._ring[k]
            slot[lo:hi].copy_(self._host[lo:hi])
            if not self._stage_poses(k, lo, hi):
                self._stage[lo:hi].copy_(slot[lo:hi], non_blocking=True)
            self._slot_release(k, False)
        self._stale -= self._dirty
        self._dirty.clear()

    def _stage_poses(self, k: int, lo: int, hi: int) -> bool:
        """A run of the staging vector that holds this step's poses goes up through ``gsr_stage_step``: the kernel reads
        pinned slot ``k`` itself, writes the device twin and packs the pose table(s) in the same launch.  -> False: not
        such a run (cameras only), or no pose table to pack -- the caller copies."""
        if self._table is None or self._ring_dev is None or lo != 0 or "poses" not in self._dirty:
            return False
        self._launch_stage(k, hi)
        return True

    def _launch_stage(self, k: int, n: int):
        """``gsr_stage_step`` of the first ``n`` floats of ring slot ``k`` on the current stream (capturable)."""
        import ctypes as C

        from ._lib import check, lib

        nm, ns = self.num_envs * self.K * 16, self.num_envs * self.K
        with torch.cuda.device(self.device):
            check(lib().gsr_stage_step(n, C.c_void_p(self._ring_dev[k]), C.c_void_p(self._stage.data_ptr()), ns, 0, nm,
                                       C.c_void_p(self._table.data_ptr()),
                                       C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def _slot_stage(self) -> int:
        """The host mirror into the next pinned ring slot (what the step's staging kernel reads)."""
        k = self._slot_acquire()
        np.copyto(self._ring_np[k], self._host_np)
        self._dirty.clear()
        return k

    def _stage_host_values_fast(self, matrices, scales, cameras):
        """This step's poses and cameras -- plain host float32 tensors, the usual case -- into the host mirror and the next
        ring slot in ONE call into the compiled binding (csrc_torch/ext.cpp stage_host_values) instead of set_poses +
        set_cameras + the ring copy: 13 -> ~4 us of host work that sits on the critical path of a waited-for step.
        Returns the slot, or None when anything is not of that kind (nothing written: the general path takes over)."""
        segs = []
        if matrices is not None:
            # (device tensors, other dtypes, strided views: the compiled side says no and nothing is written)
            if tuple(matrices.shape) != self._mat_shape or (scales is not None and scales.numel() != self.num_envs * self.K):
                return None
            segs.append((0, matrices))
            if scales is not None:
                segs.append((self.num_envs * self.K * 16, scales))
        if cameras:
            for name, cam in cameras.items():
                meta = self._cam_meta.get(name)
                if meta is None:
                    return None  # (set_cameras raises the KeyError)
                o, w, h, fx, fy = meta
                if cam.image_width != w or cam.image_height != h or abs(cam.FoVx - fx) > 1e-9 or abs(cam.FoVy - fy) > 1e-9:
                    return None  # (set_cameras raises the ValueError)
                wv, fp, cc = cam.world_view_transform, cam.full_proj_transform, cam.camera_center
                if wv.numel() != 16 or fp.numel() != 16 or cc.numel() != 3:
                    return None
                segs += [(o, wv), (o + 16, fp), (o + 32, cc)]
        # (the slot is taken before the write: its last reader has finished by then)
        ring_k, ring_ev, waited = self._ring_k, list(self._ring_ev), self._ring_waited
        k = self._slot_acquire()
        if not self._stage_fn(self._host, self._ring[k], segs):
            self._ring_k, self._ring_ev, self._ring_waited = ring_k, ring_ev, waited  # (slot handed back)
            return None
        self._dirty.clear()
        return k

    def set_poses(self, matrices: torch.Tensor, scales: torch.Tensor | None = None):
        """This step's part poses ((K,4,4) or (E,K,4,4), host or device; + uniform scales) for the persistent device
        buffers the (possibly captured) step reads.  Host tensors are taken over at once (the caller may reuse them) and
        uploaded with the step's other host values in one copy; device tensors are copied device to device."""
        if tuple(matrices.shape) != tuple(self.matrices.shape):
            raise ValueError(f"expected poses of shape {tuple(self.matrices.shape)}, got {tuple(matrices.shape)}")
        if matrices.is_cuda or (scales is not None and scales.is_cuda):
            self._flush()  # (host values set earlier keep their order with respect to this write)
            self.matrices.copy_(matrices.to(torch.float32), non_blocking=True)
            if scales is not None:
                self.scales.copy_(scales.to(torch.float32).reshape(self.scales.shape), non_blocking=True)
            if self._table is not None:
                self.op.pack_on_device(self.matrices, self.scales)  # (into self._table: the buffer is persistent)
            self._stale.add("poses")
            return
        if "poses" in self._stale and scales is None:
            # the poses were last written from device tensors: the mirror's scales are read back once (a host
            # synchronisation on the change from device to host poses only -- a GPU simulator that falls back to host
            # poses in mid-rollout keeps going)
            self._hv["scales"].copy_(self.scales.detach().to("cpu"))
        _to_mirror(self._hv["matrices"], self._hv_np["matrices"], matrices)
        if scales is not None:
            _to_mirror(self._hv["scales"], self._hv_np["scales"], scales)
        self._dirty.add("poses")

    def set_cameras(self, cameras: dict):
        """This step's camera poses: ``{name: ViewParams}`` for any subset of the cameras given at construction -- the
        wrapper recomputes them from the simulator's sensor parameters on every render
        (gs_world_wrapper.py:238 ``self.gs_cam = self.cam_maniskill2gs(self.base_env.get_sensor_params(), ...)``), which
        is how a wrist-mounted camera follows the arm.  The matrices are copied INTO the device tensors the (possibly
        captured) step reads; image size and field of view are part of the launch and must not change."""
        for name, cam in cameras.items():
            if name not in self.names:
                raise KeyError(f"unknown camera {name!r} (known: {self.names})")
            mine = self.cameras[self.names.index(name)]
            if (cam.image_width, cam.image_height) != (mine.image_width, mine.image_height) or \
                    abs(cam.FoVx - mine.FoVx) > 1e-9 or abs(cam.FoVy - mine.FoVy) > 1e-9:
                raise ValueError(f"camera {name!r}: image size / field of view are fixed at construction")
            src = (cam.world_view_transform, cam.full_proj_transform, cam.camera_center)
            if any(t.is_cuda for t in src):
                self._flush()
                for dst, t in zip((mine.world_view_transform, mine.full_proj_transform, mine.camera_center), src):
                    dst.copy_(t.to(torch.float32), non_blocking=True)
                self._stale.add(name)
            else:
                for dst, dst_np, t in zip(self._hv[name], self._hv_np[name], src):
                    _to_mirror(dst, dst_np, t)
                self._dirty.add(name)

    def step(self, matrices: torch.Tensor | None = None, scales: torch.Tensor | None = None,
             cameras: dict | None = None, ensure: bool = False) -> dict:
        """-> {camera name: uint8 (num_envs, H, W, 3)} -- renderer-owned tensors, overwritten by the next step.

        A lane whose instance list is sized from earlier frames (no ``bound_capacity``) can be outgrown by a later frame
        -- an arm swinging past a wrist camera triples the count -- and such a frame shows the background only.  That is
        never silent here: every frame copies its state's overflow count to pinned host memory
        (``FrameRenderer(overflow_mirror=True)``, 8 bytes, also under graph replay), and this method compares the counts
        it can see with the ones it has handled, WITHOUT synchronising:

        * ``ensure=True`` -- what a closed loop does anyway, since the policy reads the frames before it acts: wait for
          this step's frames, and if a lane overflowed, grow it, re-render the step and re-capture the graph.  The
          frames returned are always valid.
        * ``ensure=False`` (throughput mode): no wait.  An overflow of an EARLIER step shows up here as soon as its
          mirror copy has landed (normally within a step or two); the loop then recovers as above -- the current step
          is rendered correctly -- and :attr:`late_overflow_frames` counts the frames that had already been handed out
          invalid."""
        pre_k = None
        if self._graphs is not None and not self._stale and self._stage_fn is not None:
            pre_k = self._stage_host_values_fast(matrices, scales, cameras)  # (None: not plain host tensors -- the general path)
        if pre_k is None:
            if matrices is not None:
                self.set_poses(matrices, scales)
            if cameras:
                self.set_cameras(cameras)
        if self._graphs is not None and self._stale:
            self.capture()  # (device tensors arrived since the capture: their values must not be staged over)
        if (not ensure or self.eager_when_waited) and self.eager_when_ahead and self._pack is not None and \
                self._table is not None and self._ring_dev is not None and not self._stale and \
                (self._graphs is not None or self._graph is None):
            # Steps enqueued AHEAD of the device (nobody waits for this step's frames before the next is issued): the step's
            # eleven launches one by one, from the argument pack of the last eager step.  Two graph replays in a row leave the
            # device idle for ~14 us between them (kernel trace: the last kernel of one to the first of the next, however far
            # ahead the host is); launches queue back to back.  8.58 k against 8.36 k frames/s on the configs[2] surrogate.
            # With the policy in the loop the graph wins -- one submission instead of eleven in front of every wait: 7.67 k
            # against 7.25 k -- and `ensure=True` takes it.
            k = pre_k if pre_k is not None else self._slot_stage()
            pack, caps = self._pack
            if pack[0] == "pack":
                # (the staging kernel and the frames' launches in ONE call into the compiled binding: csrc_torch/ext.cpp StepPack)
                nm, ns = self.num_envs * self.K * 16, self.num_envs * self.K
                with torch.cuda.device(self.device):
                    pack[1].run_staged(int(self._stage.numel()), self._ring_dev[k], self._stage.data_ptr(), ns, 0, nm,
                                       self._table.data_ptr())
                for lane, cap in caps:
                    lane._finish(cap)
            else:
                self._launch_stage(k, int(self._stage.numel()))
                self.multi.rerun(self._pack)
            self._slot_release(k, ensure)
        elif self._graphs is not None:
            # this step's host values travel INSIDE its graph: the whole mirror into the next pinned slot (a host copy of a
            # kilobyte), then the replay of the graph that was captured reading that slot
            k = pre_k if pre_k is not None else self._slot_stage()
            self._graphs[k].replay()
            self._slot_release(k, ensure)
        elif self._graph is not None:
            self._flush()  # this step's host values: one copy / launch on the step's stream, ahead of the replay
            self._graph.replay()
        elif self._pack is not None and self.fuse_transform and self._table is not None:
            self._flush()
            self.multi.rerun(self._pack)  # (the frames' arguments have not changed: device buffers, written by the flush)
        else:
            self._gpu_step()
            self._pack = self.multi.last_pack if self.fuse_transform and self._table is not None else None
        if ensure:
            if self._sync_fn is not None:
                self._sync_fn(self._dev_index)
            else:
                torch.cuda.current_stream(self.device).synchronize()
        self._check_overflow(late=not ensure)
        return self.frames

    # The pinned ring: slot k is read by the device during the step that staged it, and written again eight steps later.
    # An event behind EVERY step is a marker the queue stops at: ~6 us between a step's last kernel and the next step's
    # first by the kernel trace (round 6).  So only every fourth step records one, and a slot's writer waits for the first
    # event at or after the step that last read the slot -- at most three steps later, still five steps back.
    _EV_EVERY = 4

    def _slot_acquire(self) -> int:
        n = len(self._ring)
        s = self._ring_k
        self._ring_k += 1
        last_reader = s - n
        if last_reader >= 0 and self._on_gpu:
            e = last_reader + ((self._EV_EVERY - 1 - last_reader) % self._EV_EVERY)  # first recording step >= last_reader
            ev = self._ring_ev[e % n]
            if ev is not None and ev[0] == e:
                ev[1].synchronize()
            elif ev is None or ev[0] < last_reader:
                # (no marker covers the slot's last reader -- e.g. that step ended in a wait of its own: ensure=True)
                if self._ring_waited < last_reader:
                    torch.cuda.current_stream(self.device).synchronize()
                    self._ring_waited = s - 1
        return s % n

    def _slot_release(self, k: int, waited: bool):
        s = self._ring_k - 1
        if waited:
            self._ring_waited = s  # (the caller waits for this step's frames: every slot up to here is free after that)
        elif s % self._EV_EVERY == self._EV_EVERY - 1 and self._on_gpu:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._ring_ev[s % len(self._ring)] = (s, ev)

    def _check_overflow(self, late: bool):
        pending = 0
        for lane in self.multi.lanes:
            if lane._mirror is not None and not lane.bounded:
                seen = lane.overflows_seen()
                if seen < lane.overflows_handled:
                    # the lane's state was reallocated (its count restarts at 0) or its mirror has not landed yet: the
                    # tally restarts with it -- a stale, larger "handled" would hide this lane's next overflows, and
                    # summed over the lanes would cancel another lane's
                    lane.overflows_handled = seen
                pending += seen - lane.overflows_handled
        if pending <= 0:
            return
        recapture = self._graph is not None or self._graphs is not None
        torch.cuda.synchronize(self.device)
        for lane in self.multi.lanes:
            if lane._mirror is not None and not lane.bounded:
                lane.overflows_handled = lane.overflows_seen()
        # frames of the CURRENT step are re-rendered below; with late=True the earlier ones had been returned already
        seen_now = sum(1 for lane in self.multi.lanes if lane.stats().overflow)
        if late:
            self.late_overflow_frames += max(pending - seen_now, 0)
        self._graph = self._graphs = None
        self._pack = None  # (capacities change below: the pack holds the old ones)
        self._flush_all()
        for lane in self.multi.lanes:
            st = lane.stats()
            if st.overflow:
                lane.r_capacity = lane._capacity_for(st.num_rendered)
        # exact-mode safety net for whatever the grown capacities still do not hold
        self._gpu_step()
        self.multi.ensure_valid(self._gpu_step)
        torch.cuda.synchronize(self.device)
        for lane in self.multi.lanes:
            if lane._mirror is not None and not lane.bounded:
                lane.overflows_handled = lane.overflows_seen()
        self.recovered_steps += 1
        if recapture:
            self.capture()

    def reset(self, matrices: torch.Tensor | None = None, scales: torch.Tensor | None = None) -> dict:
        """First frame(s): exact-mode render that sizes every lane's binning capacity, then the validity check."""
        if matrices is not None:
            self.set_poses(matrices, scales)
        self._pack = None
        self._gpu_step()
        self.multi.ensure_valid(self._gpu_step)
        self._overflows_acknowledged()
        return self.frames

    def _overflows_acknowledged(self):
        """(after a synchronising validity check: what the mirrors show from now on is news)"""
        torch.cuda.synchronize(self.device)
        for lane in self.multi.lanes:
            if lane._mirror is not None and not lane.bounded:
                lane.overflows_handled = lane.overflows_seen()

    def ensure_valid(self):
        """Overflow check of the last step (synchronises); re-renders exactly if a lane's capacity was exceeded.  A
        captured graph is dropped in that case (its capacities are baked in): call :meth:`capture` again."""
        def again():
            self._graph = self._graphs = None
            self._pack = None
            self._flush_all()
            self._gpu_step()
        return self.multi.ensure_valid(again)

    def overflow_frames(self) -> int:
        """How many frames of this loop exceeded their lane's binning capacity since the lane's state was allocated
        (such a frame shows the background only).  Counted on the device by the frames themselves, so a rollout that
        never calls :meth:`ensure_valid` between steps can still tell at its end whether every frame was valid
        (synchronises)."""
        return sum(lane.stats().overflow_frames for lane in self.multi.lanes)

    @property
    def captured(self) -> bool:
        """A step is replayed from a hipGraph (one graph, or one per ring slot with the host values staged inside)."""
        return self._graph is not None or self._graphs is not None

    def _flush_all(self):
        """(after staged graphs: the eager path stages what is dirty only -- everything the host holds is, once)"""
        self._dirty.update(n for n in ["poses"] + self.names if n not in self._stale)

    def capture(self):
        """Captures the GPU side of a step into a hipGraph (call after :meth:`reset`).

        When every per-step value comes from the host (poses and cameras handed over as host tensors: a CPU simulator, the
        rollouts of the tests and of ``bench.py``), the step's host values are staged INSIDE the graph: one graph per pinned
        ring slot, each starting with the ``gsr_stage_step`` launch that reads its slot -- :meth:`step` then writes the
        host mirror into the next slot and replays that slot's graph: nothing is enqueued between two replays (round 5: an
        H2D copy and the pack kernel; an eager launch between two graph replays costs ~14 us of idle device per step by the
        kernel trace).  Values handed over as device tensors are never staged over: such a loop captures ONE graph of the
        frames, and whatever the host still provides goes up in front of each replay."""
        dev = self.device
        self._graph = self._graphs = None
        self._gpu_step()
        self.multi.ensure_valid(self._gpu_step)
        self._pack = self.multi.last_pack if self.fuse_transform and self._table is not None else None
        staged = self._table is not None and self._ring_dev is not None and not self._stale and self.fuse_transform
        n_all = int(self._stage.numel())

        def one(k):
            if k is not None:
                self._launch_stage(k, n_all)
            self._gpu_step()

        graphs = []
        for k in (range(len(self._ring)) if staged else [None]):
            if k is not None:
                self._ring[k].copy_(self._host)
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                one(k)
            torch.cuda.current_stream(dev).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                one(k)
            graphs.append(g)
        if staged:
            self._graphs = graphs
            self._dirty.clear()
        else:
            self._graph = graphs[0]
        self._overflows_acknowledged()
        return graphs[0]


class PipelinedClosedLoop:
    """``depth`` closed loops over ONE copy of the model that take the steps of a rollout in turn, each on its own HIP
    stream with its own poses, cameras, renderer states, frames and graph: step k + 1 is enqueued while step k still
    renders, so ``depth x cameras`` frames are in flight instead of ``cameras``.

    A single :class:`ClosedLoopRenderer` replays one graph per step on one stream: the two frames of a step overlap,
    consecutive steps do not, and at 640 x 480 two frames leave the chip half idle (section 4 of DESIGN.md: 9.7 k frames/s
    with two identical frames in flight against 11.9 k with four).  Whether consecutive steps MAY overlap is the caller's
    matter: a random-action or scripted rollout (BASELINE.json configs[2]: ``gsworld_rand_action_tabletop.py:107-133``
    never looks at its observations), an open-loop replay, a data-collection run with a planner, or several environments
    stepping independently can; a policy that needs frame k before it chooses action k + 1 cannot, and gains nothing
    here.

    :meth:`step` returns the frames of the step it enqueued -- tensors owned by that step's loop, overwritten ``depth``
    steps later.  With ``wait=True`` (default) the CURRENT stream is made to wait for them, so whatever the caller
    enqueues next may read them; poses handed over as DEVICE tensors are read after what the current stream has already
    enqueued (a GPU simulator).  Both are stream dependencies, never host synchronisations; but a current stream that
    waits for step k and then produces the poses of step k + 1 chains the steps together again -- hand host (pinned)
    poses over, or ``wait=False`` and :meth:`wait_for` before reading, to keep them apart.

    Reading a step's frames: enqueue the reads (or a clone) on the stream that waited for them BEFORE the next call of
    :meth:`step`.  That call marks the consumer stream, and the step that reuses the loop -- ``depth`` calls later, on
    another stream -- waits for the mark before it overwrites the frames (write-after-read across streams: the step's
    own stream knows nothing of the caller's reads).  The mark is taken before the consumer stream is made to wait for
    the NEW step, so it orders against the reads only and consecutive steps still overlap."""

    def __init__(self, raw, part_labels: dict, cameras: dict, depth: int = 2, **kw):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        first = ClosedLoopRenderer(raw, part_labels, cameras, **kw)
        self.loops = [first] + [ClosedLoopRenderer(raw, part_labels, cameras, share_model_of=first, **kw)
                                for _ in range(depth - 1)]
        self.device = first.device
        self.streams = [torch.cuda.Stream(self.device) for _ in self.loops]
        self._events = [None] * depth
        self._released = [None] * depth   # per loop: event on its consumer stream, recorded behind the reads of its frames
        self._consumer = [None] * depth   # per loop: the stream that was made to wait for its last step (None: nobody yet)
        self._k = 0

    @property
    def depth(self) -> int:
        return len(self.loops)

    def reset(self, matrices=None, scales=None) -> dict:
        """Sizes every loop's lanes on the reset poses; returns the first loop's frames."""
        cur = torch.cuda.current_stream(self.device)
        out = None
        for loop, st in zip(self.loops, self.streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                f = loop.reset(matrices, scales)
            out = out if out is not None else f
        torch.cuda.synchronize(self.device)
        self._k = 0
        self._released = [None] * len(self.loops)
        self._consumer = [None] * len(self.loops)
        return out

    def capture(self):
        for loop, st in zip(self.loops, self.streams):
            with torch.cuda.stream(st):
                loop.capture()
        torch.cuda.synchronize(self.device)

    def step(self, matrices=None, scales=None, cameras: dict | None = None, wait: bool = True, ensure: bool = False) -> dict:
        i = self._k % len(self.loops)
        self._k += 1
        loop, st = self.loops[i], self.streams[i]
        cur = torch.cuda.current_stream(self.device)
        for j, consumer in enumerate(self._consumer):
            if consumer is not None:
                # whatever the caller has enqueued behind those frames on the stream that was made to wait for them
                rel = torch.cuda.Event()
                rel.record(consumer)
                self._released[j] = rel
                self._consumer[j] = None
        if self._released[i] is not None:  # this loop's frames were handed out `depth` steps ago: their readers first
            st.wait_event(self._released[i])
            self._released[i] = None
        on_device = any(isinstance(t, torch.Tensor) and t.is_cuda for t in (matrices, scales)) or \
            any(c.world_view_transform.is_cuda for c in (cameras or {}).values())
        if on_device:
            st.wait_stream(cur)  # the poses were produced by what the current stream holds
        with torch.cuda.stream(st):
            frames = loop.step(matrices, scales, cameras, ensure=ensure)
            ev = torch.cuda.Event()
            ev.record(st)
        self._events[i] = ev
        if wait:
            cur.wait_event(ev)
            self._consumer[i] = cur
        return frames

    def wait_for(self, frames: dict | None = None, stream=None):
        """Makes ``stream`` (default: current) wait for the step that produced ``frames`` (default: every step enqueued so
        far)."""
        stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        for i, (loop, ev) in enumerate(zip(self.loops, self._events)):
            if ev is not None and (frames is None or frames is loop.frames):
                stream.wait_event(ev)
                self._consumer[i] = stream

    def overflow_frames(self) -> int:
        return sum(loop.overflow_frames() for loop in self.loops)

    def ensure_valid(self):
        return [s for loop in self.loops for s in loop.ensure_valid()]


def part_poses_from_sim(sim2gs_arm: torch.Tensor, link_now: torch.Tensor, link_scan: torch.Tensor, link_offset=None,
                         actor_now: torch.Tensor | None = None, sim2gs_obj: torch.Tensor | None = None,
                         actor_offset: torch.Tensor | None = None, actor_scale: torch.Tensor | None = None):
    """The per-part matrices of one simulation step, from what a ManiSkill env exposes -- the host-side arithmetic of
    ``GSWorldWrapper.transform_gs_perlink`` (gs_world_wrapper.py:114-120, 139-156) in its float32 operation order, so a
    caller holding the simulator state gets exactly the ``rot_mat`` / ``translation`` / ``scale`` the reference hands
    to ``transform_gaussians`` (pinned by tests/golden/wrapper_glue.npz, which records those arguments from the
    reference's own run).

    ``link_now`` (E,L,4,4): ``link.pose.to_transformation_matrix()`` of every robot link; ``link_scan`` (L,4,4): the same
    at the qpos the robot was scanned in (``gs_link_pose_mats``, ``__init__`` :94-103); ``link_offset`` (3,): the
    ``object_offset["xarm_arm"]`` shift added to every link position for xarm robots (:117-119) or None;
    ``actor_now`` (E,A,4,4): tracked actor poses; ``sim2gs_obj`` (A,4,4): their ``sim2gs_object_transforms``;
    ``actor_offset`` (A,3) and ``actor_scale`` (A,): ``object_offset`` / ``object_scale`` of each (zeros / ones when an
    actor has no entry).  -> ``(matrices (E,L+A,4,4), scales (E,L+A))`` in link-then-actor order, the layout
    :class:`ClosedLoopRenderer` takes (links carry scale 1 and are not rescaled)."""
    from .camera import extract_rigid_transform

    f32 = torch.float32
    dev = link_now.device  # (a GPU simulator hands over device tensors: everything is computed where the poses live)
    on = lambda t: torch.as_tensor(t, dtype=f32, device=dev)  # noqa: E731
    sim2gs_arm = on(sim2gs_arm)
    inv_arm = torch.linalg.inv(sim2gs_arm)
    link_mat = link_now.to(f32).clone()
    link_scan = on(link_scan)
    E, L = link_mat.shape[:2]
    if link_offset is not None:
        link_mat[:, :, :3, 3] += on(link_offset)
    mats, scales = [], []
    for k in range(L):  # (:120) sim2gs @ link_now @ inv(link_scan) @ inv(sim2gs), left to right
        mats.append(sim2gs_arm @ link_mat[:, k] @ torch.linalg.inv(link_scan[k]) @ inv_arm)
        scales.append(torch.ones(E, dtype=f32, device=dev))
    if actor_now is not None:
        A = actor_now.shape[1]
        sim2gs_obj = on(sim2gs_obj)
        for a in range(A):
            mat = on(actor_now[:, a]).clone()
            if actor_offset is not None:
                mat[:, :3, 3] += on(actor_offset[a])
            full = sim2gs_arm @ mat @ torch.linalg.inv(sim2gs_obj[a])  # (:147)
            rigid, scale, _, _ = extract_rigid_transform(full)          # (:150)
            mats.append(rigid)
            scales.append(on(scale) * (1.0 if actor_scale is None else on(actor_scale[a])))
    return torch.stack(mats, 1).contiguous(), torch.stack(scales, 1).to(f32).contiguous()


_ROLLOUT_NAMES = ("small_rigid", "random_walk_poses", "xarm6_rollout", "rollout_poses", "xarm6_rollout_parts", "xarm6_parts")


def __getattr__(name):
    """The synthetic rollouts (stand-ins for the simulator) live in :mod:`gsworld_amd.rollouts`; their old names here
    keep resolving."""
    if name in _ROLLOUT_NAMES:
        from . import rollouts

        return getattr(rollouts, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
