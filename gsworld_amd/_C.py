"""``_C`` of the drop-in ``diff_gaussian_rasterization`` package: the three functions the upstream pybind
module exports (rasterize_points.h / ext.cpp), same positional signatures, same return
tuples, same error strings -- implemented over the C ABI of libgsr_hip.so (include/gsr.h).

Caller: ``_RasterizeGaussians`` in :mod:`gsworld_amd.rasterizer`, which GSWorld reaches through
``gaussian_renderer.render`` (/root/reference/gsworld/mani_skill/utils/wrappers/gs_world_wrapper.py:266-267).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import (GsrBuffers, GsrFrameStats, GsrInputs, GsrOutputs, GsrSettings, RESIZE_FN, check, lib)

# The compiled binding (csrc_torch/ext.cpp, built by gsworld_amd/build_ext.py): the `_C` pybind module upstream's
# python package imports.  Used whenever it has been built; the ctypes path below drives the SAME shared library and
# remains for environments without the torch headers (GSWORLD_AMD_CTYPES=1 forces it, e.g. to A/B the host overhead).
import os as _os

_ext = None
if _os.environ.get("GSWORLD_AMD_CTYPES", "") != "1":
    try:
        from . import _C_ext as _ext  # noqa: F401
    except ImportError:
        _ext = None


def _tuning_list(forward_only=None):
    """The A/B selectors of this call.  ``forward_only`` None: a frame whose state a backward will read (the
    ``rasterize_gaussians*`` paths) -- a forced ``TUNING["forward_only"]`` (GSWORLD_AMD_TUNING, an A/B aid for the frame
    renderers) must never turn such a frame into an inference frame: gsr_backward would carve the full state layout
    over a lean buffer."""
    t = _lib.TUNING
    if forward_only is None:
        fo = 0
    else:
        fo = int(t["forward_only"]) if int(t["forward_only"]) >= 0 else int(forward_only)
    return [int(t["binning_path"]), int(t["render_variant"]), int(t["render_blocks_per_cu"]), int(t["depth_sort"]),
            int(t["render_split"]), fo]


# GSWorld's edit of cuda_rasterizer/auxiliary.h (/root/reference/README.md:33).  Module-level so that a stock
# 3DGS caller can restore 0.2 without touching the ABI.
NEAR_PLANE = _lib.GSR_NEAR_PLANE


def _ptr(t: torch.Tensor | None):
    if t is None or t.numel() == 0:
        return None
    return C.c_void_p(t.data_ptr())


def _f32(t: torch.Tensor, device, what: str) -> torch.Tensor:
    if t.numel() == 0:
        return t
    if t.dtype != torch.float32:
        raise RuntimeError(f"{what} must be float32")
    if t.device != device:
        t = t.to(device)
    return t.contiguous()


def _resizer(t: torch.Tensor, header: bool = False):
    """``header``: the geometry state -- storage the allocator hands out may be a freed state of another renderer, whose
    frame header still says how many of ITS frames overflowed: on new storage the count (the header's last two words)
    is zeroed, enqueued on the frame's stream ahead of the frame's first kernel.  The rest of a recycled header is left
    alone on purpose: kept splitters / cuts are tied to model size and layout and checked before use (depthsort.hip),
    and a training loop, which gets the block it freed a step ago back every step, keeps its splitters that way."""
    def fn(_user, nbytes):
        before = t.data_ptr() if t.numel() else 0
        t.resize_(int(nbytes))
        if header and t.data_ptr() != before and nbytes >= 256:
            t[248:256].zero_()
        return t.data_ptr()

    return RESIZE_FN(fn)


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"{what} is on {t.device}: the MI355X rasterizer has no CPU path (tensors must live on a HIP device)")


def forward_raw(settings: GsrSettings, background, means3D, colors, opacity, scales, rotations, cov3D_precomp,
                viewmatrix, projmatrix, sh, campos, out_color, out_invdepth, radii, geomBuffer, binningBuffer,
                imgBuffer, r_capacity: int = 0, want_stats: bool = True, sh_rest=None, param_space: int = 0,
                rgb8_out=None, parts=None, forward_only: bool | None = False, layout=None, overflow_mirror: int = 0):
    """Thin call into gsr_forward with caller-owned output and state tensors (no allocation here).
    ``overflow_mirror``: GsrOutputs.overflow_mirror as an address (0 = not wanted).
    ``sh_rest``: optional features_rest (P,M-1,3); ``sh`` is then features_dc (P,1,3) -- no per-frame concatenation.
    ``param_space``: OR of ``_lib.RAW_*`` -- opacity logits / log scales / un-normalised rotations are activated
    inside preprocess instead of by three torch passes.
    ``parts``: optional ``(labels (P,) float32, lut (L,) int32, table (K,17) float32, rescale (K,) uint8 | None)`` -- the
    per-frame rigid transform of labelled Gaussians applied inside preprocess (GsrInputs.part_*).
    ``forward_only``: GsrSettings.forward_only (inference frame; ``radii`` may then be None); None = a training frame
    (never an inference frame, whatever ``TUNING["forward_only"]`` forces for A/B runs).
    ``layout``: optional ``(cull_blocks (ceil(P/256),8) float32, orig_index (P,) int32 | None)`` -- block bounds for
    view-frustum culling and the original numbering of a permuted model (GsrInputs.cull_blocks / orig_index;
    :mod:`gsworld_amd.layout` builds them)."""
    dev = means3D.device
    if _ext is not None:
        st = settings
        e = torch.empty(0, device=dev)
        nv, nr, ov = _ext.forward_frame(
            st.image_height, st.image_width, st.tanfovx, st.tanfovy, st.scale_modifier, st.sh_degree, st.sh_coeffs,
            bool(st.antialiasing), bool(st.debug), st.near_plane, background, means3D,
            colors if colors is not None else e, opacity, scales if scales is not None else e,
            rotations if rotations is not None else e, cov3D_precomp if cov3D_precomp is not None else e, viewmatrix,
            projmatrix, sh if sh is not None else e, sh_rest if sh_rest is not None else e, campos, out_color,
            out_invdepth, radii if radii is not None else torch.empty(0, dtype=torch.int32, device=dev), geomBuffer,
            binningBuffer, imgBuffer,
            rgb8_out if rgb8_out is not None else torch.empty(0, dtype=torch.uint8, device=dev), int(r_capacity),
            bool(want_stats), int(param_space), _tuning_list(forward_only),
            parts[0] if parts is not None else e, parts[1] if parts is not None else torch.empty(0, dtype=torch.int32, device=dev),
            parts[2] if parts is not None else e,
            parts[3] if (parts is not None and parts[3] is not None) else torch.empty(0, dtype=torch.uint8, device=dev),
            layout[0] if layout is not None else e,
            layout[1] if (layout is not None and layout[1] is not None) else torch.empty(0, dtype=torch.int32, device=dev),
            int(overflow_mirror or 0))
        stats = GsrFrameStats()
        stats.num_visible, stats.num_rendered, stats.overflow = nv, nr, ov
        return stats
    settings, inp, out, buf, _keep = _frame_structs(
        settings, background, means3D, colors, opacity, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, sh,
        campos, out_color, out_invdepth, radii, geomBuffer, binningBuffer, imgBuffer, sh_rest=sh_rest,
        param_space=param_space, rgb8_out=rgb8_out, parts=parts, forward_only=forward_only, layout=layout,
        overflow_mirror=overflow_mirror)
    stats = GsrFrameStats()
    with torch.cuda.device(dev):
        check(lib().gsr_forward(C.byref(settings), C.byref(inp), C.byref(out), C.byref(buf), C.c_int64(r_capacity),
                                C.byref(stats) if want_stats else None, _stream(dev)))
    return stats


def _frame_structs(settings: GsrSettings, background, means3D, colors, opacity, scales, rotations, cov3D_precomp,
                   viewmatrix, projmatrix, sh, campos, out_color, out_invdepth, radii, geomBuffer, binningBuffer,
                   imgBuffer, sh_rest=None, param_space: int = 0, rgb8_out=None, parts=None,
                   forward_only: bool | None = False, layout=None, overflow_mirror: int = 0):
    """The four argument structs of one frame (include/gsr.h) from tensors; the fifth value keeps the resize callbacks
    alive while the structs are in use."""
    settings.forward_only = int(bool(forward_only))
    _lib.apply_tuning(settings, allow_forward_only=forward_only is not None)
    inp = GsrInputs(
        P=means3D.size(0), background=_ptr(background), means3D=_ptr(means3D), shs=_ptr(sh),
        colors_precomp=_ptr(colors), opacities=_ptr(opacity), scales=_ptr(scales), rotations=_ptr(rotations),
        cov3D_precomp=_ptr(cov3D_precomp), viewmatrix=_ptr(viewmatrix), projmatrix=_ptr(projmatrix),
        campos=_ptr(campos), shs_rest=_ptr(sh_rest) if sh_rest is not None else None, param_space=int(param_space))
    if parts is not None:
        labels, lut, table, rescale = parts
        inp.part_labels, inp.part_lut, inp.part_lut_size = _ptr(labels), _ptr(lut), int(lut.numel())
        inp.part_transforms, inp.part_count = _ptr(table), int(table.shape[0])
        inp.part_rescale = _ptr(rescale) if rescale is not None else None
    if layout is not None:
        blocks, orig = layout
        if blocks.numel() != 8 * ((means3D.size(0) + 255) // 256) or blocks.dtype != torch.float32:
            raise ValueError("layout: cull_blocks must be a float32 tensor of shape (ceil(P / 256), 8)")
        if orig is not None and (orig.numel() != means3D.size(0) or orig.dtype != torch.int32):
            raise ValueError("layout: orig_index must be an int32 tensor of shape (P,)")
        inp.cull_blocks = _ptr(blocks)
        inp.orig_index = _ptr(orig) if orig is not None else None
    out = GsrOutputs(_ptr(out_color), _ptr(out_invdepth), _ptr(radii),
                     _ptr(rgb8_out) if rgb8_out is not None else None, int(overflow_mirror) if overflow_mirror else None)
    cbs = (_resizer(geomBuffer, header=True), _resizer(binningBuffer), _resizer(imgBuffer))
    buf = GsrBuffers(cbs[0], None, cbs[1], None, cbs[2], None)
    return settings, inp, out, buf, cbs


MAX_FRAMES_PER_LAUNCH = 8  # include/gsr.h GSR_MAX_FRAMES_PER_LAUNCH


def forward_batch_raw(frames, device=None):
    """B frames of one step through gsr_forward_batch: one set of launches whose grids span the frames (include/gsr.h;
    GSWorld's per-step double loop over cameras and environments, gs_world_wrapper.py:238-267).  ``frames``: a list of
    dicts with the keyword arguments of :func:`forward_raw` (``settings`` ... ``imgBuffer``, ``r_capacity`` > 0 for the
    frames that are to share launches, no ``want_stats``: nothing is read back).  Enqueued on the current stream of the
    frames' device; returns nothing -- ``gsr_frame_stats`` on a frame's geometry state tells V / R / overflow."""
    if not frames:
        return
    dev = frames[0]["means3D"].device if device is None else device
    run_packed_batch(pack_batch(frames), dev)


def pack_batch(frames):
    """The argument pack of :func:`forward_batch_raw` for these frames, built once: a caller that renders the SAME frames
    again and again -- same tensors, same settings, same state buffers: a closed loop whose per-step values live in device
    buffers the kernels read -- keeps it and calls :func:`run_packed_batch` per step (the per-frame Python that builds it
    costs more host time than the step's eleven launches).  With the compiled binding the pack is a ``StepPack``: the argument
    structs are filled ONCE, so every tensor but the three state buffers (which the resize callbacks follow) must keep its
    storage for as long as the pack is used -- build a new pack after replacing or resizing any of them."""
    B = len(frames)
    dev = frames[0]["means3D"].device
    if _ext is not None and hasattr(_ext, "forward_batch"):
        e = torch.empty(0, device=dev)
        ei = torch.empty(0, dtype=torch.int32, device=dev)
        eb = torch.empty(0, dtype=torch.uint8, device=dev)
        packed = []
        for f in frames:
            st, parts, layout = f["settings"], f.get("parts"), f.get("layout")
            o = lambda k, d=e: f.get(k) if f.get(k) is not None else d  # noqa: E731
            packed.append((
                st.image_height, st.image_width, st.tanfovx, st.tanfovy, st.scale_modifier, st.sh_degree, st.sh_coeffs,
                bool(st.antialiasing), bool(st.debug), st.near_plane, f["background"], f["means3D"], o("colors"),
                f["opacity"], o("scales"), o("rotations"), o("cov3D_precomp"), f["viewmatrix"], f["projmatrix"],
                o("sh"), o("sh_rest"), f["campos"], f["out_color"], f["out_invdepth"], o("radii", ei), f["geomBuffer"],
                f["binningBuffer"], f["imgBuffer"], o("rgb8_out", eb), int(f.get("r_capacity", 0)),
                int(f.get("param_space", 0)), _tuning_list(f.get("forward_only", False)),
                parts[0] if parts is not None else e, parts[1] if parts is not None else ei,
                parts[2] if parts is not None else e,
                parts[3] if (parts is not None and parts[3] is not None) else eb,
                layout[0] if layout is not None else e,
                layout[1] if (layout is not None and layout[1] is not None) else ei,
                int(f.get("overflow_mirror") or 0)))
        if hasattr(_ext, "StepPack"):
            return ("pack", _ext.StepPack(packed))  # (the argument structs filled once, kept on the C++ side)
        return ("ext", packed)
    built, caps = [], (C.c_int64 * B)()
    for k, f in enumerate(frames):
        f = dict(f)
        caps[k] = int(f.pop("r_capacity", 0))
        f.pop("want_stats", None)
        built.append(_frame_structs(**f))
    St = (GsrSettings * B)(*[b[0] for b in built])
    In = (GsrInputs * B)(*[b[1] for b in built])
    Out = (GsrOutputs * B)(*[b[2] for b in built])
    Buf = (GsrBuffers * B)(*[b[3] for b in built])
    return ("ctypes", (B, St, In, Out, Buf, caps, built, frames))  # (built / frames: keep callbacks and tensors alive)


def run_packed_batch(pack, device):
    """Enqueues the frames of :func:`pack_batch` on the current stream of ``device``."""
    kind, p = pack
    if kind == "pack":
        p.run()
        return
    if kind == "ext":
        _ext.forward_batch(p)
        return
    B, St, In, Out, Buf, caps = p[:6]
    with torch.cuda.device(device):
        check(lib().gsr_forward_batch(B, St, In, Out, Buf, caps, _stream(device)))


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, antialiasing, debug, sh_rest=None, param_space: int = 0):
    """-> (num_rendered, out_color (3,H,W), radii (P,), geomBuffer, binningBuffer, imgBuffer, out_invdepth (1,H,W)).

    Upstream's positional signature; the extensions are keywords (forward only): ``sh_rest`` -- ``sh`` = features_dc
    (P,1,3), ``sh_rest`` = features_rest (P,M-1,3), read in place instead of a per-frame ``cat``; ``param_space`` --
    OR of ``_lib.RAW_*``: ``opacity`` / ``scales`` / ``rotations`` are the raw parameters and are activated inside
    preprocess."""
    if _ext is not None:
        return _ext.rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, float(scale_modifier),
                                        cov3D_precomp, viewmatrix, projmatrix, float(tan_fovx), float(tan_fovy),
                                        int(image_height), int(image_width), sh, int(degree), campos,
                                        bool(prefiltered), bool(antialiasing), bool(debug), sh_rest, int(param_space),
                                        float(NEAR_PLANE), _tuning_list(None))
    if means3D.ndim != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    if sh_rest is not None and (sh.ndim != 3 or sh.size(1) != 1 or sh_rest.ndim != 3 or sh_rest.size(0) != sh.size(0)):
        raise RuntimeError("sh_rest needs sh = features_dc of shape (num_points, 1, 3)")
    _require_gpu(means3D, "means3D")
    dev = means3D.device
    P = means3D.size(0)
    H, W = int(image_height), int(image_width)
    f32 = dict(dtype=torch.float32, device=dev)
    # (a frame writes every pixel and every radius itself; upstream's zero fill is what P == 0 returns)
    alloc = torch.empty if P != 0 else torch.zeros
    out_color = alloc((3, H, W), **f32)
    out_invdepth = alloc((1, H, W), **f32)
    radii = alloc((P,), dtype=torch.int32, device=dev)
    geomBuffer = torch.empty(0, dtype=torch.uint8, device=dev)
    binningBuffer = torch.empty(0, dtype=torch.uint8, device=dev)
    imgBuffer = torch.empty(0, dtype=torch.uint8, device=dev)
    rendered = 0
    if P != 0:
        M = sh.size(1) if sh.numel() != 0 else 0
        if sh_rest is not None:
            M = 1 + sh_rest.size(1)
        st = GsrSettings(H, W, float(tan_fovx), float(tan_fovy), float(scale_modifier), int(degree), int(M),
                         int(bool(prefiltered)), int(bool(antialiasing)), int(bool(debug)), float(NEAR_PLANE))
        stats = forward_raw(
            st, _f32(background, dev, "background"), _f32(means3D, dev, "means3D"), _f32(colors, dev, "colors"),
            _f32(opacity, dev, "opacity"), _f32(scales, dev, "scales"), _f32(rotations, dev, "rotations"),
            _f32(cov3D_precomp, dev, "cov3D_precomp"), _f32(viewmatrix, dev, "viewmatrix"),
            _f32(projmatrix, dev, "projmatrix"), _f32(sh, dev, "sh"), _f32(campos, dev, "campos"),
            out_color, out_invdepth, radii, geomBuffer, binningBuffer, imgBuffer, r_capacity=0,
            sh_rest=_f32(sh_rest, dev, "sh_rest") if sh_rest is not None else None, param_space=param_space,
            forward_only=None)
        rendered = int(stats.num_rendered)
    return rendered, out_color, radii, geomBuffer, binningBuffer, imgBuffer, out_invdepth


# Largest instance list the no-sync training forward may allocate (bytes; 4 per instance).  288 GB of HBM make the TRUE
# bound affordable for training-sized problems: a Gaussian touches at most every tile, so num_rendered <= P x tiles.
NOSYNC_LIST_BYTES = 8 << 30
_FREE_CACHE = {}


def nosync_capacity(P: int, image_height: int, image_width: int, device=None):
    """Instance capacity with which a frame can never overflow (P x tiles), or None when that list would not fit
    ``NOSYNC_LIST_BYTES`` -- or, with ``device``, a quarter of what is free there -- or the grid is wider than the
    counting placement takes: the caller keeps sizing the list from a read-back."""
    gx, gy = (int(image_width) + 15) // 16, (int(image_height) + 15) // 16
    cap = int(P) * gx * gy
    # (gx * gy > 16384 = GSR_MAX_COUNT_TILES: such grids take the radix placement, whose binning state is 24 B per
    # instance plus sort tables, not 4 -- the byte budget below would be off by 6x)
    if P <= 0 or gx > 256 or gx * gy > 16384 or cap >= (1 << 31) or 4 * cap > NOSYNC_LIST_BYTES:
        return None
    if device is not None:
        # a quarter of what is free, looked up at most every 32nd call per (size, device): hipMemGetInfo is a driver call
        key = (int(P), gx, gy, str(device))
        n, free = _FREE_CACHE.get(key, (0, None))
        if free is None or n % 32 == 0:
            free = torch.cuda.mem_get_info(device)[0]
        _FREE_CACHE[key] = (n + 1, free)
        if 4 * cap > free // 4:
            return None
    if _lib.TUNING["binning_path"] != 0 or _lib.TUNING["depth_sort"] != 0:
        return None  # (A/B paths keep keys / ping-pong sides per instance: their lists are sized exactly)
    return cap


def rasterize_gaussians_nosync(capacity, background, means3D, opacity, scales, rotations, scale_modifier, viewmatrix,
                               projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                               antialiasing, debug, sh_rest=None, param_space: int = 0, colors=None, cov3D_precomp=None):
    """The training forward WITHOUT upstream's host read of num_rendered in the middle of the frame (a ~40 us hole in
    the GPU's timeline per step: D2H copy, host wake-up, allocation, launch): the instance list is sized by ``capacity``
    = :func:`nosync_capacity`, a bound no frame can exceed, so nothing has to be read back.  Same kernels, same image
    and state as :func:`rasterize_gaussians`; returns ``capacity`` where that returns num_rendered (it is what the
    backward carves the binning buffer with)."""
    _require_gpu(means3D, "means3D")
    dev = means3D.device
    P = means3D.size(0)
    H, W = int(image_height), int(image_width)
    f32 = dict(dtype=torch.float32, device=dev)
    out_color, out_invdepth = torch.empty((3, H, W), **f32), torch.empty((1, H, W), **f32)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    geomBuffer, imgBuffer = (torch.empty(0, dtype=torch.uint8, device=dev) for _ in range(2))
    # the instance list is allocated HERE (4 B per instance on the counting placement + one alignment unit; nosync_capacity
    # only hands out capacities for that path): a list that does not fit raises torch.cuda.OutOfMemoryError in Python,
    # where the caller's fallback to exact sizing catches it -- inside the library's resize callback the exception would be
    # swallowed by ctypes and come back as a generic allocation error.  The callback's resize_ to a smaller size keeps
    # this storage.
    binningBuffer = torch.empty(4 * int(capacity) + 512, dtype=torch.uint8, device=dev)
    M = (1 + sh_rest.size(1)) if sh_rest is not None else (sh.size(1) if sh.numel() != 0 else 0)
    st = GsrSettings(H, W, float(tan_fovx), float(tan_fovy), float(scale_modifier), int(degree), int(M), 0,
                     int(bool(antialiasing)), int(bool(debug)), float(NEAR_PLANE))
    e = torch.empty(0, device=dev)
    forward_raw(st, _f32(background, dev, "background"), _f32(means3D, dev, "means3D"),
                _f32(colors, dev, "colors") if colors is not None else e,
                _f32(opacity, dev, "opacity"), _f32(scales, dev, "scales"), _f32(rotations, dev, "rotations"),
                _f32(cov3D_precomp, dev, "cov3D_precomp") if cov3D_precomp is not None else e,
                _f32(viewmatrix, dev, "viewmatrix"), _f32(projmatrix, dev, "projmatrix"), _f32(sh, dev, "sh"),
                _f32(campos, dev, "campos"), out_color, out_invdepth, radii, geomBuffer, binningBuffer, imgBuffer,
                r_capacity=int(capacity), want_stats=False,
                sh_rest=_f32(sh_rest, dev, "sh_rest") if sh_rest is not None else None, param_space=param_space,
                forward_only=None)
    return int(capacity), out_color, radii, geomBuffer, binningBuffer, imgBuffer, out_invdepth


def rasterize_gaussians_backward(background, means3D, radii, colors, opacities, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                 dL_dout_invdepth, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer,
                                 antialiasing, debug, sh_rest=None, param_space=0):
    """-> (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)
    [+ dL_dsh_rest with ``sh_rest``; opacity / scale / rotation gradients w.r.t. the raw parameters with ``param_space``]."""
    if _ext is not None:
        return tuple(_ext.rasterize_gaussians_backward(
            background, means3D, radii, colors, opacities, scales, rotations, float(scale_modifier), cov3D_precomp,
            viewmatrix, projmatrix, float(tan_fovx), float(tan_fovy), dL_dout_color,
            dL_dout_invdepth if dL_dout_invdepth is not None else None, sh, int(degree), campos, geomBuffer, int(R),
            binningBuffer, imageBuffer, bool(antialiasing), bool(debug), sh_rest, int(param_space), float(NEAR_PLANE)))
    from . import _backward

    return _backward.rasterize_gaussians_backward(
        background, means3D, radii, colors, opacities, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
        projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_invdepth, sh, degree, campos, geomBuffer, R,
        binningBuffer, imageBuffer, antialiasing, debug, NEAR_PLANE, sh_rest=sh_rest, param_space=param_space)


def mark_visible(means3D, viewmatrix, projmatrix):
    """-> bool (P,): Gaussians in front of the near plane (upstream markVisible; projmatrix is unused there too)."""
    if _ext is not None:
        return _ext.mark_visible(means3D, viewmatrix, projmatrix, float(NEAR_PLANE))
    _require_gpu(means3D, "means3D")
    dev = means3D.device
    P = means3D.size(0)
    present = torch.zeros((P,), dtype=torch.bool, device=dev)
    if P != 0:
        m3 = _f32(means3D, dev, "means3D")
        vm = _f32(viewmatrix, dev, "viewmatrix")
        with torch.cuda.device(dev):
            check(lib().gsr_mark_visible(P, _ptr(m3), _ptr(vm), C.c_float(NEAR_PLANE), C.c_void_p(present.data_ptr()),
                                         _stream(dev)))
    return present
