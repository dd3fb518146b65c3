"""Typed views into the opaque state buffers of a forward pass (tools / tests; not on the hot path)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import GsrStateView, check, lib


def set_binning_mode(mode: int) -> None:
    """A/B and tests: 1 = global depth sort + counting placement (default), 0 = depth sort + emit + tile-id radix sort,
    2 = unordered binning + per-tile LDS sort, 3 = round 1's counting placement by rank chunks, 4 = placement by chunks
    of the depth order with per-tile rank masks (chunkplace.hip).  Sets the selector every later call of this package carries
    (``GsrSettings.binning_path``); the shared library keeps no state."""
    if mode not in (0, 1, 2, 3, 4):
        raise ValueError("binning mode must be 0, 1, 2, 3 or 4")
    _lib.TUNING["binning_path"] = {1: 0, 0: 1, 2: 2, 3: 3, 4: 4}[mode]  # 3 = counting placement by rank chunks (round 1)


def set_depth_sort(variant: int) -> None:
    """A/B and tests: 0 = sample sort (default), 1 = 3-pass LSD radix sort of the visible Gaussians."""
    if variant not in (0, 1):
        raise ValueError("depth sort variant must be 0 or 1")
    _lib.TUNING["depth_sort"] = variant


def set_render_split(on) -> None:
    """A/B and tests: what the compositor does with the costliest quadrants of the previous frame (GsrSettings.render_split).
    ``False`` / 0 = default: inference frames hand them to cooperative workgroups (three waves cull, one composites);
    ``True`` / 1 = two 8x4 halves on two waves (round 3's experiment); 2 = as 0; 3 = one wave per quadrant, always."""
    mode = int(on)
    if mode not in (0, 1, 2, 3):
        raise ValueError("render_split must be 0 (default), 1 (halves), 2 (cooperative always) or 3 (neither)")
    _lib.TUNING["render_split"] = mode


def set_render_variant(variant: int, blocks_per_cu: int = 0) -> None:
    """A/B and tests: 4 = wave-decoupled culling kernel (default), 0 = LDS-staged per tile (upstream's structure),
    2 = batched tile kernel, 3 = the same with per-quadrant culling; ``blocks_per_cu`` 1..8 sizes the persistent grid
    (0 = keep).  All variants produce bit-identical image state."""
    if variant not in (0, 2, 3, 4) or not 0 <= blocks_per_cu <= 8:
        raise ValueError("variant must be 0, 2, 3 or 4; blocks_per_cu 0..8")
    _lib.TUNING["render_variant"] = {4: 0, 0: 1, 2: 2, 3: 3}[variant]
    if blocks_per_cu > 0:
        _lib.TUNING["render_blocks_per_cu"] = blocks_per_cu


def _view(buf: torch.Tensor, ptr, count: int, dtype: torch.dtype) -> torch.Tensor:
    off = int(ptr) - buf.data_ptr()
    nbytes = count * torch.empty((), dtype=dtype).element_size()
    assert 0 <= off and off + nbytes <= buf.numel(), "state view outside its buffer"
    return buf[off:off + nbytes].view(dtype)


def state_view(P: int, width: int, height: int, num_rendered: int, num_visible: int, geomBuffer: torch.Tensor,
               binningBuffer: torch.Tensor | None, imgBuffer: torch.Tensor | None, r_capacity: int | None = None):
    """Returns a dict of tensors aliasing the state buffers (see GsrStateView in include/gsr.h).

    ``r_capacity`` must be the capacity the forward pass used (``num_rendered`` in exact mode).
    """
    v = GsrStateView()
    cap = num_rendered if r_capacity is None else r_capacity
    check(lib().gsr_state_view(P, width, height, C.c_int64(cap), C.c_void_p(geomBuffer.data_ptr()),
                               C.c_void_p(binningBuffer.data_ptr()) if binningBuffer is not None else None,
                               C.c_void_p(imgBuffer.data_ptr()) if imgBuffer is not None else None, C.byref(v)))
    gx, gy = (width + 15) // 16, (height + 15) // 16
    out = {}
    splat = _view(geomBuffer, v.splat, 12 * P, torch.float32).view(P, 12)
    out["splat"] = splat
    out["means2D"] = splat[:, 0:2]
    out["depths"] = splat[:, 2]
    out["invdepths"] = splat[:, 3]
    out["conic_opacity"] = splat[:, 4:8]
    out["rgb"] = splat[:, 8:11]
    out["cov3D"] = _view(geomBuffer, v.cov3D, 6 * P, torch.float32).view(P, 6)
    out["clamped"] = _view(geomBuffer, v.clamped, 4 * P, torch.uint8).view(P, 4)[:, :3]
    out["tiles_touched"] = _view(geomBuffer, v.tiles_touched, P, torch.int32)
    out["rects"] = _view(geomBuffer, v.rects, 4 * P, torch.int16).view(P, 4)
    if v.depth_order and _lib.TUNING["binning_path"] != 2:  # bin-then-sort builds no global depth order
        out["depth_order"] = _view(geomBuffer, v.depth_order, max(num_visible, 0), torch.int32)
    if binningBuffer is not None and num_rendered > 0:
        out["point_list"] = _view(binningBuffer, v.point_list, num_rendered, torch.int32)
    if imgBuffer is not None and "point_list" in out:
        # tile id of every slot of the point list, reconstructed from the ranges
        r = _view(imgBuffer, v.ranges, 2 * gx * gy, torch.int32).view(gx * gy, 2).to(torch.int64)
        tiles = torch.zeros(num_rendered, dtype=torch.int64, device=r.device)
        nonempty = r[:, 1] > r[:, 0]
        tiles[r[nonempty, 0]] = torch.nonzero(nonempty).squeeze(1)
        out["point_tiles"] = torch.cummax(tiles, 0).values.to(torch.int32)
    if imgBuffer is not None:
        out["ranges"] = _view(imgBuffer, v.ranges, 2 * gx * gy, torch.int32).view(gx * gy, 2)
        out["final_T"] = _view(imgBuffer, v.final_T, width * height, torch.float32).view(height, width)
        out["n_contrib"] = _view(imgBuffer, v.n_contrib, width * height, torch.int32).view(height, width)
    return out


def sort_state(geomBuffer: torch.Tensor) -> dict:
    """What the sample sort of the last frame on this geometry state did with the splitters it found there
    (``gsr_debug_sort_state``; synchronises the current stream): ``blind`` -- taken unchecked, ``fresh`` -- drawn anew from
    samples, neither -- the kept ones checked against samples and kept; ``bad`` -- some depth bucket came out above what
    quantiles of an unchanged scene give; ``trust`` -- consecutive balanced frames on kept splitters before this one;
    ``buckets`` -- depth buckets of the frame; ``stride`` -- 2 when it took every second entry of a kept table;
    ``coop_quads`` -- quadrants the frame's compositor handed to cooperative workgroups; ``near`` -- ``blind`` under a view
    matrix that differs a little from the one the kept table was built under (a camera that moves slowly); ``kept_blocks``
    -- the frame's preprocess left blocks as the previous frame on the state had computed them (the block cache)."""
    out = (C.c_int32 * 8)()
    with torch.cuda.device(geomBuffer.device):
        check(lib().gsr_debug_sort_state(C.c_void_p(geomBuffer.data_ptr()), out,
                                         C.c_void_p(torch.cuda.current_stream(geomBuffer.device).cuda_stream)))
    return dict(blind=bool(out[0]), fresh=bool(out[1]), bad=bool(out[2]), trust=int(out[3]), buckets=int(out[4]),
                stride=int(out[5]), coop_quads=int(out[6]), near=bool(out[7] & 1), kept_blocks=bool(out[7] & 2), kept_tiles=bool(out[7] & 4))
