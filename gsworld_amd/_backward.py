"""Python side of ``_C.rasterize_gaussians_backward`` (upstream rasterize_points.cu
RasterizeGaussiansBackwardCUDA) over gsr_backward of the C ABI."""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import GsrInputs, GsrSettings, check, lib


class GsrBackwardInputs(C.Structure):
    _fields_ = [("dL_dout_color", C.c_void_p), ("dL_dout_invdepth", C.c_void_p), ("radii", C.c_void_p),
                ("num_rendered", C.c_int64), ("geom", C.c_void_p), ("binning", C.c_void_p), ("image", C.c_void_p)]


class GsrGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D",
                                          "dL_dsh", "dL_dscales", "dL_drots", "dL_dconic", "dL_dinvdepths",
                                          "dL_dsh_rest")]


def _bind():
    L = lib()
    if not getattr(L, "_bwd_bound", False):
        L.gsr_backward.restype = C.c_int
        L.gsr_backward.argtypes = [C.POINTER(GsrSettings), C.POINTER(GsrInputs), C.POINTER(GsrBackwardInputs),
                                   C.POINTER(GsrGrads), C.c_void_p]
        L.gsr_selftest_wave_sum.restype = C.c_int
        L.gsr_selftest_wave_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        sizes = (C.c_int32 * 6)()
        L.gsr_abi_sizes(sizes)
        if (sizes[4], sizes[5]) != (C.sizeof(GsrBackwardInputs), C.sizeof(GsrGrads)):
            raise RuntimeError("libgsr_hip.so was built from a different include/gsr.h (backward structs): rebuild it")
        L._bwd_bound = True
    return L


def _ptr(t):
    return None if t is None or t.numel() == 0 else C.c_void_p(t.data_ptr())


def _f32(t, dev):
    if t.numel() == 0:
        return t
    if t.dtype != torch.float32:
        raise RuntimeError("rasterizer tensors must be float32")
    return (t.to(dev) if t.device != dev else t).contiguous()


def rasterize_gaussians_backward(background, means3D, radii, colors, opacities, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                 dL_dout_invdepth, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer,
                                 antialiasing, debug, near_plane, sh_rest=None, param_space=0):
    """``sh_rest`` / ``param_space`` as in :func:`gsworld_amd._C.rasterize_gaussians`: with ``sh_rest`` the SH gradient
    comes back in two parts (dL_dsh is then (P,1,3), and a ninth result dL_dsh_rest (P,M-1,3) is appended); with
    ``param_space`` bits the opacity / scale / rotation gradients are w.r.t. the RAW parameters."""
    L = _bind()
    dev = means3D.device
    P = means3D.size(0)
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    M = sh.size(1) if sh.numel() != 0 else 0
    split = sh_rest is not None
    if split:
        if M != 1:
            raise RuntimeError("sh_rest needs sh = features_dc of shape (P,1,3)")
        M = 1 + sh_rest.size(1)
    # One uninitialised arena sliced into the gradient buffers.  The SH gradients come first (16-byte aligned: the SH
    # backward kernel then writes every word of them itself, through whole lines); behind them, without gaps, the
    # buffers gsr_backward clears -- adjacent buffers cost ONE memset on the stream.
    shapes = [("dL_dsh_rest", (P, M - 1 if split else 0, 3)), ("dL_dsh", (P, 1 if split else M, 3)),
              ("dL_drotations", (P, 4)), ("dL_dcov3D", (P, 6)), ("dL_dmeans3D", (P, 3)), ("dL_dmeans2D", (P, 3)),
              ("dL_dcolors", (P, 3)), ("dL_dscales", (P, 3)), ("dL_dopacity", (P, 1))]
    sizes = [max(1, int(torch.tensor(s).prod())) if 0 not in s else 0 for _, s in shapes]
    steps = [(sz + 3) // 4 * 4 if name in ("dL_dsh_rest", "dL_dsh") else sz for (name, _), sz in zip(shapes, sizes)]
    arena = torch.empty(sum(steps), dtype=torch.float32, device=dev)
    views, off = {}, 0
    for (name, shape), sz, step in zip(shapes, sizes, steps):
        views[name] = arena[off:off + sz].view(shape)
        off += step
    dL_drotations, dL_dsh, dL_dcov3D = (views[k] for k in ("dL_drotations", "dL_dsh", "dL_dcov3D"))
    dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dscales = (views[k] for k in ("dL_dmeans3D", "dL_dmeans2D",
                                                                           "dL_dcolors", "dL_dscales"))
    dL_dopacity = views["dL_dopacity"]
    dL_dsh_rest = views["dL_dsh_rest"]
    if P != 0:
        tensors = [_f32(t, dev) for t in (background, means3D, colors, opacities, scales, rotations, cov3D_precomp,
                                          viewmatrix, projmatrix, sh, campos, dL_dout_color)]
        (background, means3D, colors, opacities, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, sh, campos,
         dL_dout_color) = tensors
        if split:
            sh_rest = _f32(sh_rest, dev)
        dLd = None
        if dL_dout_invdepth is not None and dL_dout_invdepth.numel() != 0:
            dLd = _f32(dL_dout_invdepth, dev)
        st = GsrSettings(H, W, float(tan_fovx), float(tan_fovy), float(scale_modifier), int(degree), int(M), 0,
                         int(bool(antialiasing)), int(bool(debug)), float(near_plane))
        inp = GsrInputs(P=P, background=_ptr(background), means3D=_ptr(means3D), shs=_ptr(sh),
                        colors_precomp=_ptr(colors), opacities=_ptr(opacities), scales=_ptr(scales),
                        rotations=_ptr(rotations), cov3D_precomp=_ptr(cov3D_precomp), viewmatrix=_ptr(viewmatrix),
                        projmatrix=_ptr(projmatrix), campos=_ptr(campos), shs_rest=_ptr(sh_rest) if split else None,
                        param_space=int(param_space))
        bw = GsrBackwardInputs(_ptr(dL_dout_color), _ptr(dLd), _ptr(radii), int(R), _ptr(geomBuffer),
                               _ptr(binningBuffer), _ptr(imageBuffer))
        gr = GsrGrads(_ptr(dL_dmeans2D), _ptr(dL_dcolors), _ptr(dL_dopacity), _ptr(dL_dmeans3D), _ptr(dL_dcov3D),
                      _ptr(dL_dsh), _ptr(dL_dscales), _ptr(dL_drotations), None, None,
                      _ptr(dL_dsh_rest) if split else None)
        with torch.cuda.device(dev):
            check(L.gsr_backward(C.byref(st), C.byref(inp), C.byref(bw), C.byref(gr),
                                 C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    out = (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)
    return out + (dL_dsh_rest,) if split else out


def selftest_wave_sum(x256: torch.Tensor) -> torch.Tensor:
    L = _bind()
    out = torch.zeros(44, dtype=torch.float32, device=x256.device)  # 4 wave sums + 4 x 10 transpose-reduce totals
    with torch.cuda.device(x256.device):
        check(L.gsr_selftest_wave_sum(_ptr(x256), _ptr(out), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out
