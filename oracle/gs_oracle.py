"""ctypes front-end of the CPU oracle (oracle/gs_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``.  The product package ``gsworld_amd`` must never import this module.

Parity status: UNPINNED against the CUDA reference (see the header of gs_oracle.c and DESIGN.md); pinned
against analytic known-answer tests (tests/test_oracle_kat.py) and an independent numpy/torch
restatement (oracle/torch_cpu_render.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GS_ORACLE_LIB: an exposure build of the same source (tools/fma_exposure.py only; never a checker)
_LIB_PATH = os.environ.get("GS_ORACLE_LIB") or os.path.join(_HERE, "libgs_oracle.so")
_lib = None


class GsoSettings(C.Structure):
    _fields_ = [
        ("image_height", C.c_int32),
        ("image_width", C.c_int32),
        ("tanfovx", C.c_float),
        ("tanfovy", C.c_float),
        ("scale_modifier", C.c_float),
        ("sh_degree", C.c_int32),
        ("sh_coeffs", C.c_int32),
        ("prefiltered", C.c_int32),
        ("antialiasing", C.c_int32),
        ("near_plane", C.c_float),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gs_oracle.c")
    if os.environ.get("GS_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.gso_higher_msb.restype = C.c_uint32
        _lib.gso_higher_msb.argtypes = [C.c_uint32]
        _lib.gso_inclusive_sum.restype = C.c_int64
        _lib.gso_max_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def higher_msb(n: int) -> int:
    return int(lib().gso_higher_msb(n))


def set_threads(n: int) -> None:
    lib().gso_set_threads(C.c_int(n))


def max_threads() -> int:
    return int(lib().gso_max_threads())


@dataclass
class Settings:
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    scale_modifier: float = 1.0
    sh_degree: int = 3
    sh_coeffs: int = 16
    prefiltered: bool = False
    antialiasing: bool = False
    near_plane: float = 0.05

    def c(self) -> GsoSettings:
        return GsoSettings(
            self.image_height, self.image_width, self.tanfovx, self.tanfovy, self.scale_modifier,
            self.sh_degree, self.sh_coeffs, int(self.prefiltered), int(self.antialiasing), self.near_plane,
        )

    @property
    def grid(self):
        return ((self.image_width + 15) // 16, (self.image_height + 15) // 16)


def preprocess(st: Settings, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
               viewmatrix, projmatrix, campos):
    means3D = _f32(means3D)
    P = means3D.shape[0]
    shs, colors_precomp, opacities = _f32(shs), _f32(colors_precomp), _f32(opacities)
    scales, rotations, cov3D_precomp = _f32(scales), _f32(rotations), _f32(cov3D_precomp)
    viewmatrix, projmatrix, campos = _f32(viewmatrix), _f32(projmatrix), _f32(campos)
    out = dict(
        depths=np.zeros(P, np.float32), radii=np.zeros(P, np.int32), means2D=np.zeros((P, 2), np.float32),
        cov3D=np.zeros((P, 6), np.float32), conic_opacity=np.zeros((P, 4), np.float32),
        rgb=np.zeros((P, 3), np.float32), clamped=np.zeros((P, 3), np.uint8),
        tiles_touched=np.zeros(P, np.uint32), rects=np.zeros((P, 4), np.int32),
    )
    cs = st.c()
    lib().gso_preprocess(
        C.byref(cs), C.c_int(P), _p(means3D), _p(shs), _p(colors_precomp), _p(opacities), _p(scales),
        _p(rotations), _p(cov3D_precomp), _p(viewmatrix), _p(projmatrix), _p(campos),
        _p(out["depths"]), _p(out["radii"]), _p(out["means2D"]), _p(out["cov3D"]), _p(out["conic_opacity"]),
        _p(out["rgb"]), _p(out["clamped"]), _p(out["tiles_touched"]), _p(out["rects"]),
    )
    return out


def mark_visible(means3D, viewmatrix, near_plane=0.05):
    means3D, viewmatrix = _f32(means3D), _f32(viewmatrix)
    P = means3D.shape[0]
    present = np.zeros(P, np.uint8)
    lib().gso_mark_visible(C.c_int(P), _p(means3D), _p(viewmatrix), C.c_float(near_plane), _p(present))
    return present.astype(bool)


def bin_tiles(st: Settings, geom):
    P = geom["depths"].shape[0]
    offsets = np.zeros(P, np.uint32)
    R = int(lib().gso_inclusive_sum(C.c_int(P), _p(geom["tiles_touched"]), _p(offsets)))
    gx, gy = st.grid
    keys = np.zeros(max(R, 1), np.uint64)
    values = np.zeros(max(R, 1), np.uint32)
    keys_unsorted = np.zeros(max(R, 1), np.uint64)
    ranges = np.zeros((gx * gy, 2), np.uint32)
    cs = st.c()
    lib().gso_bin(C.byref(cs), C.c_int(P), _p(geom["depths"]), _p(geom["radii"]), _p(geom["rects"]),
                  _p(offsets), C.c_int64(R), _p(keys), _p(values), _p(keys_unsorted), _p(ranges))
    return dict(offsets=offsets, num_rendered=R, keys=keys[:R], point_list=values[:R],
                keys_unsorted=keys_unsorted[:R], ranges=ranges)


def render(st: Settings, geom, binning, bg, border_eps: float = 0.0, border_eps_T: float = 0.0):
    H, W = st.image_height, st.image_width
    bg = _f32(bg)
    out_color = np.zeros((3, H, W), np.float32)
    out_invdepth = np.zeros((1, H, W), np.float32)
    final_T = np.zeros((H, W), np.float32)
    n_contrib = np.zeros((H, W), np.uint32)
    borderline = np.zeros((H, W), np.uint32) if border_eps > 0 else None
    cs = st.c()
    pl = binning["point_list"] if binning["num_rendered"] > 0 else np.zeros(1, np.uint32)
    pl = np.ascontiguousarray(pl, np.uint32)
    lib().gso_render(C.byref(cs), _p(binning["ranges"]), _p(pl), _p(geom["means2D"]), _p(geom["conic_opacity"]),
                     _p(geom["rgb"]), _p(geom["depths"]), _p(bg), _p(out_color), _p(out_invdepth),
                     _p(final_T), _p(n_contrib), C.c_float(border_eps), C.c_float(border_eps_T), _p(borderline))
    return dict(color=out_color, invdepth=out_invdepth, final_T=final_T, n_contrib=n_contrib,
                borderline=borderline)


def forward(st: Settings, bg, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
            viewmatrix, projmatrix, campos, border_eps: float = 0.0, border_eps_T: float = 0.0):
    """Whole forward pass; returns every intermediate (geometry, binning, image)."""
    geom = preprocess(st, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                      viewmatrix, projmatrix, campos)
    binning = bin_tiles(st, geom)
    img = render(st, geom, binning, bg, border_eps, border_eps_T)
    return dict(geom=geom, binning=binning, **img)


def render_bruteforce(st: Settings, geom, bg):
    H, W = st.image_height, st.image_width
    P = geom["depths"].shape[0]
    bg = _f32(bg)
    out_color = np.zeros((3, H, W), np.float64)
    out_invdepth = np.zeros((1, H, W), np.float64)
    cs = st.c()
    lib().gso_render_bruteforce(C.byref(cs), C.c_int(P), _p(geom["radii"]), _p(geom["rects"]),
                                _p(geom["means2D"]), _p(geom["conic_opacity"]), _p(geom["rgb"]),
                                _p(geom["depths"]), _p(bg), _p(out_color), _p(out_invdepth))
    return out_color, out_invdepth


def backward(st: Settings, fwd, inp, bg, dL_dcolor, dL_dinvdepth=None, colors_precomp=None, cov3D_precomp=None):
    """Backward pass on top of a ``forward`` result.  ``inp`` is the dict of forward inputs (numpy).
    Returns the 8 gradients of upstream's rasterize_gaussians_backward, same order and shapes:
    (dL_dmeans2D (P,3), dL_dcolors (P,3), dL_dopacity (P,1), dL_dmeans3D (P,3), dL_dcov3D (P,6),
     dL_dsh (P,M,3), dL_dscales (P,3), dL_drotations (P,4))."""
    geom, binning = fwd["geom"], fwd["binning"]
    P = geom["depths"].shape[0]
    H, W = st.image_height, st.image_width
    bg = _f32(bg)
    dL_dcolor = _f32(dL_dcolor).reshape(3, H, W)
    dLd = None if dL_dinvdepth is None else _f32(dL_dinvdepth).reshape(H, W)
    d_mean2D = np.zeros((P, 2), np.float64)
    d_conic = np.zeros((P, 3), np.float64)
    d_opac = np.zeros(P, np.float64)
    d_colors = np.zeros((P, 3), np.float64)
    d_invd = np.zeros(P, np.float64)
    cs = st.c()
    pl = np.ascontiguousarray(binning["point_list"] if binning["num_rendered"] > 0 else np.zeros(1), np.uint32)
    lib().gso_render_backward(
        C.byref(cs), C.c_int(P), _p(binning["ranges"]), _p(pl), _p(geom["means2D"]), _p(geom["conic_opacity"]),
        _p(geom["rgb"]), _p(geom["depths"]), _p(bg), _p(fwd["final_T"]), _p(fwd["n_contrib"]), _p(dL_dcolor),
        _p(dLd), _p(d_mean2D), _p(d_conic), _p(d_opac), _p(d_colors), _p(d_invd))
    M = st.sh_coeffs
    f_mean2D, f_conic = d_mean2D.astype(np.float32), d_conic.astype(np.float32)
    f_opac, f_colors, f_invd = d_opac.astype(np.float32), d_colors.astype(np.float32), d_invd.astype(np.float32)
    d_means3D = np.zeros((P, 3), np.float32)
    d_cov3D = np.zeros((P, 6), np.float32)
    d_sh = np.zeros((P, max(M, 1), 3), np.float32)
    d_scales = np.zeros((P, 3), np.float32)
    d_rots = np.zeros((P, 4), np.float32)
    cov = _f32(cov3D_precomp) if cov3D_precomp is not None else geom["cov3D"]
    shs = None if colors_precomp is not None else _f32(inp["shs"])
    scales = None if cov3D_precomp is not None else _f32(inp["scales"])
    rots = None if cov3D_precomp is not None else _f32(inp["rotations"])
    lib().gso_preprocess_backward(
        C.byref(cs), C.c_int(P), _p(_f32(inp["means3D"])), _p(geom["radii"]), _p(shs), _p(geom["clamped"]),
        _p(_f32(inp["opacities"])), _p(scales), _p(rots), _p(cov), C.c_int(cov3D_precomp is not None),
        C.c_int(colors_precomp is not None), _p(_f32(inp["viewmatrix"])), _p(_f32(inp["projmatrix"])),
        _p(_f32(inp["campos"])), _p(f_mean2D), _p(f_conic), _p(f_opac), _p(f_colors),
        _p(f_invd) if dLd is not None else None, _p(d_means3D), _p(d_cov3D),
        _p(d_sh) if colors_precomp is None else None, _p(d_scales) if cov3D_precomp is None else None,
        _p(d_rots) if cov3D_precomp is None else None)
    d_means2D3 = np.zeros((P, 3), np.float32)
    d_means2D3[:, :2] = f_mean2D
    return dict(dL_dmeans2D=d_means2D3, dL_dcolors=f_colors, dL_dopacity=f_opac.reshape(P, 1),
                dL_dmeans3D=d_means3D, dL_dcov3D=d_cov3D, dL_dsh=d_sh, dL_dscales=d_scales, dL_drotations=d_rots,
                dL_dconic=f_conic, dL_dinvdepths=f_invd)


def knn_dist2(points):
    points = _f32(points)
    P = points.shape[0]
    out = np.zeros(P, np.float32)
    lib().gso_knn_dist2(C.c_int(P), _p(points), _p(out))
    return out


def activate_params(opacities=None, scales=None, rotations=None, flags=7):
    """Canonical float32 activations of raw 3DGS parameters (gso_activate_params): flags 1 = sigmoid(opacity
    logits), 2 = exp(log scales), 4 = rotation / max(|rotation|, 1e-12).  Returns (opacities, scales, rotations)."""
    L = lib()
    L.gso_activate_params.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 6
    L.gso_activate_params.restype = None
    P = next(a.shape[0] for a in (opacities, scales, rotations) if a is not None)
    op = _f32(opacities).reshape(-1) if opacities is not None else None
    sc = _f32(scales) if scales is not None else None
    ro = _f32(rotations) if rotations is not None else None
    op_o = np.zeros_like(op) if op is not None else None
    sc_o = np.zeros_like(sc) if sc is not None else None
    ro_o = np.zeros_like(ro) if ro is not None else None
    L.gso_activate_params(C.c_int(P), C.c_int(flags), _p(op), _p(sc), _p(ro), _p(op_o), _p(sc_o), _p(ro_o))
    return op_o, sc_o, ro_o


def expf(x):
    """gso_expf elementwise (the canonical exp of the raw-parameter path)."""
    L = lib()
    L.gso_expf.argtypes = [C.c_float]
    L.gso_expf.restype = C.c_float
    return np.array([L.gso_expf(float(v)) for v in np.asarray(x, np.float32).reshape(-1)], np.float32)
