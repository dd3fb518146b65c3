"""ctypes binding of libgsr_hip.so -- the C ABI declared in include/gsr.h.

The library is built in-tree (``gsworld_amd/libgsr_hip.so``) by ``gsworld_amd.build`` / ``__graft_entry__.build()``.
There is NO fallback: if the shared object is missing or a symbol cannot be resolved, importing the binding
raises, and every operator that needs it fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgsr_hip.so")

GSR_OK = 0
GSR_E_INVALID = -1
GSR_E_HIP = -2
GSR_E_ALLOC = -3
GSR_E_OVERFLOW = -4
GSR_E_TRUNCATED = -5
GSR_NEAR_PLANE = 0.05  # /root/reference/README.md:33
MAX_FRAMES_PER_LAUNCH = 8  # include/gsr.h GSR_MAX_FRAMES_PER_LAUNCH


class GsrSettings(C.Structure):
    _fields_ = [
        ("image_height", C.c_int32), ("image_width", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("sh_degree", C.c_int32), ("sh_coeffs", C.c_int32),
        ("prefiltered", C.c_int32), ("antialiasing", C.c_int32), ("debug", C.c_int32),
        ("near_plane", C.c_float),
        # A/B and test selectors, 0 = library default (include/gsr.h); they travel with every call
        ("binning_path", C.c_int32), ("render_variant", C.c_int32), ("render_blocks_per_cu", C.c_int32),
        ("depth_sort", C.c_int32), ("render_split", C.c_int32),
        # 1 = inference frame: nothing a backward would read is written, instances are binned per 2 x 1 super-tile
        # (bit-identical image; include/gsr.h)
        ("forward_only", C.c_int32),
    ]


# Python-side defaults of the three selectors above (tests, tools/ab_render.py, bench.py --render-bpc): the shared
# library itself keeps no mutable state, every GsrSettings built by this package copies these in.
TUNING = {"binning_path": 0, "render_variant": 0, "render_blocks_per_cu": 0, "depth_sort": 0, "render_split": 0,
          "forward_only": -1}  # forward_only: -1 = each caller's own choice, 0 / 1 = forced (A/B runs)
# GSWORLD_AMD_TUNING="binning_path=4,depth_sort=1": A/B runs of the tools and bench.py without editing them
for _kv in filter(None, os.environ.get("GSWORLD_AMD_TUNING", "").split(",")):
    _k, _, _v = _kv.partition("=")
    if _k.strip() not in TUNING:
        raise RuntimeError(f"GSWORLD_AMD_TUNING: unknown selector {_k!r} (known: {sorted(TUNING)})")
    TUNING[_k.strip()] = int(_v)


def apply_tuning(st: "GsrSettings", allow_forward_only: bool = True) -> "GsrSettings":
    st.binning_path = int(TUNING["binning_path"])
    st.render_variant = int(TUNING["render_variant"])
    st.render_blocks_per_cu = int(TUNING["render_blocks_per_cu"])
    st.depth_sort = int(TUNING["depth_sort"])
    st.render_split = int(TUNING["render_split"])
    # (allow_forward_only False: a frame whose state a backward reads -- a forced inference frame there would leave a
    #  lean state that gsr_backward carves as a full one)
    if allow_forward_only and int(TUNING["forward_only"]) >= 0:
        st.forward_only = int(TUNING["forward_only"])
    return st


class GsrInputs(C.Structure):
    _fields_ = [
        ("P", C.c_int32),
        ("background", C.c_void_p), ("means3D", C.c_void_p), ("shs", C.c_void_p),
        ("colors_precomp", C.c_void_p), ("opacities", C.c_void_p), ("scales", C.c_void_p),
        ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p), ("viewmatrix", C.c_void_p),
        ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
        ("shs_rest", C.c_void_p),  # optional features_rest (P,M-1,3): `shs` is then features_dc (forward only)
        ("param_space", C.c_int32),  # RAW_* flags: activations evaluated inside preprocess (forward only)
        # optional per-frame rigid transform of labelled Gaussians inside preprocess (forward only)
        ("part_labels", C.c_void_p), ("part_lut", C.c_void_p), ("part_lut_size", C.c_int32),
        ("part_transforms", C.c_void_p), ("part_count", C.c_int32), ("part_rescale", C.c_void_p),
        # optional block bounds for view-frustum culling + original numbering of a permuted model (gsworld_amd/layout.py)
        ("cull_blocks", C.c_void_p), ("orig_index", C.c_void_p),
    ]


RAW_OPACITY, RAW_SCALES, RAW_ROTATIONS = 1, 2, 4  # include/gsr.h GSR_RAW_*
FRAME_KEPT = 8  # include/gsr.h GSR_FRAME_KEPT: out_rgb8 still holds the previous frame of this state (tile reuse)


def model_version(v: int) -> int:
    """include/gsr.h GSR_MODEL_VERSION: bits 8..31 of ``param_space`` -- the caller's promise that the model arrays, labels,
    LUT and block bounds hold what they held in the previous frame on the state that carried the same (nonzero) version."""
    x = (int(v) & 0xFFFFFF) << 8
    return x - (1 << 32) if x >= (1 << 31) else x  # (param_space is an int32)


class GsrOutputs(C.Structure):
    _fields_ = [("out_color", C.c_void_p), ("out_invdepth", C.c_void_p), ("radii", C.c_void_p),
                ("out_rgb8", C.c_void_p),  # optional (H,W,3) uint8 frame written by the compositor
                ("overflow_mirror", C.c_void_p)]  # optional: two host-readable words for the state's overflow count


RESIZE_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)


class GsrBuffers(C.Structure):
    _fields_ = [
        ("geom_resize", RESIZE_FN), ("geom_user", C.c_void_p),
        ("binning_resize", RESIZE_FN), ("binning_user", C.c_void_p),
        ("image_resize", RESIZE_FN), ("image_user", C.c_void_p),
    ]


class GsrFrameStats(C.Structure):
    _fields_ = [("num_visible", C.c_int64), ("num_rendered", C.c_int64), ("overflow", C.c_int32),
                ("overflow_frames", C.c_int32), ("truncated", C.c_int32), ("coop_timeouts", C.c_int32)]


class GsrStateView(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "splat", "cov3D", "clamped", "tiles_touched", "rects", "depth_order", "point_list", "point_tiles",
        "ranges", "final_T", "n_contrib")]


class GsrProfile(C.Structure):
    _fields_ = [("frames", C.c_int32), ("stage_ms", C.c_double * 5)]


PROFILE_STAGES = ("preprocess", "depth_sort", "tile_counts+ranges", "placement", "render")


class GsrError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libgsr_hip error {code}: {message}")
        self.code = code


_lib = None


def lib() -> C.CDLL:
    """Load libgsr_hip.so (once).  Raises ``RuntimeError`` if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C gsworld_amd/csrc`). "
            "There is no CPU fallback for the rasterizer.")
    L = C.CDLL(LIB_PATH)
    L.gsr_last_error.restype = C.c_char_p
    L.gsr_version.restype = C.c_char_p
    # ABI handshake: the structs above mirror include/gsr.h by hand; a library built from another header revision
    # would read garbage selectors and pointers out of them
    if not hasattr(L, "gsr_abi_sizes"):
        raise RuntimeError(f"{LIB_PATH} predates the ABI handshake (gsr_abi_sizes): rebuild it (make -C gsworld_amd/csrc)")
    sizes = (C.c_int32 * 6)()
    L.gsr_abi_sizes(sizes)
    mine = (C.sizeof(GsrSettings), C.sizeof(GsrInputs), C.sizeof(GsrOutputs), C.sizeof(GsrBuffers))
    if tuple(sizes[:4]) != mine:
        raise RuntimeError(f"{LIB_PATH} was built from a different include/gsr.h (struct sizes {tuple(sizes[:4])} there, "
                           f"{mine} in gsworld_amd/_lib.py): rebuild the library")
    L.gsr_geom_bytes.restype = C.c_size_t
    L.gsr_geom_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    L.gsr_binning_bytes.restype = C.c_size_t
    L.gsr_binning_bytes.argtypes = [C.c_int64]
    L.gsr_image_bytes.restype = C.c_size_t
    L.gsr_image_bytes.argtypes = [C.c_int32, C.c_int32]
    L.gsr_forward.restype = C.c_int
    L.gsr_forward.argtypes = [C.POINTER(GsrSettings), C.POINTER(GsrInputs), C.POINTER(GsrOutputs),
                              C.POINTER(GsrBuffers), C.c_int64, C.POINTER(GsrFrameStats), C.c_void_p]
    L.gsr_forward_batch.restype = C.c_int
    L.gsr_forward_batch.argtypes = [C.c_int32, C.POINTER(GsrSettings), C.POINTER(GsrInputs), C.POINTER(GsrOutputs),
                                    C.POINTER(GsrBuffers), C.POINTER(C.c_int64), C.c_void_p]
    L.gsr_frame_stats.restype = C.c_int
    L.gsr_frame_stats.argtypes = [C.c_void_p, C.POINTER(GsrFrameStats), C.c_void_p]
    L.gsr_debug_sort_state.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]
    L.gsr_debug_sort_state.restype = C.c_int
    L.gsr_state_view.restype = C.c_int
    L.gsr_state_view.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.POINTER(GsrStateView)]
    L.gsr_mark_visible.restype = C.c_int
    L.gsr_mark_visible.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    L.gsr_pack_rgb8.restype = C.c_int
    L.gsr_pack_rgb8.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.gsr_plan_query.restype = C.c_int
    L.gsr_plan_query.argtypes = [C.POINTER(GsrSettings), C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_int32)]
    L.gsr_pinned_device_address.restype = C.c_int
    L.gsr_pinned_device_address.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.gsr_stage_step.restype = C.c_int
    L.gsr_stage_step.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.gsr_profile_enable.restype = C.c_int
    L.gsr_profile_enable.argtypes = [C.c_int]
    L.gsr_profile_collect.restype = C.c_int
    L.gsr_profile_collect.argtypes = [C.POINTER(GsrProfile)]
    _lib = L
    return L


def check(code: int) -> None:
    if code != GSR_OK:
        raise GsrError(code, lib().gsr_last_error().decode("utf-8", "replace"))


def plan_query(width: int, height: int, P: int, permuted: bool = False, forward_only: bool = True,
               r_capacity: int = 1, tuned: bool = True) -> dict | None:
    """How ``gsr_forward`` would run a frame of this size on a model of ``P`` Gaussians under the current ``TUNING``
    selectors (``tuned``) -- the library's own ``make_plan`` (csrc/api.hip), asked on the host; ``None``: it would refuse the frame
    (a permuted model on a path that does not take one)."""
    st = GsrSettings(int(height), int(width), 1.0, 1.0, 1.0, 3, 16, 0, 0, 0, float(GSR_NEAR_PLANE))
    st.forward_only = int(bool(forward_only))
    if tuned:  # (False: the library's defaults, whatever A/B selectors this process runs under)
        apply_tuning(st)
    out = (C.c_int32 * 8)()
    if lib().gsr_plan_query(C.byref(st), int(P), int(bool(permuted)), int(r_capacity), out) != GSR_OK:
        return None
    keys = ("mode", "placement", "infer", "super", "lean", "radix_depth", "order_early", "exact")
    return dict(zip(keys, (int(v) for v in out)))


def exported_symbols() -> list[str]:
    """Names declared in include/gsr.h (parsed from the header) -- used by the CPU test-suite."""
    import re

    header = os.path.join(os.path.dirname(_HERE), "include", "gsr.h")
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", text)) - {"gsr_resize_fn"})
