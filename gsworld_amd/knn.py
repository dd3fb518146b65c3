"""``simple_knn._C.distCUDA2`` on MI355X: called once by ``GaussianModel.create_from_pcd`` to
initialise the scales (``scales = log(sqrt(clamp_min(distCUDA2(xyz), 1e-7)))``)."""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import check, lib


def _bind():
    L = lib()
    if not getattr(L, "_knn_bound", False):
        L.gsr_knn_workspace_bytes.restype = C.c_size_t
        L.gsr_knn_workspace_bytes.argtypes = [C.c_int32]
        L.gsr_knn_dist2.restype = C.c_int
        L.gsr_knn_dist2.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L._knn_bound = True
    return L


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """points: (P,3) float32 on a HIP device -> (P,) mean squared distance to the 3 nearest other points."""
    if not points.is_cuda:
        raise RuntimeError(f"distCUDA2: points are on {points.device}; the MI355X op has no CPU path")
    if points.ndim != 2 or points.size(1) != 3:
        raise RuntimeError("distCUDA2: points must have dimensions (num_points, 3)")
    L = _bind()
    pts = points.detach().to(torch.float32).contiguous()
    P = pts.size(0)
    out = torch.zeros((P,), dtype=torch.float32, device=pts.device)
    if P == 0:
        return out
    nbytes = int(L.gsr_knn_workspace_bytes(P))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
    with torch.cuda.device(pts.device):
        check(L.gsr_knn_dist2(P, C.c_void_p(pts.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()),
                              nbytes, C.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream)))
    return out
