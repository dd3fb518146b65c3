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


_WORKSPACES: dict = {}  # (device, bytes) -> the last workspace of that size (callers that ask again: no allocation)


def distCUDA2(points: torch.Tensor, workspace: torch.Tensor | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
    """points: (P,3) float32 on a HIP device -> (P,) mean squared distance to the 3 nearest other points.
    ``workspace`` / ``out``: optional caller-owned uint8 workspace of ``gsr_knn_workspace_bytes(P)`` bytes and (P,) float32
    result (a timing loop, a caller that runs the op repeatedly); by default the last workspace of the same size on the
    device is reused and the result is a fresh tensor, as upstream's."""
    if not points.is_cuda:
        raise RuntimeError(f"distCUDA2: points are on {points.device}; the MI355X op has no CPU path")
    if points.ndim != 2 or points.size(1) != 3:
        raise RuntimeError("distCUDA2: points must have dimensions (num_points, 3)")
    L = _bind()
    pts = points.detach().to(torch.float32).contiguous()
    P = pts.size(0)
    if out is None:
        out = torch.zeros((P,), dtype=torch.float32, device=pts.device)
    elif tuple(out.shape) != (P,) or out.dtype != torch.float32 or out.device != pts.device or not out.is_contiguous():
        raise RuntimeError("distCUDA2: out must be a dense (P,) float32 tensor on the points' device")
    if P == 0:
        return out
    nbytes = int(L.gsr_knn_workspace_bytes(P))
    ws = workspace
    if ws is None:
        key = (pts.device, nbytes)
        ws = _WORKSPACES.get(key)
        if ws is None:
            _WORKSPACES.clear()  # (one size at a time: create_from_pcd calls this once per model)
            ws = _WORKSPACES[key] = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
    elif ws.numel() < nbytes or ws.dtype != torch.uint8 or ws.device != pts.device:
        raise RuntimeError(f"distCUDA2: workspace must hold {nbytes} bytes (uint8) on the points' device")
    with torch.cuda.device(pts.device):
        check(L.gsr_knn_dist2(P, C.c_void_p(pts.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()),
                              nbytes, C.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream)))
    return out
