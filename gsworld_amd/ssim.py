"""``fused_ssim`` on MI355X: the structural-similarity term of the 3DGS training loss
``(1 - lambda) * L1 + lambda * (1 - ssim)`` (lambda_dssim = 0.2, /root/reference/gsworld/utils/gs_utils.py:96).
Mirrors the python side of rahul-goel/fused-ssim: ``fused_ssim(img1, img2, padding="same", train=True)`` and the
``FusedSSIMMap`` autograd function over ``fusedssim`` / ``fusedssim_backward``."""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import check, lib

allowed_padding = ["same", "valid"]


def _bind():
    L = lib()
    if not getattr(L, "_ssim_bound", False):
        L.gsr_ssim_forward.restype = C.c_int
        L.gsr_ssim_forward.argtypes = [C.c_int32] * 4 + [C.c_float, C.c_float] + [C.c_void_p] * 2 + [C.c_int32] + \
            [C.c_void_p] * 5
        L.gsr_ssim_backward.restype = C.c_int
        L.gsr_ssim_backward.argtypes = [C.c_int32] * 4 + [C.c_float, C.c_float] + [C.c_void_p] * 8
        L.gsr_photometric_loss_scratch_floats.restype = C.c_size_t
        L.gsr_photometric_loss_scratch_floats.argtypes = [C.c_int32] * 3
        L.gsr_photometric_loss.restype = C.c_int
        L.gsr_photometric_loss.argtypes = [C.c_int32] * 3 + [C.c_void_p] * 2 + [C.c_float, C.c_int32] + \
            [C.c_void_p] * 2 + [C.c_int32, C.c_void_p]
        L.gsr_photometric_loss_backward.restype = C.c_int
        L.gsr_photometric_loss_backward.argtypes = [C.c_int32] * 3 + [C.c_void_p] * 2 + [C.c_float, C.c_int32] + \
            [C.c_void_p] * 4
        L._ssim_bound = True
    return L


def _p(t):
    return C.c_void_p(t.data_ptr())


def _check_imgs(img1, img2):
    if not (img1.is_cuda and img2.is_cuda):
        raise RuntimeError("fused_ssim: images must live on a HIP device (no CPU path)")
    if img1.ndim != 4 or img1.shape != img2.shape:
        raise RuntimeError("fused_ssim: images must both be (B, C, H, W)")


def fusedssim(C1, C2, img1, img2, train):
    """-> (ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12), all (B,C,H,W)."""
    _check_imgs(img1, img2)
    L = _bind()
    img1 = img1.to(torch.float32).contiguous()
    img2 = img2.to(torch.float32).contiguous()
    B, CH, H, W = img1.shape
    ssim_map = torch.empty_like(img1)
    if train:
        dm_dmu1, dm_dsigma1_sq, dm_dsigma12 = (torch.empty_like(img1) for _ in range(3))
    else:
        dm_dmu1 = dm_dsigma1_sq = dm_dsigma12 = torch.empty(0, device=img1.device)
    with torch.cuda.device(img1.device):
        check(L.gsr_ssim_forward(B, CH, H, W, float(C1), float(C2), _p(img1), _p(img2), int(bool(train)), _p(ssim_map),
                                 _p(dm_dmu1) if train else None, _p(dm_dsigma1_sq) if train else None,
                                 _p(dm_dsigma12) if train else None,
                                 C.c_void_p(torch.cuda.current_stream(img1.device).cuda_stream)))
    return ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12


def fusedssim_backward(C1, C2, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12):
    _check_imgs(img1, img2)
    L = _bind()
    img1 = img1.to(torch.float32).contiguous()
    img2 = img2.to(torch.float32).contiguous()
    dL_dmap = dL_dmap.to(torch.float32).contiguous()
    B, CH, H, W = img1.shape
    dL_dimg1 = torch.empty_like(img1)
    with torch.cuda.device(img1.device):
        check(L.gsr_ssim_backward(B, CH, H, W, float(C1), float(C2), _p(img1), _p(img2), _p(dL_dmap), _p(dm_dmu1),
                                  _p(dm_dsigma1_sq), _p(dm_dsigma12), _p(dL_dimg1),
                                  C.c_void_p(torch.cuda.current_stream(img1.device).cuda_stream)))
    return dL_dimg1


class FusedSSIMMap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, C1, C2, img1, img2, padding="same", train=True):
        ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12 = fusedssim(C1, C2, img1, img2, train)
        if padding == "valid":
            ssim_map = ssim_map[:, :, 5:-5, 5:-5]
        ctx.save_for_backward(img1.detach(), img2, dm_dmu1, dm_dsigma1_sq, dm_dsigma12)
        ctx.C1, ctx.C2, ctx.padding = C1, C2, padding
        return ssim_map

    @staticmethod
    def backward(ctx, opt_grad):
        img1, img2, dm_dmu1, dm_dsigma1_sq, dm_dsigma12 = ctx.saved_tensors
        dL_dmap = opt_grad
        if ctx.padding == "valid":
            dL_dmap = torch.zeros_like(img1)
            dL_dmap[:, :, 5:-5, 5:-5] = opt_grad
        grad = fusedssim_backward(ctx.C1, ctx.C2, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12)
        return None, None, grad, None, None, None


def fused_ssim(img1, img2, padding="same", train=True):
    C1 = 0.01 ** 2
    C2 = 0.03 ** 2
    assert padding in allowed_padding
    ssim_map = FusedSSIMMap.apply(C1, C2, img1, img2, padding, train)
    return ssim_map.mean()


class _PhotometricLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, target, lambda_dssim, clamp01):
        L = _bind()
        if not (img.is_cuda and target.is_cuda):
            raise RuntimeError("photometric_loss: images must live on a HIP device (no CPU path)")
        if img.shape != target.shape or img.ndim < 2:
            raise RuntimeError("photometric_loss: img and target must have the same (..., H, W) shape")
        x = img.detach().to(torch.float32).contiguous()
        t = target.detach().to(torch.float32).contiguous()
        H, W = x.shape[-2:]
        planes = x.numel() // (H * W) if H * W else 0
        need_grad = ctx.needs_input_grad[0]
        scratch = torch.empty(int(L.gsr_photometric_loss_scratch_floats(planes, H, W)), dtype=torch.float32, device=x.device)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            check(L.gsr_photometric_loss(planes, H, W, _p(x), _p(t), float(lambda_dssim), int(bool(clamp01)), _p(scratch),
                                         _p(loss), int(need_grad), C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        if need_grad:
            ctx.save_for_backward(x, t, scratch)
            ctx.args = (planes, H, W, float(lambda_dssim), int(bool(clamp01)), img.dtype)
        return loss

    @staticmethod
    def backward(ctx, g):
        x, t, scratch = ctx.saved_tensors
        planes, H, W, lam, clamp01, in_dtype = ctx.args
        g = g.detach().to(device=x.device, dtype=torch.float32).contiguous()
        grad = torch.empty_like(x)
        with torch.cuda.device(x.device):  # (the incoming gradient is read by the kernel: no multiply pass afterwards)
            check(_bind().gsr_photometric_loss_backward(planes, H, W, _p(x), _p(t), lam, clamp01, _p(scratch), _p(g),
                                                        _p(grad), C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        return grad.to(in_dtype), None, None, None


def photometric_loss(img, target, lambda_dssim: float = 0.2, clamp01: bool = False):
    """``(1 - lambda) * l1_loss(x, target) + lambda * (1 - fused_ssim(x[None], target[None]))`` with
    ``x = img.clamp(0, 1)`` when ``clamp01`` -- the loss of the 3DGS training step (gs_utils.py:96 ``lambda_dssim``;
    upstream train.py) as ONE autograd node over three kernels (``gsr_photometric_loss`` + ``_backward``) instead of ~25
    elementwise / reduction launches: the value is summed in a fixed order; the backward pass writes the gradient
    w.r.t. ``img`` (through the clamp), already scaled by the incoming gradient.  An extension: ``fused_ssim`` stays the drop-in."""
    return _PhotometricLoss.apply(img, target, lambda_dssim, clamp01)
