"""Vectorised PyTorch-CPU restatement of the forward rasterizer (BASELINE.json configs[0]: "PyTorch-CPU
reference render (no GPU, numerics gate)").

TEST INFRASTRUCTURE ONLY (see oracle/gs_oracle.c header; parity vs the CUDA reference is UNPINNED).  This is a
second, independently written restatement of the same published algorithm: plain torch
tensor ops, no explicit FMA order, torch.sort for the keys.  Its job is to cross-check the C oracle -- the two
must agree to <= 1e-5 on the image and exactly on every integer -- so that a slip in one restatement does not
silently become "the reference".
"""
from __future__ import annotations

import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


def _sh_to_rgb(deg, sh, dirs):
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
                   + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6] + SH_C2[3] * xz * sh[:, 7]
                   + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    res = res + 0.5
    return torch.clamp_min(res, 0.0), res < 0


def render(means3D, shs, opacities, scales, rotations, viewmatrix, projmatrix, campos, bg, H, W, tanfovx, tanfovy,
           sh_degree=3, scale_modifier=1.0, near_plane=0.05, antialiasing=False):
    """All arguments are CPU tensors of one floating dtype (float32 for the numerics gate, float64 for the
    autograd check of the backward oracle) laid out as the rasterizer takes them (viewmatrix/projmatrix are the
    transposed 4x4s).  Differentiable w.r.t. means3D, shs, opacities, scales, rotations (piecewise: the cull /
    sort / threshold decisions are constants).  Returns dict(color (3,H,W), invdepth (1,H,W), radii, ...)."""
    P = means3D.shape[0]
    dt = means3D.dtype
    _default = torch.get_default_dtype()
    torch.set_default_dtype(dt)
    try:
        return _render(means3D, shs, opacities, scales, rotations, viewmatrix, projmatrix, campos, bg, H, W, tanfovx,
                       tanfovy, sh_degree, scale_modifier, near_plane, antialiasing)
    finally:
        torch.set_default_dtype(_default)


def _render(means3D, shs, opacities, scales, rotations, viewmatrix, projmatrix, campos, bg, H, W, tanfovx, tanfovy,
            sh_degree, scale_modifier, near_plane, antialiasing):
    P = means3D.shape[0]
    dt = means3D.dtype
    V = viewmatrix.reshape(4, 4)  # row-vector convention: p_view = [p,1] @ V
    Pm = projmatrix.reshape(4, 4)
    ones = torch.ones(P, 1)
    p_view = torch.cat((means3D, ones), 1) @ V
    p_hom = torch.cat((means3D, ones), 1) @ Pm
    p_w = 1.0 / (p_hom[:, 3] + 0.0000001)
    ndc = p_hom[:, :2] * p_w[:, None]
    in_front = p_view[:, 2] > near_plane

    # 3D covariance
    r, x, y, z = rotations.unbind(1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(P, 3, 3)
    S = scale_modifier * scales
    RS = R * S[:, None, :]
    Sigma = RS @ RS.transpose(1, 2)

    # EWA 2D covariance
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    tz = p_view[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tx = torch.clamp(p_view[:, 0] / tz, -limx, limx) * tz
    ty = torch.clamp(p_view[:, 1] / tz, -limy, limy) * tz
    J = torch.zeros(P, 2, 3)
    J[:, 0, 0] = fx / tz
    J[:, 0, 2] = -(fx * tx) / (tz * tz)
    J[:, 1, 1] = fy / tz
    J[:, 1, 2] = -(fy * ty) / (tz * tz)
    Wm = V[:3, :3].T  # world -> camera rotation
    A = J @ Wm
    cov = A @ Sigma @ A.transpose(1, 2)
    cxx, cxy, cyy = cov[:, 0, 0], cov[:, 0, 1], cov[:, 1, 1]
    det0 = cxx * cyy - cxy * cxy
    cxx = cxx + 0.3
    cyy = cyy + 0.3
    det = cxx * cyy - cxy * cxy
    h_scale = torch.sqrt(torch.clamp_min(det0 / det, 0.000025)) if antialiasing else torch.ones(P)
    det_ok = det != 0
    det_inv = 1.0 / det
    conic = torch.stack((cyy * det_inv, -cxy * det_inv, cxx * det_inv), 1)
    mid = 0.5 * (cxx + cyy)
    root = torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(mid + root, mid - root)))
    pix = torch.stack((((ndc[:, 0].double() + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1].double() + 1.0) * H - 1.0) * 0.5),
                      1).to(dt)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ir = radius.detach().nan_to_num(0.0).to(torch.int32).to(dt)

    def tile(v, g):
        return torch.clamp(torch.trunc(v.detach() / 16.0).nan_to_num(0.0).to(torch.int64), 0, g)

    rminx, rminy = tile(pix[:, 0] - ir, gx), tile(pix[:, 1] - ir, gy)
    rmaxx, rmaxy = tile(pix[:, 0] + ir + 15.0, gx), tile(pix[:, 1] + ir + 15.0, gy)
    touched = (rmaxx - rminx) * (rmaxy - rminy)
    visible = in_front & det_ok & (touched > 0)
    radii = torch.where(visible, ir.to(torch.int32), torch.zeros(P, dtype=torch.int32))

    dirs = means3D - campos[None, :]
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    rgb, _ = _sh_to_rgb(sh_degree, shs, dirs)
    opac = opacities.reshape(-1) * h_scale
    depth = p_view[:, 2]

    # binning: key = tile << 32 | depth bits, stable sort == sort by (tile, depth bits, index)
    vis_idx = torch.nonzero(visible).squeeze(1)
    counts = touched[vis_idx]
    R_total = int(counts.sum())
    g_rep = torch.repeat_interleave(vis_idx, counts)
    start = torch.cumsum(counts, 0) - counts
    local = torch.arange(R_total) - torch.repeat_interleave(start, counts)
    wdt = (rmaxx - rminx)[g_rep]
    ty_ = rminy[g_rep] + local // wdt
    tx_ = rminx[g_rep] + local % wdt
    tile_id = ty_ * gx + tx_
    dbits = depth.detach().float().view(torch.int32).to(torch.int64)[g_rep]
    keys = (tile_id << 32) | dbits
    order = torch.sort(keys, stable=True).indices
    keys_s, g_s = keys[order], g_rep[order]
    tiles_s = keys_s >> 32
    bounds = torch.searchsorted(tiles_s, torch.arange(gx * gy + 1))

    color_tiles, invd_tiles = {}, {}
    final_T = torch.ones(H, W)
    n_contrib = torch.zeros(H, W, dtype=torch.int32)
    ys, xs = torch.meshgrid(torch.arange(16), torch.arange(16), indexing="ij")
    for t in range(gx * gy):
        a, b = int(bounds[t]), int(bounds[t + 1])
        x0, y0 = (t % gx) * 16, (t // gx) * 16
        pxs = (x0 + xs).reshape(-1).to(dt)
        pys = (y0 + ys).reshape(-1).to(dt)
        n = b - a
        T = torch.ones(256)
        C = torch.zeros(256, 3)
        D = torch.zeros(256)
        last = torch.zeros(256, dtype=torch.int32)
        if n > 0:
            g = g_s[a:b]
            dx = pix[g, 0][:, None] - pxs[None, :]
            dy = pix[g, 1][:, None] - pys[None, :]
            co = conic[g]
            power = -0.5 * (co[:, 0:1] * dx * dx + co[:, 2:3] * dy * dy) - co[:, 1:2] * dx * dy
            alpha = torch.clamp_max(opac[g][:, None] * torch.exp(power), 0.99)
            valid = (power <= 0) & (alpha >= 1.0 / 255.0)
            a_eff = torch.where(valid, alpha, torch.zeros(()))
            Tcum = torch.cumprod(1.0 - a_eff, 0)
            T_before = torch.cat((torch.ones(1, 256), Tcum[:-1]), 0)
            stop = valid & ((T_before * (1.0 - a_eff)).detach() < 0.0001)
            alive = torch.cumsum(stop.to(torch.int32), 0) == 0  # strictly before the terminating instance
            contrib = valid & alive
            w = torch.where(contrib, a_eff * T_before, torch.zeros(()))
            C = torch.einsum("np,nc->pc", w, rgb[g])
            D = (w * (1.0 / depth[g])[:, None]).sum(0)
            n_alive = alive.to(torch.int32).sum(0)
            T = torch.where(n_alive > 0, torch.gather(Tcum, 0, (n_alive - 1).clamp_min(0).long()[None, :])[0],
                            torch.ones(256))
            # final T excludes the terminating instance: it is T after the last ALIVE instance
            idx = torch.arange(1, n + 1, dtype=torch.int32)[:, None]
            last = torch.where(contrib, idx, torch.zeros((), dtype=torch.int32)).max(0).values
        hh, ww = min(16, H - y0), min(16, W - x0)
        sel = (ys < hh) & (xs < ww)
        col = (C + T[:, None] * bg[None, :]).reshape(16, 16, 3)
        color_tiles[t] = col.permute(2, 0, 1)
        invd_tiles[t] = D.reshape(16, 16)
        final_T[y0:y0 + hh, x0:x0 + ww] = T.detach().reshape(16, 16)[:hh, :ww]
        n_contrib[y0:y0 + hh, x0:x0 + ww] = last.reshape(16, 16)[:hh, :ww]
        del sel
    # assemble the image from the tiles without in-place writes (keeps autograd intact)
    rows_c, rows_d = [], []
    for ty in range(gy):
        rows_c.append(torch.cat([color_tiles[ty * gx + tx] for tx in range(gx)], dim=2))
        rows_d.append(torch.cat([invd_tiles[ty * gx + tx] for tx in range(gx)], dim=1))
    color = torch.cat(rows_c, dim=1)[:, :H, :W]
    invdepth = torch.cat(rows_d, dim=0)[None, :H, :W]
    return dict(color=color, invdepth=invdepth, radii=radii, final_T=final_T, n_contrib=n_contrib,
                num_rendered=R_total, point_list=g_s, keys=keys_s, means2D=pix, depths=depth, rgb=rgb,
                conic=conic, opacity=opac, tiles_touched=torch.where(visible, touched, torch.zeros(())).long())


def fov2tan(fov_rad: float) -> float:
    return math.tan(fov_rad * 0.5)
