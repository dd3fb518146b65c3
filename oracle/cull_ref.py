"""Numpy restatement of the block view-frustum test of csrc/preprocess.hip ``prep_block_culled`` (GsrInputs.cull_blocks).

TEST INFRASTRUCTURE ONLY (tests/, tools/): the product never imports this module.  Same bound, same margins, float32
arithmetic in numpy's order -- not bit-identical to the kernel's, which does not matter: what the tests check is the
PROPERTY the kernel relies on, against the oracle's preprocess: no Gaussian of a culled block has ``radii > 0``."""
from __future__ import annotations

import numpy as np

F = np.float32


def blocks_culled(blocks, viewmatrix, projmatrix, width, height, tanfovx, tanfovy, near_plane=0.05, scale_modifier=1.0,
                  part_pose=None):
    """``blocks`` (NB,8) float32 -> bool (NB,): True = the kernel would skip the block.  ``part_pose``: optional callable
    ``label -> (17,) float32 pose row or None`` (the kernel's label -> LUT -> table lookup); blocks with a NaN label are
    then never culled."""
    b = np.asarray(blocks, F)
    m = np.asarray(viewmatrix, F).reshape(-1)
    q = np.asarray(projmatrix, F).reshape(-1)
    nb = b.shape[0]
    out = np.zeros(nb, bool)
    gx, gy = (width + 15) // 16, (height + 15) // 16
    fx, fy = F(width) / (F(2) * F(tanfovx)), F(height) / (F(2) * F(tanfovy))
    g = np.array([[m[0], m[1], m[2]], [m[4], m[5], m[6]], [m[8], m[9], m[10]]], F)
    G = g @ g.T - np.eye(3, dtype=F)
    wn2 = F(1) + F(np.sqrt((G.astype(np.float64) ** 2).sum()))
    limx, limy = F(1.3) * F(tanfovx), F(1.3) * F(tanfovy)
    corners = np.array([[(c >> k) & 1 for k in range(3)] for c in range(8)], bool)
    with np.errstate(all="ignore"):
        for i in range(nb):
            lo, hi, rho, lab = b[i, 0:3], b[i, 3:6], b[i, 6], b[i, 7]
            p = np.where(corners, hi[None, :], lo[None, :]).astype(F)  # (8,3)
            if part_pose is not None:
                if not lab == lab:
                    continue
                xf = part_pose(int(lab))
                if xf is not None:
                    xf = np.asarray(xf, F)
                    s = xf[12]
                    p = p * s
                    R = xf[0:9].reshape(3, 3)
                    p = (p @ R.T + xf[9:12]).astype(F)
                    n = F((xf[13:17] ** 2).sum())
                    rho = rho * max(F(1), F(2) * n - F(1)) * F(1.0001)
            vz = p[:, 0] * m[2] + p[:, 1] * m[6] + p[:, 2] * m[10] + m[14]
            hx = p[:, 0] * q[0] + p[:, 1] * q[4] + p[:, 2] * q[8] + q[12]
            hy = p[:, 0] * q[1] + p[:, 1] * q[5] + p[:, 2] * q[9] + q[13]
            hw = p[:, 0] * q[3] + p[:, 1] * q[7] + p[:, 2] * q[11] + q[15] + F(1e-7)
            cx = ((hx / hw + F(1)) * F(width) - F(1)) * F(0.5)
            cy = ((hy / hw + F(1)) * F(height) - F(1)) * F(0.5)
            vals = np.concatenate([vz, hw, cx, cy, [rho]])
            if not np.isfinite(vals.sum()) and np.isnan(vals.sum()):
                continue
            zlo, zhi, wlo = vz.min(), vz.max(), hw.min()
            zm = F(1e-5) * (abs(zlo) + abs(zhi) + F(1))
            if zhi < F(near_plane) - zm:
                out[i] = True
                continue
            if not (zlo > zm and wlo > F(1e-6)):
                continue
            j22 = (max(fx * fx * (F(1) + limx * limx), fy * fy * (F(1) + limy * limy)) + fx * fy * limx * limy) / (
                (zlo - zm) * (zlo - zm))
            sr = rho * abs(F(scale_modifier))
            rb = F(3) * np.sqrt(sr * sr * wn2 * j22 * F(1.001) + F(0.6163)) * F(1.0001) + F(2)
            xend, yend = F(16 * gx) + F(1), F(16 * gy) + F(1)
            out[i] = bool(cx.max() + rb < F(-1) or cx.min() - rb > xend or cy.max() + rb < F(-1) or cy.min() - rb > yend)
    return out
