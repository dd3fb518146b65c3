"""Shared helpers of the parity tests: run the oracle (CPU) and the HIP path (GPU) on the same inputs and compare
every stage.  The oracle is the checker only; nothing here is imported by the product package."""
from __future__ import annotations

import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import gs_oracle as go  # noqa: E402

RGB_TOL = 1e-4  # north_star: RGB / inverse depth max-abs <= 1e-4
BORDER_EPS = 5e-6  # relative band around alpha = 1/255 that an exp() ulp difference can flip
BORDER_EPS_T = 1e-4  # relative band around T = 1e-4 (T accumulates the exp() differences of all earlier splats)


def np_inputs(raw, cam):
    means, shs, op, sc, rot = raw.activated()
    return dict(
        means3D=means.numpy(), shs=shs.numpy(), opacities=op.numpy().reshape(-1), scales=sc.numpy(),
        rotations=rot.numpy(), viewmatrix=cam.world_view_transform.numpy().reshape(-1),
        projmatrix=cam.full_proj_transform.numpy().reshape(-1), campos=cam.camera_center.numpy())


def oracle_settings(cam, sh_degree=3, sh_coeffs=16, antialiasing=False, scale_modifier=1.0, near_plane=0.05):
    return go.Settings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, scale_modifier, sh_degree,
                       sh_coeffs, False, antialiasing, near_plane)


def oracle_forward(inp, st, bg, border_eps=BORDER_EPS, colors_precomp=None, cov3D_precomp=None,
                   border_eps_T=BORDER_EPS_T):
    return go.forward(st, bg, inp["means3D"], None if colors_precomp is not None else inp["shs"], colors_precomp,
                      inp["opacities"], None if cov3D_precomp is not None else inp["scales"],
                      None if cov3D_precomp is not None else inp["rotations"], cov3D_precomp, inp["viewmatrix"],
                      inp["projmatrix"], inp["campos"], border_eps=border_eps, border_eps_T=border_eps_T)


def gpu_forward(inp, st, bg, device="cuda", colors_precomp=None, cov3D_precomp=None, debug=False, param_space=0):
    """Runs gsworld_amd._C.rasterize_gaussians on ``device`` and returns outputs + typed state views (CPU numpy)."""
    from gsworld_amd import _C, debug as dbg

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
    empty = torch.empty(0, device=device)
    P = inp["means3D"].shape[0]
    sh = empty if colors_precomp is not None else t(inp["shs"])
    colors = t(colors_precomp) if colors_precomp is not None else empty
    scales = empty if cov3D_precomp is not None else t(inp["scales"])
    rots = empty if cov3D_precomp is not None else t(inp["rotations"])
    cov = t(cov3D_precomp) if cov3D_precomp is not None else empty
    old_near = _C.NEAR_PLANE
    _C.NEAR_PLANE = st.near_plane
    try:
        R, color, radii, geomB, binB, imgB, invd = _C.rasterize_gaussians(
            t(np.asarray(bg, np.float32)), t(inp["means3D"]), colors, t(inp["opacities"]).reshape(-1, 1), scales, rots,
            st.scale_modifier, cov, t(inp["viewmatrix"]).reshape(4, 4), t(inp["projmatrix"]).reshape(4, 4),
            st.tanfovx, st.tanfovy, st.image_height, st.image_width, sh, st.sh_degree, t(inp["campos"]),
            st.prefiltered, st.antialiasing, debug, param_space=param_space)
    finally:
        _C.NEAR_PLANE = old_near
    torch.cuda.synchronize()
    out = dict(num_rendered=R, color=color.cpu().numpy(), radii=radii.cpu().numpy(), invdepth=invd.cpu().numpy())
    if P > 0:
        V = int((radii > 0).sum().item())
        views = dbg.state_view(P, st.image_width, st.image_height, R, V, geomB, binB, imgB)
        out["num_visible"] = V
        out["views"] = {k: v.cpu().numpy() for k, v in views.items()}
    return out


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def compare_forward(o, g, st, rgb_tol=RGB_TOL, check_image=True, all_pixel_tol=0.02):
    """Asserts stage-by-stage parity; returns a report dict (counts / max errors).  ``all_pixel_tol``: bound on EVERY
    pixel, borderline ones included (a borderline pixel may flip one alpha >= 1/255 or T >= 1e-4 decision under a
    1-ulp exp() difference: 0.02 covers the worst case and is kept for fuzz inputs; the BASELINE configurations are held
    to north_star's 1e-4 on all pixels)."""
    rep = {}
    geom, binning = o["geom"], o["binning"]
    vis = geom["radii"] > 0
    rep["P"] = int(vis.shape[0])
    rep["V"] = int(vis.sum())
    rep["R"] = int(binning["num_rendered"])
    # ---- preprocess: integers and every float bit-exact ----------------------------------------------------
    np.testing.assert_array_equal(g["radii"], geom["radii"], err_msg="radii")
    v = g["views"]
    np.testing.assert_array_equal(v["tiles_touched"].astype(np.uint32), geom["tiles_touched"], err_msg="tiles_touched")
    np.testing.assert_array_equal(v["rects"][vis].astype(np.int32), geom["rects"][vis], err_msg="rects")
    for name, ga, oa in (("means2D", v["means2D"], geom["means2D"]), ("depths", v["depths"], geom["depths"]),
                         ("conic_opacity", v["conic_opacity"], geom["conic_opacity"]), ("rgb", v["rgb"], geom["rgb"]),
                         ("cov3D", v["cov3D"], geom["cov3D"])):
        gb, ob = _bits(ga[vis]), _bits(oa[vis])
        bad = int((gb != ob).sum())
        rep[f"{name}_bit_mismatch"] = bad
        assert bad == 0, f"{name}: {bad} float(s) differ bitwise from the oracle " \
                         f"(max abs {np.abs(ga[vis] - oa[vis]).max()})"
    np.testing.assert_array_equal(v["clamped"][vis], geom["clamped"][vis], err_msg="clamped")
    # ---- binning: bit-exact indices -------------------------------------------------------------------------
    assert g["num_visible"] == rep["V"], "V"
    assert g["num_rendered"] == rep["R"], f"num_rendered {g['num_rendered']} != {rep['R']}"
    idx = np.nonzero(vis)[0].astype(np.uint32)
    order = idx[np.lexsort((idx, _bits(geom["depths"][vis])))]
    if "depth_order" in v:  # binning modes 0 / 1 only; the default path never builds a global depth order
        np.testing.assert_array_equal(v["depth_order"].astype(np.uint32), order, err_msg="depth order")
    if rep["R"] > 0:
        np.testing.assert_array_equal(v["point_list"].astype(np.uint32), binning["point_list"], err_msg="point_list")
        np.testing.assert_array_equal(v["point_tiles"].astype(np.uint64), binning["keys"] >> np.uint64(32),
                                      err_msg="tile ids of sorted keys")
        gkeys = (v["point_tiles"].astype(np.uint64) << np.uint64(32)) | \
            _bits(v["depths"][v["point_list"]]).astype(np.uint64)
        np.testing.assert_array_equal(gkeys, binning["keys"], err_msg="reconstructed 64-bit keys")
    np.testing.assert_array_equal(v["ranges"].astype(np.uint32), binning["ranges"], err_msg="ranges")
    if not check_image:
        return rep
    # ---- image: <= rgb_tol except pixels with a decision inside the exp()-ulp band -------------------------
    border = o["borderline"] > 0 if o.get("borderline") is not None else np.zeros_like(o["final_T"], bool)
    rep["borderline_pixels"] = int(border.sum())
    ok = ~border
    dc = np.abs(g["color"] - o["color"])
    dd = np.abs(g["invdepth"] - o["invdepth"])
    rep["rgb_max_abs"] = float(dc[:, ok].max()) if ok.any() else 0.0
    rep["invdepth_max_abs"] = float(dd[:, ok].max()) if ok.any() else 0.0
    rep["rgb_max_abs_all"] = float(dc.max())
    assert rep["rgb_max_abs"] <= rgb_tol, f"RGB max abs {rep['rgb_max_abs']} > {rgb_tol}"
    # inverse depth is a sum of alpha*T/z; same tolerance relative to its scale (1/z can exceed 1)
    scale = max(1.0, float(np.abs(o["invdepth"]).max()))
    assert rep["invdepth_max_abs"] <= rgb_tol * scale, f"invdepth max abs {rep['invdepth_max_abs']}"
    nc = v["n_contrib"].astype(np.uint32)
    rep["n_contrib_mismatch"] = int((nc != o["n_contrib"])[ok].sum())
    assert rep["n_contrib_mismatch"] == 0, "n_contrib differs on non-borderline pixels"
    dT = np.abs(v["final_T"] - o["final_T"])
    rep["final_T_max_abs"] = float(dT[ok].max()) if ok.any() else 0.0
    assert rep["final_T_max_abs"] <= rgb_tol
    # borderline pixels may flip one contribution: bounded by alpha_min * T * c <= 1/255 (+ termination 1e-4)
    rep["invdepth_max_abs_all"] = float(dd.max())
    assert rep["rgb_max_abs_all"] <= all_pixel_tol, f"all-pixel RGB error {rep['rgb_max_abs_all']} > {all_pixel_tol}"
    if all_pixel_tol <= rgb_tol:
        assert rep["invdepth_max_abs_all"] <= all_pixel_tol * scale, f"all-pixel invdepth {rep['invdepth_max_abs_all']}"
    assert rep["borderline_pixels"] <= max(8, 0.01 * border.size), "too many borderline pixels"
    return rep


def fov2tan(fov):
    return math.tan(fov * 0.5)
