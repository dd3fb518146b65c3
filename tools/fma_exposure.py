#!/usr/bin/env python
"""How much of the forward pass depends on FMA-contraction choices nobody here can see?  

The CUDA reference is compiled by nvcc with -fmad=true; which multiply-adds it fuses is invisible from this
container.  ``oracle/gs_oracle.c`` FIXES one choice (the canonical order every parity test checks the HIP kernels
against).  This tool renders BASELINE.json configs[0] and configs[1] with three builds of that one source --

  canonical   explicit fmaf() where a left-to-right contraction of the published expressions would fuse, -ffp-contract=off
  nofma       no fused operation anywhere (nvcc -fmad=false)
  contract    every a * b + c written plainly and left to gcc under -ffp-contract=fast
  assoc_upstream  canonical, but the blend in upstream's association ((rgb * alpha) * T) + C instead of rgb * (alpha * T) + C

-- and counts what changes against the canonical build: radii, tile rects, num_rendered, point_list entries,
pixels beyond north_star's 1e-4.  The spread between the builds is the exposure of "bit-exact tile / key indices
vs the CUDA rasterizer" to a choice that cannot be pinned here.  CPU only.   python tools/fma_exposure.py [--json out]
"""
import argparse
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {"canonical": "", "nofma": "oracle/_variants/libgs_oracle_nofma.so",
            "contract": "oracle/_variants/libgs_oracle_contract.so",
            # the blend in upstream's association, ((rgb * alpha) * T) + C, instead of the canonical rgb * (alpha * T) + C
            "assoc_upstream": "oracle/_variants/libgs_oracle_assoc.so"}


def worker(config: int, out_path: str):
    import torch  # noqa: F401

    from gsworld_amd import scenes
    from tests import helpers as hp

    if config == 0:
        raw, cam = scenes.random_scene_camera_frame(100_000, seed=0), scenes.identity_camera(256, 256, 60.0)
    else:
        raw, cam = scenes.tabletop_scene("xarm6_align"), scenes.sensor_camera("xarm6_align")
    inp, st = hp.np_inputs(raw, cam), hp.oracle_settings(cam)
    o = hp.oracle_forward(inp, st, np.zeros(3, np.float32), border_eps=0.0, border_eps_T=0.0)
    g, b = o["geom"], o["binning"]
    np.savez(out_path, radii=g["radii"], rects=g["rects"], tiles_touched=g["tiles_touched"],
             depth_bits=g["depths"].view(np.uint32), point_list=b["point_list"], keys=b["keys"], ranges=b["ranges"],
             color=o["color"], invdepth=o["invdepth"], n_contrib=o["n_contrib"])


def compare(a, b):
    vis_a, vis_b = a["radii"] > 0, b["radii"] > 0
    both = vis_a & vis_b
    r = dict(
        visible=int(vis_a.sum()), visibility_flips=int((vis_a != vis_b).sum()),
        radii_differ=int((a["radii"] != b["radii"]).sum()),
        rects_differ=int((a["rects"] != b["rects"]).any(1).sum()),
        tiles_touched_differ=int((a["tiles_touched"] != b["tiles_touched"]).sum()),
        depth_key_bits_differ=int((a["depth_bits"][both] != b["depth_bits"][both]).sum()),
        num_rendered=[int(len(a["point_list"])), int(len(b["point_list"]))],
    )
    if len(a["point_list"]) == len(b["point_list"]):
        r["point_list_positions_differ"] = int((a["point_list"] != b["point_list"]).sum())
        r["keys_differ"] = int((a["keys"] != b["keys"]).sum())
    else:
        r["point_list_positions_differ"] = None  # different lengths: every position after the first change shifts
    # instance SETS per tile: which (tile, Gaussian) pairs exist in one build only
    ta = np.repeat(np.arange(len(a["ranges"])), (a["ranges"][:, 1] - a["ranges"][:, 0]).astype(np.int64))
    tb = np.repeat(np.arange(len(b["ranges"])), (b["ranges"][:, 1] - b["ranges"][:, 0]).astype(np.int64))
    pa = ta.astype(np.int64) << 32 | a["point_list"].astype(np.int64)
    pb = tb.astype(np.int64) << 32 | b["point_list"].astype(np.int64)
    r["instances_in_one_build_only"] = int(len(np.setxor1d(pa, pb)))
    d = np.abs(a["color"] - b["color"]).max(0)
    r["pixels_rgb_gt_1e-4"] = int((d > 1e-4).sum())
    r["rgb_max_abs"] = float(d.max())
    r["invdepth_max_abs"] = float(np.abs(a["invdepth"] - b["invdepth"]).max())
    r["n_contrib_differ"] = int((a["n_contrib"] != b["n_contrib"]).sum())
    r["pixels"] = int(d.size)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--worker", nargs=2, default=None)
    ap.add_argument("--configs", default="0,1")
    args = ap.parse_args()
    if args.worker:
        worker(int(args.worker[0]), args.worker[1])
        return
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "all", "variants"])
    tmp = os.path.join(ROOT, "tools", "scratch")
    os.makedirs(tmp, exist_ok=True)
    report = {}
    for c in [int(x) for x in args.configs.split(",")]:
        outs = {}
        for name, lib in VARIANTS.items():
            path = os.path.join(tmp, f"fma_{c}_{name}.npz")
            env = dict(os.environ)
            if lib:
                env["GS_ORACLE_LIB"] = os.path.join(ROOT, lib)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--worker", str(c), path], env=env)
            outs[name] = dict(np.load(path))
        report[f"configs[{c}]"] = {n: compare(outs["canonical"], outs[n]) for n in VARIANTS if n != "canonical"}
        print(json.dumps({f"configs[{c}]": report[f"configs[{c}]"]}, indent=1))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
