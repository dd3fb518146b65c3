"""GPU diagnostic sweep (run via gpurun): forward parity cases with full tracebacks + timing of each stage.
Writes gpurun_out/diag_forward.json so a failed run still tells which stage diverged first."""
import json
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsworld_amd import scenes  # noqa: E402
from tests import helpers as hp  # noqa: E402

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
report = {"device": torch.cuda.get_device_name(0), "cases": {}}


def case(name, raw, cam, bg=(0, 0, 0), **kw):
    t0 = time.time()
    try:
        inp = hp.np_inputs(raw, cam)
        st = hp.oracle_settings(cam, **kw)
        bgn = np.asarray(bg, np.float32)
        o = hp.oracle_forward(inp, st, bgn)
        g = hp.gpu_forward(inp, st, bgn, debug=True)
        rep = hp.compare_forward(o, g, st)
        rep["ok"] = True
    except Exception as ex:  # noqa: BLE001
        rep = {"ok": False, "error": f"{type(ex).__name__}: {str(ex)[:2000]}", "trace": traceback.format_exc()[-3000:]}
    rep["seconds"] = time.time() - t0
    report["cases"][name] = rep
    print(name, json.dumps({k: v for k, v in rep.items() if k != "trace"}), flush=True)
    with open(os.path.join(ROOT, "gpurun_out", "diag_forward.json"), "w") as f:
        json.dump(report, f, indent=1)


case("tiny_1k_64", scenes.random_scene_camera_frame(1000, seed=1), scenes.identity_camera(64, 64, 60.0))
case("small_20k_128", scenes.random_scene_camera_frame(20000, seed=2), scenes.identity_camera(128, 128, 60.0))
case("ragged_20k_70x50", scenes.random_scene_camera_frame(20000, seed=3), scenes.identity_camera(70, 50, 70.0),
     bg=(0.2, 0.5, 0.9))
case("config1_100k_256", scenes.random_scene_camera_frame(100000, seed=0), scenes.identity_camera(256, 256, 60.0))
case("config2_full", scenes.tabletop_scene("xarm6_align"), scenes.sensor_camera("xarm6_align"))
print("DIAG DONE", all(c["ok"] for c in report["cases"].values()))
