#!/usr/bin/env python
"""One screen of a bench.py line (or several): value, the arrangement's other figures, roofline, closed loop, parity."""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(path) if l.startswith("{")][-1])
    except (OSError, IndexError) as ex:
        print(path, "unreadable:", ex)
        continue
    c = d.get("config", {})
    tb = c.get("timed_blocks", {})
    print(path, "value", round(d["value"]), "blocks", tb.get("blocks"), [round(x) for x in tb.get("frames_per_s_min_median_max", [])],
          "total-time", round(tb.get("frames_per_s_total_time") or 0), "one", round(c.get("one_frame_in_flight_frames_per_s") or 0),
          "one stream", round(c.get("one_stream_frames_per_s") or 0), "verified after", c.get("frame_slots_verified_after_timed_region"))
    r = d.get("roofline", {})
    print("  roofline", {k: r.get(k) for k in ("frac", "kernel_ms", "frames_per_launch", "traffic_source")},
          "one/launch", (r.get("one_frame_per_launch") or {}).get("frac"), "frame", (d.get("frame_roofline") or {}).get("frac_of_8TBs"),
          "p50", (d.get("frame_roofline") or {}).get("frame_ms_p50"))
    for k in ("dense_view", "closed_loop", "moving_camera", "cpu_baseline"):
        if k in d:
            print("  ", k, {kk: vv for kk, vv in d[k].items() if kk not in ("workload", "sample", "three_steps_in_flight", "kernels")})
    if "parity" in d:
        p = d["parity"]
        print("   parity", p.get("worst_pixel_all_scenes"), p.get("worst_pixel_off_borderline"), p.get("per_scene_worst"),
              (p.get("config5_gradients") or {}).get("worst_normalised_error"))
    if "train_step" in d:
        print("   train", {k: (v.get("ms_per_step") if isinstance(v, dict) else None) for k, v in d["train_step"].items()})
