"""In-process A/B of GsrSettings selectors on the headline frame: frames/s at 1 and 3 frames in flight under hipGraph
replay (the bench's own loop), the scene loaded once, configurations interleaved over several rounds.

    python tools/ab_frame.py "render_blocks_per_cu=6 render_blocks_per_cu=4 render_blocks_per_cu=3" [--view dense] [--steps 300]

Every configuration's frame is compared byte for byte with the first one's."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsworld_amd import _lib, scenes  # noqa: E402
from gsworld_amd.renderer import FrameRenderer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("configs", nargs="?", default="render_blocks_per_cu=6 render_blocks_per_cu=4")
ap.add_argument("--view", default="sensor")
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--in-flight", default="1,3")
ap.add_argument("--tag", default="")
ap.add_argument("--layouts", default="0", help="comma list: 0 = model as given, 1 = SceneLayout (Morton order + block culling), "
                "2 = block bounds on the unsorted model")
ap.add_argument("--pre-alloc", type=int, default=0, help="MiB of device memory allocated (and kept) before anything else")
ap.add_argument("--pre-streams", type=int, default=0, help="streams taken from torch's pool before the lanes'")
args = ap.parse_args()

dev = torch.device("cuda:0")
W, H = 640, 480
_keep = [torch.empty(args.pre_alloc << 20, dtype=torch.uint8, device=dev)] if args.pre_alloc else []
_keep += [torch.cuda.Stream(dev) for _ in range(args.pre_streams)]
for _s in _keep[1 if args.pre_alloc else 0:]:  # (a stream gets its hardware queue when it is first used)
    with torch.cuda.stream(_s):
        _keep.append(torch.zeros(1, device=dev))
torch.cuda.synchronize()
raw = scenes.tabletop_scene("xarm6_align")
cam = (scenes.sensor_camera("xarm6_align", W, H) if args.view == "sensor"
       else scenes.dense_view_camera("xarm6_align", W, H)).to(dev)
means, shs, op, sc, rot = (t.to(dev) for t in raw.activated())
bg = torch.zeros(3, device=dev)
configs = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in c.split(",") if kv) for c in args.configs.split()]
from gsworld_amd import layout as gl  # noqa: E402

models = {}
for lay in sorted({int(x) for x in args.layouts.split(",")}):
    if lay == 0:
        models[0] = (means, shs, op, sc, rot, None)
        continue
    t0 = time.perf_counter()
    L = gl.SceneLayout.build(means, sc, rot, reorder=(lay == 1), shs=shs, opacities=op)
    torch.cuda.synchronize()
    print(f"layout {lay}: built in {time.perf_counter() - t0:.3f} s", file=sys.stderr)
    a = L.arrays
    models[lay] = (a["means3D"], a["shs"], a["opacities"], a["scales"], a["rotations"], L.layout)
configs = [dict(c, layout=lay) for c in configs for lay in sorted(models)]
flights = [int(x) for x in args.in_flight.split(",")]
base = dict(_lib.TUNING)


def build(cfg, S):
    cfg = dict(cfg)
    means, shs, op, sc, rot, lay = models[cfg.pop("layout")]
    _lib.TUNING.update(base)
    _lib.TUNING.update(cfg)
    rs = [FrameRenderer(dev, forward_only=True, want_radii=False) for _ in range(S)]
    outs = [torch.zeros((H, W, 3), dtype=torch.uint8, device=dev) for _ in range(S)]
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    fns = [(lambda l=l: rs[l].render(cam, means, op, shs=shs, scales=sc, rotations=rot, bg=bg, rgb8_out=outs[l],
                                     layout=lay))
           for l in range(S)]
    graphs = []
    for l in range(S):
        for _ in range(2):
            fns[l]()
            rs[l].ensure_valid(fns[l])
        streams[l].wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(streams[l]):
            fns[l]()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=streams[l]):
            fns[l]()
        graphs.append(g)
    torch.cuda.synchronize()
    return rs, outs, streams, graphs


def run(streams, graphs, steps):
    S = len(graphs)
    for i in range(30):
        with torch.cuda.stream(streams[i % S]):
            graphs[i % S].replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % S]):
            graphs[i % S].replay()
    torch.cuda.synchronize()
    return steps / (time.perf_counter() - t0)


built = {}
ref = None
for ci, cfg in enumerate(configs):
    for S in flights:
        built[(ci, S)] = build(cfg, S)
        frame = built[(ci, S)][1][0]
        if ref is None:
            ref = frame.clone()
        elif not torch.equal(frame, ref):
            raise SystemExit(f"config {cfg} renders a different frame")
res = {}
for rnd in range(args.rounds):
    for ci, cfg in enumerate(configs):
        for S in flights:
            _, _, streams, graphs = built[(ci, S)]
            res.setdefault((ci, S), []).append(run(streams, graphs, args.steps))
out = {}
for ci, cfg in enumerate(configs):
    name = ",".join(f"{k}={v}" for k, v in cfg.items()) or "default"
    out[name] = {f"in_flight_{S}": sorted(res[(ci, S)])[len(res[(ci, S)]) // 2] for S in flights}
    print(f"{args.tag} {name}: " + "  ".join(f"S={S}: {out[name][f'in_flight_{S}']:.0f} fps "
                                              f"({min(res[(ci, S)]):.0f}-{max(res[(ci, S)]):.0f})" for S in flights))
print(json.dumps({"tag": args.tag, "view": args.view, "results": out}))
