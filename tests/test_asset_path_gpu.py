"""SURVEY.md 8f-3 end to end ON THE DEVICE: the on-disk asset path of a GSWorld scene -- ``configs/<scene>.json`` in
the schema of /root/reference/configs/xarm6_align.json:1-22 (one robot/scene scan with an ``.npy`` of per-point link
labels + object PLYs labelled by an integer) -> ``merge_scene`` (the counterpart of
/root/reference/gsworld/utils/gaussian_merger.py:213-274 + semantic_3dgs_wrapper.py:100-167) -> a frame through the
drop-in ``render()``, compared with the oracle on the merged arrays -> the merged labels driving one closed-loop step
(``FusedPartTransform`` inside ``ClosedLoopRenderer``) against the wrapper's glue restated in torch.

``tests/test_ply_cpu.py`` pins the merger against the reference's own merged tensors (tests/golden/merger.npz); this
test joins the pieces on the GPU."""
import json
import os
import sys
import types

import numpy as np
import pytest
import torch

from gsworld_amd import closed_loop as cl
from gsworld_amd import merger, ply, scenes
from gsworld_amd.camera import look_at_view
from oracle import wrapper_glue_ref as ref
from tests import helpers as hp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ATTRS = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def _subset(raw, rows):
    return types.SimpleNamespace(
        _xyz=raw.xyz[rows], _features_dc=raw.features_dc[rows], _features_rest=raw.features_rest[rows],
        _opacity=raw.opacity[rows].reshape(-1, 1, 1), _scaling=raw.scaling[rows], _rotation=raw.rotation[rows],
        _semantics=raw.semantics[rows])


@pytest.fixture()
def gs_paths():
    added = [os.path.join(ROOT, "gsworld_amd", "dropin"), os.path.join(ROOT, "gsworld_amd", "gs_compat")]
    for p in added:
        sys.path.insert(0, p)
    yield
    for p in added:
        sys.path.remove(p)


def test_config_json_to_merged_model_to_frames(cuda_device, gs_paths, tmp_path):
    from arguments import PipelineParams
    from gaussian_renderer import render
    from scene.cameras import Camera

    dev = cuda_device
    raw = scenes.tabletop_scene("xarm6_align", n=120_000, seed=4)
    lab = raw.semantics.reshape(-1)
    # the synthetic scene labels its 20 clusters 1..20: 1..16 are the robot links, 17 / 18 become the two tracked
    # objects of AlignXArmEnv-v1 (configs/xarm6_align.json: dtc_green_can = 201, tomato_soup_can = 110)
    rows = {"scene/gs/xarm6/xarm6.ply": torch.where((lab != 17) & (lab != 18))[0],
            "objs/dtc_green_can.ply": torch.where(lab == 18)[0], "objs/tomato_soup_can.ply": torch.where(lab == 17)[0]}
    for rel, r in rows.items():
        os.makedirs(os.path.dirname(tmp_path / rel), exist_ok=True)
        ply.write_gaussian_ply(str(tmp_path / rel), _subset(raw, r), with_semantics=False)
    np.save(str(tmp_path / "scene/gs/xarm6/xarm6_semantics_gs.npy"),
            lab[rows["scene/gs/xarm6/xarm6.ply"]].numpy().astype(np.int64))
    config = {"models": [
        {"data_path": "./scene/gs/xarm6/xarm6.ply", "semantic_labels": "./scene/gs/xarm6/xarm6_semantics_gs.npy",
         "transformation": []},
        {"data_path": "./objs/dtc_green_can.ply", "semantic_labels": 201, "transformation": []},
        {"data_path": "./objs/tomato_soup_can.ply", "semantic_labels": 110, "transformation": []}]}
    with open(tmp_path / "xarm6_align.json", "w") as f:
        json.dump(config, f)

    merged = merger.merge_scene(str(tmp_path / "xarm6_align.json"), asset_dir=str(tmp_path), device=dev)
    order = torch.cat(list(rows.values()))
    want = _subset(raw, order)
    for a in ATTRS:  # float32 columns survive the PLY round trip bit for bit
        got = getattr(merged, a)
        assert got.device.type == "cuda" and tuple(got.shape) == tuple(getattr(want, a).shape), a
        assert torch.equal(got.cpu(), getattr(want, a)), a
    want_labels = torch.cat((lab[rows["scene/gs/xarm6/xarm6.ply"]], torch.full((len(rows["objs/dtc_green_can.ply"]),), 201.0),
                             torch.full((len(rows["objs/tomato_soup_can.ply"]),), 110.0)))
    assert merged._semantics.shape == (raw.num, 1)
    assert torch.equal(merged._semantics.reshape(-1).cpu().float(), want_labels)

    # ---- a frame through the drop-in render(), against the oracle on the merged arrays
    ref_cam = scenes.sensor_camera("xarm6_align")
    W2C = ref_cam.world_view_transform.T
    cam = Camera(resolution=(640, 480), colmap_id=0, R=W2C[:3, :3].T.numpy(), T=W2C[:3, 3].numpy(), FoVx=ref_cam.FoVx,
                 FoVy=ref_cam.FoVy, depth_params=None, image=None, invdepthmap=None, image_name="right_cam", uid=0,
                 data_device=dev)
    pipe = PipelineParams().extract(types.SimpleNamespace())
    out = render(cam, merged, pipe, torch.zeros(3, device=dev), use_trained_exp=False, separate_sh=False)
    img = out["render"].detach()
    raw_m = scenes.RawGaussians(want._xyz, want._features_dc, want._features_rest, want._opacity.reshape(-1, 1),
                                want._scaling, want._rotation, want_labels.reshape(-1, 1))
    inp = hp.np_inputs(raw_m, ref_cam)
    o = hp.oracle_forward(inp, hp.oracle_settings(ref_cam), np.zeros(3, np.float32))
    ok = o["borderline"] == 0
    assert np.abs(img.cpu().numpy() - np.clip(o["color"], 0, 1))[:, ok].max() <= 1e-4
    assert int((out["radii"] > 0).sum()) == int((o["geom"]["radii"] > 0).sum())
    assert float(img.max()) > 0.4

    # ---- the merged labels drive a closed-loop step: links 1..16 + the two objects, moved by a seeded pose walk
    parts = {f"link{k}": k for k in range(1, 17)}
    parts.update({"005_tomato_soup_can": 110, "dtc_green_can": 201})
    actors = ("005_tomato_soup_can", "dtc_green_can")
    cams = {"right_cam": ref_cam,
            "wrist_cam": look_at_view([0.55, 0.35, 0.25], [0.35, 0.05, 0.05], [0, 0, 1], 0.9715089, 0.7551448, 640, 480)}
    loop = cl.ClosedLoopRenderer(merged, parts, cams, scaled_parts=actors, num_envs=1, device=dev)
    poses = list(cl.random_walk_poses(torch.tensor(scenes.SIM2GS_XARM_TRANS), len(parts), len(actors), steps=3, seed=2))
    loop.reset(*poses[0])
    from tests.test_closed_loop_gpu import _compare, _rasterize

    cams_d = {k: v.to(dev) for k, v in cams.items()}
    still = {k: v.clone() for k, v in loop.frames.items()}
    for M, s in poses[1:]:
        frames = {k: v.clone() for k, v in loop.step(M, s).items()}
        _compare(frames, ref.render_step(merged, parts, cams_d, M, s, _rasterize, actors), "merged scene step")
    assert not any(st.overflow for st in loop.ensure_valid())
    assert not torch.equal(frames["right_cam"], still["right_cam"])  # the labelled parts really moved
