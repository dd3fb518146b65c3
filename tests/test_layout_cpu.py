"""gsworld_amd/layout.py on the CPU: the permutation, the block bounds, and the PROPERTY the kernel's block test relies
on (oracle/cull_ref.py restates it): no Gaussian of a culled block has radii > 0 in the oracle's preprocess."""
import numpy as np
import torch

from gsworld_amd import layout as gl, scenes
from gsworld_amd.camera import look_at_view
from oracle import cull_ref, gs_oracle as go


def _radii(raw_act, cam, scale_modifier=1.0):
    means, shs, op, sc, rot = raw_act
    st = go.Settings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, scale_modifier)
    o = go.preprocess(st, means.numpy(), shs.numpy(), None, op.numpy().reshape(-1), sc.numpy(), rot.numpy(), None,
                      cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), cam.camera_center.numpy())
    return o["radii"]


def test_morton_order_is_a_stable_permutation_and_blocks_hold_their_gaussians():
    raw = scenes.tabletop_scene("xarm6_align", n=50_000, seed=3)
    means, shs, op, sc, rot = raw.activated()
    L = gl.SceneLayout.build(means, sc, rot, labels=raw.semantics.reshape(-1), shs=shs, opacities=op)
    assert torch.equal(torch.sort(L.perm).values, torch.arange(raw.num))
    a = L.arrays
    assert torch.equal(a["means3D"], means[L.perm]) and torch.equal(a["shs"], shs[L.perm])
    b = L.cull_blocks
    nb = (raw.num + 255) // 256
    assert b.shape == (nb, 8) and L.orig_index.dtype == torch.int32
    m = torch.cat([a["means3D"], a["means3D"][-1:].expand(nb * 256 - raw.num, 3)]).reshape(nb, 256, 3)
    assert bool((m >= b[:, None, 0:3]).all()) and bool((m <= b[:, None, 3:6]).all())
    rho = torch.cat([a["scales"].amax(1), a["scales"].amax(1)[-1:].expand(nb * 256 - raw.num)]).reshape(nb, 256)
    assert bool((rho.amax(1) <= b[:, 6]).all())
    lab = a["labels"]
    pure = ~torch.isnan(b[:, 7])
    assert float(pure.float().mean()) > 0.85  # (20 parts of ~620 Gaussians each: a seam per part)
    for k in torch.nonzero(pure).reshape(-1)[:50].tolist():
        assert bool((lab[256 * k:256 * k + 256].long() == int(b[k, 7])).all())


def test_no_culled_block_holds_a_visible_gaussian():
    """Random cameras around and inside the scene, two image sizes, a scale modifier: the restated block test never culls a
    block in which the oracle finds radii > 0; and it does cull (a test that never culls proves nothing)."""
    raw = scenes.tabletop_scene("xarm6_align", n=120_000, seed=4)
    act = raw.activated()
    means, shs, op, sc, rot = act
    L = gl.SceneLayout.build(means, sc, rot)
    a = L.arrays
    act_p = (a["means3D"], shs[L.perm], op[L.perm], a["scales"], a["rotations"])
    rng = np.random.default_rng(0)
    cams = [scenes.sensor_camera("xarm6_align"), scenes.dense_view_camera("xarm6_align", 640, 480)]
    for _ in range(10):
        eye = rng.uniform([-0.5, -1.0, 0.05], [1.5, 1.0, 1.5])
        tgt = eye + rng.normal(size=3)
        W, H = [(640, 480), (200, 152), (70, 50)][rng.integers(3)]
        cams.append(look_at_view(eye, tgt, [0, 0, 1], rng.uniform(0.4, 1.6), rng.uniform(0.4, 1.4), W, H))
    total_culled = 0
    for i, cam in enumerate(cams):
        mod = 1.0 if i % 3 else 1.7
        radii = _radii(act_p, cam, mod)
        culled = cull_ref.blocks_culled(L.cull_blocks.numpy(), cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(),
                                        cam.image_width, cam.image_height, cam.tanfovx, cam.tanfovy, 0.05, mod)
        nb = culled.shape[0]
        r = np.concatenate([radii, np.zeros(nb * 256 - radii.shape[0], radii.dtype)]).reshape(nb, 256)
        assert int((r[culled] > 0).sum()) == 0, f"camera {i}: a culled block holds visible Gaussians"
        total_culled += int(culled.sum())
    assert total_culled > 0.2 * len(cams) * nb


def test_unsorted_bounds_are_valid_and_degenerate_inputs_are_never_culled():
    raw = scenes.random_scene_camera_frame(10_000, seed=5)
    means, shs, op, sc, rot = raw.activated()
    b = gl.build_cull_blocks(means, sc, rot)
    cam = scenes.identity_camera(128, 128, 60.0)
    radii = _radii((means, shs, op, sc, rot), cam)
    culled = cull_ref.blocks_culled(b.numpy(), cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), 128, 128,
                                    cam.tanfovx, cam.tanfovy)
    nb = culled.shape[0]
    r = np.concatenate([radii, np.zeros(nb * 256 - radii.shape[0], radii.dtype)]).reshape(nb, 256)
    assert int((r[culled] > 0).sum()) == 0
    bad = b.clone()
    bad[0, 0] = float("nan")
    bad[1, 6] = float("inf")
    bad[2, 6] = float("nan")
    c2 = cull_ref.blocks_culled(bad.numpy(), cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), 128, 128,
                                cam.tanfovx, cam.tanfovy)
    assert not c2[0] and not c2[2]
    # rho = inf: only the near-plane test can still cull
    assert not c2[1] or bad[1, [2, 5]].max() < 0.05


def test_blocks_built_without_labels_carry_no_common_label():
    """a layout built without labels must never let a block be tested under ONE pose when it is later
    rendered with a part transform -- its label field is NaN ("members differ"), which the kernel never culls by while
    part labels are given, and does not read at all without them."""
    raw = scenes.tabletop_scene("xarm6_align", n=5_000, seed=3)
    means, shs, op, sc, rot = raw.activated()
    b = gl.build_cull_blocks(means, sc, rot, labels=None)
    assert bool(torch.isnan(b[:, 7]).all())
    L = gl.SceneLayout.build(means, sc, rot, shs=shs, opacities=op)
    assert bool(torch.isnan(L.cull_blocks[:, 7]).all())
