"""The caller glue pinned to the reference ITSELF ("next" item 1a).

``tests/golden/wrapper_glue.npz`` is produced by ``tools/make_golden.py:wrapper_glue_golden``: the reference's own
``GSWorldWrapper.transform_gs_perlink`` / ``_render_gsworld`` / ``cam_maniskill2gs``
(/root/reference/gsworld/mani_skill/utils/wrappers/gs_world_wrapper.py:110-162, 232-275, 277-325) run unmodified on a
fake simulator, for num_envs = 1 and 3, with the arguments of every ``transform_gaussians`` call, every moved-part
tuple, the ``gs4render`` model each ``render()`` call was handed and the ``Camera`` constructor arguments recorded.

Checked here, on the CPU:
  * ``oracle/transform_ref.py`` + ``oracle/wrapper_glue_ref.py`` (the checkers of the GPU closed-loop tests) reproduce
    the moved parts and the assembled per-environment models -- including the reference's behaviour for a part that
    has exactly ``num_envs`` Gaussians (its shape tests then also fire for scaling and opacity);
  * the product's host-side pose arithmetic (``closed_loop.part_poses_from_sim``) reproduces the ``rot_mat`` /
    ``translation`` / ``scale`` arguments, ``camera.cam_maniskill2gs`` the camera's R, T and field of view.
"""
import os
import types

import numpy as np
import pytest
import torch

from gsworld_amd import camera as gcam
from gsworld_amd import closed_loop as cl
from oracle import transform_ref, wrapper_glue_ref

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "wrapper_glue.npz")


@pytest.fixture(scope="module")
def z():
    return np.load(GOLDEN)


def model_of(z):
    t = lambda k: torch.from_numpy(z["model." + k])  # noqa: E731
    return types.SimpleNamespace(_xyz=t("_xyz"), _features_dc=t("_features_dc"), _features_rest=t("_features_rest"),
                                 _opacity=t("_opacity"), _scaling=t("_scaling"), _rotation=t("_rotation"),
                                 _semantics=t("_semantics"))


def part_labels_of(z):
    """name -> label(s) exactly as the wrapper's ``gs_semantics`` holds them after ``transform_gs_perlink``."""
    consts = np.load(os.path.join(os.path.dirname(GOLDEN), "reference_constants.npz"))
    by_name = dict(zip(consts["xarm_semantic_names"].tolist(), consts["xarm_semantic_ids"].tolist()))
    labels = {}
    for n in z["E1.part_names"].tolist():
        if n in by_name:
            labels[n] = [7, 8] if n == "link6" else int(by_name[n])  # constants.py:462 (link6 carries the camera)
        else:
            labels[n] = int(z[f"label.{n}"])
    return labels


@pytest.mark.parametrize("E", [1, 3])
def test_transform_restatement_reproduces_every_moved_part(z, E):
    model = model_of(z)
    for name in z[f"E{E}.part_names"].tolist():
        p = f"E{E}.call.{name}."
        scale = torch.from_numpy(z[p + "scale"]) if p + "scale" in z.files else None
        got = transform_ref.transform_gaussians(model, torch.from_numpy(z[p + "selected"]), scale=scale,
                                                rot_mat=torch.from_numpy(z[p + "rot_mat"]),
                                                translation=torch.from_numpy(z[p + "translation"]), new_opacity=None)
        for attr, g in zip(("xyz", "scaling", "rotation", "opacity"), got):
            want = z[f"E{E}.moved.{name}.{attr}"]
            assert tuple(g.shape) == want.shape, (name, attr, tuple(g.shape), want.shape)
            np.testing.assert_allclose(g.numpy(), want, rtol=2e-6, atol=2e-6, err_msg=f"{name}.{attr}")


@pytest.mark.parametrize("E", [1, 3])
def test_glue_restatement_reproduces_the_model_each_render_call_received(z, E):
    model, labels = model_of(z), part_labels_of(z)
    names = z[f"E{E}.part_names"].tolist()
    moved = {n: tuple(torch.from_numpy(z[f"E{E}.moved.{n}.{a}"]) for a in ("xyz", "scaling", "rotation", "opacity"))
             for n in names}
    for e in range(E):
        gs = wrapper_glue_ref.assemble_env(model, labels, moved, e, E)
        for a in ("_xyz", "_scaling", "_rotation", "_opacity"):
            assert np.array_equal(getattr(gs, a).numpy(), z[f"E{E}.gs4render.env{e}.{a}"]), (e, a)
    # and from the simulator state on: transform_parts with the recorded matrices gives the same moved parts
    K = len(names)
    M = torch.eye(4).repeat(E, K, 1, 1)
    S = torch.ones(E, K)
    actors = [n for n in names if f"E{E}.call.{n}.scale" in z.files]
    for k, n in enumerate(names):
        M[:, k, :3, :3] = torch.from_numpy(z[f"E{E}.call.{n}.rot_mat"])
        M[:, k, :3, 3] = torch.from_numpy(z[f"E{E}.call.{n}.translation"])
        if n in actors:
            S[:, k] = torch.from_numpy(z[f"E{E}.call.{n}.scale"])
    again = wrapper_glue_ref.transform_parts(model, labels, M, S, actors)
    for n in names:
        for g, a in zip(again[n], ("xyz", "scaling", "rotation", "opacity")):
            want = z[f"E{E}.moved.{n}.{a}"]
            assert tuple(g.shape) == want.shape, (n, a)
            np.testing.assert_allclose(g.numpy(), want, rtol=2e-6, atol=2e-6)


def test_a_part_of_exactly_num_envs_gaussians_behaves_as_the_reference_does(z):
    """left_finger has 3 Gaussians: with num_envs = 3 the reference's ``rot_mat.size(0) == xyz.size(0)`` branch
    (gs_utils.py:331) rotates Gaussian g by ENVIRONMENT g's matrix, its rotations come back (3,4), and the write-back
    (gs_world_wrapper.py:246-265) broadcasts row i of scaling / rotation / opacity over the part.  The fixture records
    it; the restatement must follow (it is what the GPU tests are checked against)."""
    assert z["E3.moved.left_finger.rotation"].shape == (3, 4)
    assert z["E3.moved.left_finger.xyz"].shape == (3, 3, 3)
    lab = z["model._semantics"].reshape(-1)
    rows = np.where(lab == 15)[0]
    for e in range(3):
        sc = z[f"E3.gs4render.env{e}._scaling"][rows]
        assert np.array_equal(sc, np.repeat(z["model._scaling"][rows[e]][None], 3, 0))
    # num_envs = 1 with a one-Gaussian part (right_finger): the same tests fire and change nothing
    row = np.where(lab == 16)[0]
    assert np.array_equal(z["E1.gs4render.env0._scaling"][row], z["model._scaling"][row])
    assert np.array_equal(z["E1.gs4render.env0._opacity"][row], z["model._opacity"][row])


@pytest.mark.parametrize("E", [1, 3])
def test_pose_arithmetic_of_the_product_matches_what_the_reference_passed_on(z, E):
    links, actors = z["link_names"].tolist(), z["actor_names"].tolist()[:2]
    t = lambda k: torch.from_numpy(z[k])  # noqa: E731
    M, S = cl.part_poses_from_sim(
        t("sim2gs_arm_trans"), t(f"E{E}.link_pose_matrix"), t("link_scan_matrix"), t("object_offset.xarm_arm"),
        actor_now=t(f"E{E}.actor_pose_matrix")[:, :2], sim2gs_obj=torch.stack([t(f"sim2gs_object.{n}") for n in actors]),
        actor_offset=torch.stack([t(f"object_offset.{n}") for n in actors]),
        actor_scale=torch.stack([t(f"object_scale.{n}") for n in actors]))
    assert M.shape == (E, len(links) + 2, 4, 4) and S.shape == (E, len(links) + 2)
    for k, n in enumerate(links + actors):
        np.testing.assert_allclose(M[:, k, :3, :3].numpy(), z[f"E{E}.call.{n}.rot_mat"], rtol=0, atol=2e-6, err_msg=n)
        np.testing.assert_allclose(M[:, k, :3, 3].numpy(), z[f"E{E}.call.{n}.translation"], rtol=0, atol=2e-6, err_msg=n)
        if n in actors:
            np.testing.assert_allclose(S[:, k].numpy(), z[f"E{E}.call.{n}.scale"], rtol=0, atol=2e-6)
        else:
            assert f"E{E}.call.{n}.scale" not in z.files and float((S[:, k] - 1).abs().max()) == 0.0


@pytest.mark.parametrize("cam", ["right_cam", "wrist_cam"])
def test_camera_adapter_matches_the_reference_conversion(z, cam):
    """cam_maniskill2gs (:277-325): R, T, FoVx, FoVy as the reference handed them to ``Camera(...)``."""
    size = z[f"E1.cam.{cam}.size"]
    vp = gcam.cam_maniskill2gs(torch.from_numpy(z[f"E1.cam.{cam}.extrinsic_cv"]), torch.from_numpy(z["intrinsic_k"]),
                               int(size[0]), int(size[1]), torch.from_numpy(z["rigid_sim2real"]),
                               torch.from_numpy(z["scale_sim2real"]))
    want = gcam.view_params(z[f"E1.cam.{cam}.R"], z[f"E1.cam.{cam}.T"], float(z[f"E1.cam.{cam}.fov"][0]),
                            float(z[f"E1.cam.{cam}.fov"][1]), int(size[0]), int(size[1]))
    assert (vp.image_width, vp.image_height) == (640, 480)
    assert abs(vp.FoVx - want.FoVx) < 1e-7 and abs(vp.FoVy - want.FoVy) < 1e-7
    assert abs(vp.FoVx - 0.9715089) < 1e-6 and abs(vp.FoVy - 0.7551448) < 1e-6
    for a in ("world_view_transform", "full_proj_transform", "camera_center"):
        np.testing.assert_allclose(getattr(vp, a).numpy(), getattr(want, a).numpy(), rtol=0, atol=2e-6, err_msg=a)
    # and the E = 3 run built the same camera (the wrapper takes sensor parameters of environment 0, :281-282)
    assert np.array_equal(z[f"E3.cam.{cam}.R"], z[f"E1.cam.{cam}.R"])
