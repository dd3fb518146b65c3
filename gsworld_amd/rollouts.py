"""Synthetic rollouts: seeded stand-ins for the simulator, used where the closed loop is tested and measured.

SAPIEN / PhysX do not run on a headless GPU box, so :func:`rollout_poses` stands in for the simulator of BASELINE.json
configs[2]: the robot-link poses of a seeded random-action rollout
(/root/reference/examples/maniskill/gsworld_rand_action_tabletop.py:99-133) from forward kinematics of the reference's
xarm6 URDF, committed as a data fixture (:func:`xarm6_rollout`), pushed through the wrapper's own pose arithmetic
(:func:`gsworld_amd.closed_loop.part_poses_from_sim`); the tracked objects, which only contacts would move, take a seeded
random walk (:func:`random_walk_poses`, which the smaller tests also use for every part).  Nothing here is on the render
path: :mod:`gsworld_amd.closed_loop` takes whatever matrices its caller hands over.
"""
from __future__ import annotations

import torch

from .closed_loop import part_poses_from_sim


def small_rigid(gen: torch.Generator, k: int, angle: float = 0.05, shift: float = 0.01) -> torch.Tensor:
    """``k`` random small rigid 4x4 increments (Rodrigues rotation of a N(0, angle^2) axis-angle + N(0, shift^2) shift)."""
    w = torch.randn(k, 3, generator=gen) * angle
    th = w.norm(dim=1, keepdim=True).clamp_min(1e-8)
    a = w / th
    Kx = torch.zeros(k, 3, 3)
    Kx[:, 0, 1], Kx[:, 0, 2], Kx[:, 1, 0] = -a[:, 2], a[:, 1], a[:, 2]
    Kx[:, 1, 2], Kx[:, 2, 0], Kx[:, 2, 1] = -a[:, 0], -a[:, 1], a[:, 0]
    R = torch.eye(3) + torch.sin(th)[:, :, None] * Kx + (1 - torch.cos(th))[:, :, None] * (Kx @ Kx)
    M = torch.eye(4).repeat(k, 1, 1)
    M[:, :3, :3] = R
    M[:, :3, 3] = torch.randn(k, 3, generator=gen) * shift
    return M


def random_walk_poses(sim2gs: torch.Tensor, num_parts: int, num_actors: int, steps: int, seed: int = 0,
                      num_envs: int = 1):
    """Seeded stand-in for the simulator of a random-action rollout: yields, per step, the part matrices the wrapper
    would build -- ``sim2gs @ pose_now @ inv(pose_scan = I) @ inv(sim2gs)`` for links (gs_world_wrapper.py:120) and the
    rigid part + uniform scale of the same product for the last ``num_actors`` parts (``:146-156``) -- as
    ``(matrices (K,4,4) | (E,K,4,4), scales (K,) | (E,K))`` on the host."""
    from .camera import extract_rigid_transform

    gen = torch.Generator().manual_seed(seed)
    K = num_parts
    sim2gs = sim2gs.to(torch.float32)
    inv = torch.linalg.inv(sim2gs)
    now = torch.eye(4).repeat(num_envs, K, 1, 1)
    for _ in range(steps):
        now = now @ small_rigid(gen, num_envs * K).reshape(num_envs, K, 4, 4)
        full = sim2gs @ now @ inv
        rigid, scale, _, _ = extract_rigid_transform(full.reshape(-1, 4, 4))
        rigid = rigid.reshape(num_envs, K, 4, 4)
        scales = torch.ones(num_envs, K)
        if num_actors:
            scales[:, K - num_actors:] = scale.reshape(num_envs, K)[:, K - num_actors:]
        if num_envs == 1:
            yield rigid[0].contiguous(), scales[0].contiguous()
        else:
            yield rigid.contiguous(), scales.contiguous()


def xarm6_rollout(path: str | None = None) -> dict:
    """The committed link-pose fixture of a seeded random-action xarm6 rollout (tests/golden/xarm6_rollout.npz, written
    by tools/make_xarm6_rollout.py from forward kinematics of the reference's URDF; 1 reset + 200 steps): the arrays as
    torch tensors plus ``parts`` -- link name -> label(s) per ``xarm_gs_semantics`` for the 15 links that move
    (``world`` carries the background label 0 and never moves: its matrix is the identity by construction)."""
    import os

    import numpy as np

    if path is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                            "xarm6_rollout.npz")
    z = np.load(path)
    names = [str(n) for n in z["link_names"]]
    keep = [i for i, n in enumerate(names) if n != "world"]
    parts = {}
    for i in keep:
        lab = [int(v) for v in z["labels"][i] if v >= 0]
        parts[names[i]] = lab[0] if len(lab) == 1 else lab
    return dict(parts=parts, link_now=torch.from_numpy(z["link_now"][:, keep]), link_scan=torch.from_numpy(z["link_scan"][keep]),
                labels=torch.from_numpy(z["labels"][keep]),  # (per kept link: one or two semantic labels, -1 = none)
                sim2gs_arm=torch.from_numpy(z["sim2gs_arm"]), link_offset=torch.from_numpy(z["link_offset"]),
                qpos=torch.from_numpy(z["qpos"]))


def rollout_poses(rollout: dict, num_actors: int, steps: int, seed: int = 0, num_envs: int = 1):
    """Per-step part matrices of the closed loop from the FK rollout: the robot links go through
    :func:`part_poses_from_sim` (the wrapper's own arithmetic, gs_world_wrapper.py:114-120) at step ``t`` of the
    fixture; the ``num_actors`` tracked objects, which only the physics could move, keep the seeded small random walk
    of :func:`random_walk_poses`.  Environment ``e`` plays the same trajectory ``17 e`` steps ahead, reflecting at the
    ends (no jump), so environments differ the way independently acting robots would.  Yields what
    :func:`random_walk_poses` yields: ``(matrices (K,4,4) | (E,K,4,4), scales (K,) | (E,K))``, links first."""
    T = rollout["link_now"].shape[0]

    def at(t):
        t = t % (2 * (T - 1))
        return t if t < T else 2 * (T - 1) - t

    actors = random_walk_poses(rollout["sim2gs_arm"], num_actors, num_actors, steps, seed, num_envs) if num_actors else None
    for t in range(steps):
        now = torch.stack([rollout["link_now"][at(t + 17 * e)] for e in range(num_envs)])
        m, s = part_poses_from_sim(rollout["sim2gs_arm"], now, rollout["link_scan"], rollout["link_offset"])
        if actors is not None:
            am, asc = next(actors)
            m = torch.cat((m, am.reshape(num_envs, num_actors, 4, 4)), 1)
            s = torch.cat((s, asc.reshape(num_envs, num_actors)), 1)
        if num_envs == 1:
            yield m[0].contiguous(), s[0].contiguous()
        else:
            yield m.contiguous(), s.contiguous()


def xarm6_rollout_parts(rollout: dict):
    """``(parts, actors)`` for :class:`ClosedLoopRenderer` over the synthetic table-top scene: the fixture's 15 moving
    links under their ``xarm_gs_semantics`` labels (1..16) and the two tracked actors (labels 17, 18)."""
    parts = dict(rollout["parts"])
    parts.update({"005_tomato_soup_can": 17, "dtc_green_can": 18})
    return parts, ("005_tomato_soup_can", "dtc_green_can")


def xarm6_parts():
    """16 robot links (labels 1..16) and 2 tracked actors (labels 17, 18), the way ``xarm_gs_semantics`` /
    ``obj_gs_semantics`` label the synthetic table-top scenes of :mod:`gsworld_amd.scenes`."""
    parts = {f"link{k}": k for k in range(1, 17)}
    parts.update({"005_tomato_soup_can": 17, "dtc_green_can": 18})
    return parts, ("005_tomato_soup_can", "dtc_green_can")
