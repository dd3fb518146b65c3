#!/usr/bin/env python3
"""Writes tests/golden/xarm6_rollout.npz: the robot-link poses of a seeded random-action rollout of the xarm6 of
BASELINE.json configs[2] (AlignXArmEnv-v1, 1 reset + 200 steps), from forward kinematics of the reference's own URDF.

Runs HERE only (it reads /root/reference); the committed fixture is data -- 4x4 link poses per step -- and is what
``bench.py`` / ``tools/closed_loop_surrogate.py`` / the closed-loop GPU tests feed through
``gsworld_amd.closed_loop.part_poses_from_sim`` in place of the earlier independent random walk of every part.

What is taken from the reference (read at generation time, nothing copied into the repo):
  * the kinematic tree: joint origins / axes / limits of
    gsworld/mani_skill/assets/robots/xarm6/xarm6_description/xarm6_uf_gripper.urdf (the ``urdf_path`` of the
    ``xarm6_uf_gripper`` agent, agents/robots/xarm6/xarm6_uf_gripper.py:19);
  * ``xarm_gs_qpos`` (the qpos the robot was scanned in -> ``gs_link_pose_mats``, gs_world_wrapper.py:94-103) and
    ``xarm_task_init_qpos`` (the reset qpos), constants.py:75-103;
  * ``xarm_gs_semantics`` (link name -> label(s)), ``sim2gs_xarm_trans``, ``object_offset["xarm_arm"]``;
  * the rollout itself is examples/maniskill/gsworld_rand_action_tabletop.py:40-140: ``env.action_space.sample()``
    every step under the default ``pd_joint_pos`` control mode (arm: absolute joint targets, un-normalised, i.e.
    uniform over the joint limits; gripper: one mimic target for all six finger joints).

What is NOT the reference: there is no PhysX here, so the PD response is a kinematic stand-in -- each control step
(1/20 s) a joint covers ``1 - exp(-dt * stiffness / damping)`` of the distance to its target (arm 1e4 / 1e3 -> 0.39;
gripper 1e5 / 2e3 -> 0.92), capped by the URDF velocity limit x dt, and clipped to the joint limits; no gravity, no
contacts.  The link poses are therefore "a plausible random-action arm trajectory with the true kinematic coupling
between links", not the simulator's trajectory.
"""
from __future__ import annotations

import importlib.util
import math
import os
import xml.etree.ElementTree as ET

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
URDF = f"{REF}/gsworld/mani_skill/assets/robots/xarm6/xarm6_description/xarm6_uf_gripper.urdf"
STEPS = 200
DT = 1.0 / 20.0  # ManiSkill's default control_freq


def rpy_matrix(r, p, y):
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def axis_angle(axis, th):
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * (K @ K)


def read_tree(path):
    """-> (link names in file order, joints [{name, type, parent, child, T (4,4), axis, lower, upper, vel}])."""
    root = ET.parse(path).getroot()
    links = [e.get("name") for e in root.findall("link")]
    joints = []
    for j in root.findall("joint"):
        o = j.find("origin")
        xyz = [float(v) for v in (o.get("xyz") if o is not None and o.get("xyz") else "0 0 0").split()]
        rpy = [float(v) for v in (o.get("rpy") if o is not None and o.get("rpy") else "0 0 0").split()]
        T = np.eye(4)
        T[:3, :3] = rpy_matrix(*rpy)
        T[:3, 3] = xyz
        ax = j.find("axis")
        lim = j.find("limit")
        joints.append(dict(
            name=j.get("name"), type=j.get("type"), parent=j.find("parent").get("link"),
            child=j.find("child").get("link"), T=T,
            axis=[float(v) for v in ax.get("xyz").split()] if ax is not None else [0.0, 0.0, 1.0],
            lower=float(lim.get("lower")) if lim is not None else 0.0,
            upper=float(lim.get("upper")) if lim is not None else 0.0,
            vel=float(lim.get("velocity")) if lim is not None else 0.0))
    return links, joints


def forward_kinematics(links, joints, q):
    """``q``: joint name -> angle (movable joints).  -> (L,4,4) world poses in ``links`` order (root = identity)."""
    by_child = {j["child"]: j for j in joints}
    pose = {}

    def of(link):
        if link in pose:
            return pose[link]
        j = by_child.get(link)
        if j is None:
            pose[link] = np.eye(4)
        else:
            M = of(j["parent"]) @ j["T"]
            if j["type"] in ("revolute", "continuous"):
                R = np.eye(4)
                R[:3, :3] = axis_angle(j["axis"], q[j["name"]])
                M = M @ R
            pose[link] = M
        return pose[link]

    return np.stack([of(n) for n in links])


def main(out=None):
    spec = importlib.util.spec_from_file_location("ref_constants", f"{REF}/gsworld/constants.py")
    consts = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(consts)

    links, joints = read_tree(URDF)
    movable = [j for j in joints if j["type"] != "fixed"]          # articulation order = file order: 6 arm + 6 gripper
    names = [j["name"] for j in movable]
    assert len(names) == len(consts.xarm_gs_qpos) == 12, names
    arm = names[:6]
    lo = np.array([j["lower"] for j in movable])
    hi = np.array([j["upper"] for j in movable])
    vmax = np.array([j["vel"] for j in movable]) * DT
    gain = np.where(np.arange(12) < 6, 1 - math.exp(-DT * 1e4 / 1e3), 1 - math.exp(-DT * 1e5 / 2e3))

    as_q = lambda v: dict(zip(names, v))  # noqa: E731
    scan = forward_kinematics(links, joints, as_q(consts.xarm_gs_qpos))
    rng = np.random.default_rng(0)
    q = np.array(consts.xarm_task_init_qpos, dtype=np.float64)
    qs, poses = [q.copy()], [forward_kinematics(links, joints, as_q(q))]
    for _ in range(STEPS):
        target = np.empty(12)
        target[:6] = rng.uniform(lo[:6], hi[:6])              # pd_joint_pos, normalize_action=False: Box(joint limits)
        target[6:] = rng.uniform(lo[6], hi[6])                # one mimic target for the six finger joints
        q = np.clip(q + np.clip(gain * (target - q), -vmax, vmax), lo, hi)
        qs.append(q.copy())
        poses.append(forward_kinematics(links, joints, as_q(q)))

    sem = consts.xarm_gs_semantics
    assert set(links) == set(sem), (links, sorted(sem))
    width = max(len(np.atleast_1d(sem[n])) for n in links)
    labels = np.full((len(links), width), -1, dtype=np.int64)
    for i, n in enumerate(links):
        v = np.atleast_1d(sem[n])
        labels[i, :len(v)] = v
    out = out or os.path.join(ROOT, "tests", "golden", "xarm6_rollout.npz")
    np.savez_compressed(
        out, link_names=np.array(links), labels=labels, joint_names=np.array(names), arm_joints=np.array(arm),
        qpos=np.array(qs, dtype=np.float32), qpos_scan=np.asarray(consts.xarm_gs_qpos, dtype=np.float32),
        link_scan=scan.astype(np.float32), link_now=np.array(poses, dtype=np.float32),
        sim2gs_arm=np.asarray(consts.sim2gs_xarm_trans, dtype=np.float32),
        link_offset=np.asarray(consts.object_offset["xarm_arm"], dtype=np.float32), control_dt=np.float32(DT))
    tcp = np.array(poses)[:, links.index("xarm_hand_tcp"), :3, 3]
    print("wrote", out, os.path.getsize(out), "bytes;", len(links), "links;",
          "tcp range", tcp.min(0).round(3), tcp.max(0).round(3))


if __name__ == "__main__":
    main()
