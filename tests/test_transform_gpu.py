"""GPU parity of the fused per-step rigid transform: directly against the outputs of the reference's
own transform_gaussians (tests/golden/transform_gaussians.npz, captured by tools/make_golden.py from the imported
reference), and against the golden-pinned CPU restatement (oracle/transform_ref.py) applied the way GSWorldWrapper
applies it: per part, isin() mask -> transform -> masked write-back."""
import os
import types

import numpy as np
import pytest
import torch

from gsworld_amd import transform as tf
from oracle import transform_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

pytestmark = pytest.mark.gpu


def _rand_rigid(gen, k):
    q = torch.randn(k, 4, generator=gen)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1).reshape(k, 3, 3)
    M = torch.eye(4).repeat(k, 1, 1)
    M[:, :3, :3] = R
    M[:, :3, 3] = torch.randn(k, 3, generator=gen)
    return M


def test_fused_transform_matches_wrapper_semantics(cuda_device):
    gen = torch.Generator().manual_seed(0)
    N = 50_000
    parts = {f"link{k}": ([10 + k] if k % 3 else [10 + k, 100 + k]) for k in range(16)}  # some links own 2 labels
    parts["005_tomato_soup_can"] = 110
    parts["dtc_green_can"] = 201
    K = len(parts)
    labels = torch.randint(0, 260, (N, 1), generator=gen).float()
    g = types.SimpleNamespace(_xyz=torch.randn(N, 3, generator=gen), _scaling=torch.randn(N, 3, generator=gen),
                              _rotation=torch.randn(N, 4, generator=gen) * 1.7, _opacity=torch.randn(N, 1, generator=gen),
                              _semantics=labels)
    M = _rand_rigid(gen, K)
    scales = torch.ones(K)
    scales[-2:] = torch.tensor([1.07, 0.93])  # tracked actors carry a uniform scale (gs_world_wrapper.py:150-153)

    # reference semantics on the CPU, per part, as the wrapper does at num_envs = 1
    xyz_ref, rot_ref = g._xyz.clone(), g._rotation.clone()
    for k, (name, labs) in enumerate(parts.items()):
        target = torch.tensor(labs if isinstance(labs, list) else [labs])
        mask = torch.isin(g._semantics.long().squeeze(-1), target)
        idx = torch.where(mask)[0]
        sc = None if scales[k] == 1 else scales[k]
        x, _, r, _ = transform_ref.transform_gaussians(g, idx, scale=sc, rot_mat=M[k:k + 1, :3, :3], translation=M[k:k + 1, :3, 3])
        xyz_ref[mask] = x[0] if x.dim() == 3 else x
        rot_ref[mask] = r[0]

    dev = cuda_device
    op = tf.FusedPartTransform(parts, labels.to(dev))
    xyz, rot = op.apply(g._xyz.to(dev), g._rotation.to(dev), M, scales)
    np.testing.assert_allclose(xyz.cpu().numpy(), xyz_ref.numpy(), atol=5e-6, rtol=1e-6)
    np.testing.assert_allclose(rot.cpu().numpy(), rot_ref.numpy(), atol=5e-6, rtol=1e-6)
    touched = torch.isin(labels.long().squeeze(-1), torch.tensor([l for v in parts.values() for l in (v if isinstance(v, list) else [v])]))
    # untouched labels are copied through bit-exactly
    assert torch.equal(xyz.cpu()[~touched], g._xyz[~touched]) and torch.equal(rot.cpu()[~touched], g._rotation[~touched])
    assert 0.05 < touched.float().mean() < 0.2


def test_fused_transform_errors(cuda_device):
    with pytest.raises(ValueError, match="outside the LUT"):
        tf.FusedPartTransform({"a": 5000}, torch.zeros(4, 1, device=cuda_device))
    op = tf.FusedPartTransform({"a": 1, "b": 2}, torch.zeros(4, 1, device=cuda_device))
    with pytest.raises(ValueError, match="expected 2 matrices"):
        op.pack(torch.eye(4)[None])
    with pytest.raises(RuntimeError, match="no CPU path"):
        op.apply(torch.zeros(4, 3), torch.zeros(4, 4), torch.eye(4).repeat(2, 1, 1))


def _pack_ieee(M, scales):
    """matrix_to_quaternion + table layout in strict IEEE float32 scalar arithmetic (numpy), the order of
    gsworld_amd/transform.py.  torch's vectorised CPU kernels are NOT a bit-level reference: the same call differs in
    the last ulp between an AVX2 and an AVX-512 host (seen between the build container and the MI355X box)."""
    f = np.float32
    out = np.zeros((M.shape[0], 17), np.float32)
    for k, m in enumerate(M.numpy().astype(np.float32)):
        m00, m01, m02, m10, m11, m12, m20, m21, m22 = [f(x) for x in m[:3, :3].reshape(-1)]
        d = [f(1) + m00 + m11 + m22, f(1) + m00 - m11 - m22, f(1) - m00 + m11 - m22, f(1) - m00 - m11 + m22]
        qa = [np.sqrt(x) if x > 0 else f(0) for x in d]
        best = int(np.argmax(qa))
        c = [[qa[0] * qa[0], m21 - m12, m02 - m20, m10 - m01], [m21 - m12, qa[1] * qa[1], m10 + m01, m02 + m20],
             [m02 - m20, m10 + m01, qa[2] * qa[2], m12 + m21], [m10 - m01, m20 + m02, m21 + m12, qa[3] * qa[3]]]
        den = f(2) * max(qa[best], f(0.1))
        q = np.array([x / den for x in c[best]], np.float32)
        if q[0] < 0:
            q = -q
        out[k, :9] = m[:3, :3].reshape(-1)
        out[k, 9:12] = m[:3, 3]
        out[k, 12] = scales[k]
        out[k, 13:] = q
    return out


def test_device_side_table_matches_the_host_pack(cuda_device):
    """gsr_pack_part_transforms (matrix -> quaternion on the device, for poses that already live on the GPU): every
    float equal to the IEEE restatement of FusedPartTransform.pack, including the branch points of the quaternion
    extraction (identity, half turns about each axis, tiny angles); within an ulp of torch's own CPU result."""
    gen = torch.Generator().manual_seed(3)
    special = []
    for diag in ((1, 1, 1), (1, -1, -1), (-1, 1, -1), (-1, -1, 1)):  # identity and the three half turns
        M = torch.eye(4)
        M[0, 0], M[1, 1], M[2, 2] = diag
        special.append(M)
    tiny = torch.eye(4)
    tiny[0, 1], tiny[1, 0] = -1e-7, 1e-7
    special.append(tiny)
    M = torch.cat((torch.stack(special), _rand_rigid(gen, 59)))
    K = M.shape[0]
    scales = torch.rand(K, generator=gen) + 0.5
    op = tf.FusedPartTransform({f"p{k}": k for k in range(K)}, torch.zeros(8, device=cuda_device))
    got = op.pack_on_device(M.to(cuda_device).contiguous(), scales.to(cuda_device)).cpu()
    want = torch.from_numpy(_pack_ieee(M, scales.numpy()))
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    assert float((got - op.pack(M, scales)).abs().max()) <= 2.5e-7
    got1 = op.pack_on_device(M.to(cuda_device).contiguous()).cpu()  # scales default to 1
    assert torch.equal(got1[:, 12], torch.ones(K)) and torch.equal(got1[:, 13:], want[:, 13:])


def test_apply_with_device_matrices_needs_no_host_copy(cuda_device):
    gen = torch.Generator().manual_seed(4)
    N, K = 20_000, 18
    labels = torch.randint(0, 30, (N,), generator=gen).float().to(cuda_device)
    xyz = torch.randn(N, 3, generator=gen).to(cuda_device)
    rot = (torch.randn(N, 4, generator=gen) * 1.3).to(cuda_device)
    M, scales = _rand_rigid(gen, K), torch.rand(K, generator=gen) + 0.5
    op_h = tf.FusedPartTransform({f"p{k}": k + 1 for k in range(K)}, labels)
    op_d = tf.FusedPartTransform({f"p{k}": k + 1 for k in range(K)}, labels)
    xh, rh = op_h.apply(xyz, rot, M, scales)
    xd, rd = op_d.apply(xyz, rot, M.to(cuda_device), scales.to(cuda_device))
    # (host torch pack vs device pack may differ in the last ulp of the quaternion, see _pack_ieee)
    assert float((xh - xd).abs().max()) == 0.0 and float((rh - rd).abs().max()) <= 1e-6
    with pytest.raises(ValueError):
        op_d.pack_on_device(M[:3].to(cuda_device))


@pytest.mark.parametrize("case", ["link_env1", "link_env3", "actor_env1", "actor_env2"])
def test_fused_transform_reproduces_the_reference_outputs(cuda_device, case):
    """HIP operator vs the vectors the REFERENCE function produced (no project-side mirror in between): the selected
    Gaussians form one part, the golden rot_mat / translation / scale are its pose; environment e of the batched call
    must equal slice e of the reference's (B,N,.) outputs, and unselected Gaussians pass through bit-exactly."""
    ref = np.load(os.path.join(GOLD, "transform_gaussians.npz"))
    dev = cuda_device
    xyz, rot, scaling = (torch.from_numpy(ref[k]) for k in ("xyz", "rotation", "scaling"))
    sel = torch.from_numpy(ref["selected"])
    N = xyz.shape[0]
    labels = torch.zeros(N, 1)
    labels[sel] = 7.0
    R = torch.from_numpy(ref[f"{case}.in.rot_mat"])
    t = torch.from_numpy(ref[f"{case}.in.translation"])
    E = R.shape[0]
    M = torch.eye(4).repeat(E, 1, 1, 1)  # (E, K=1, 4, 4)
    M[:, 0, :3, :3] = R
    M[:, 0, :3, 3] = t
    scale = torch.from_numpy(ref[f"{case}.in.scale"]) if f"{case}.in.scale" in ref.files else None
    per_env_scale = scale is not None and scale.dim() == 1  # the reference rewrites `scaling` only for a scale VECTOR
    scales = None if scale is None else scale.reshape(-1, 1).expand(E, 1).contiguous()
    op = tf.FusedPartTransform({"part": 7}, labels.to(dev), scaled_parts=("part",) if per_env_scale else ())
    for on_device in (False, True):
        Min = M.to(dev) if on_device else M
        sc_in = None if scales is None else (scales.to(dev) if on_device else scales)
        x, r, s = op.apply(xyz.to(dev), rot.to(dev), Min, sc_in, scaling=scaling.to(dev))
        assert tuple(x.shape) == (E, N, 3) and tuple(r.shape) == (E, N, 4) and tuple(s.shape) == (E, N, 3)
        want_x, want_r, want_s = (ref[f"{case}.out.{k}"] for k in ("xyz", "rotation", "scaling"))
        np.testing.assert_allclose(x[:, sel].cpu().numpy(), want_x.reshape(E, -1, 3), atol=2e-6, rtol=1e-6)
        np.testing.assert_allclose(r[:, sel].cpu().numpy(), want_r.reshape(E, -1, 4), atol=2e-6, rtol=1e-6)
        if per_env_scale:  # (B,n,3): what the wrapper writes back
            np.testing.assert_allclose(s[:, sel].cpu().numpy(), want_s, atol=3e-6, rtol=1e-6)
        else:  # scale None / 0-dim: the wrapper's shape test fails, the parameter stays as loaded
            assert torch.equal(s[:, sel].cpu(), scaling[sel].expand(E, -1, 3))
        keep = torch.ones(N, dtype=torch.bool)
        keep[sel] = False
        for e in range(E):
            assert torch.equal(x[e, keep].cpu(), xyz[keep]) and torch.equal(r[e, keep].cpu(), rot[keep])
            assert torch.equal(s[e, keep].cpu(), scaling[keep])
    # the single-environment entry point returns un-batched buffers for (K,4,4) matrices
    x1, r1 = op.apply(xyz.to(dev), rot.to(dev), M[0], None if scales is None else scales[0])
    assert tuple(x1.shape) == (N, 3) and tuple(r1.shape) == (N, 4)
    np.testing.assert_allclose(x1[sel].cpu().numpy(), ref[f"{case}.out.xyz"].reshape(E, -1, 3)[0], atol=2e-6, rtol=1e-6)
