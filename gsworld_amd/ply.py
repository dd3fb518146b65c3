"""Binary little-endian PLY reader / writer for 3DGS models with GSWorld's optional ``semantics`` column
 -- no ``plyfile`` dependency.

Layout restated from /root/reference/gsworld/mani_skill/utils/wrappers/semantic_3dgs_wrapper.py:75-167 and
/root/reference/gsworld/utils/pcd_utils.py:33-129: one ``vertex`` element of float32 properties in the order
``x y z nx ny nz f_dc_0..2 f_rest_0..44 opacity scale_0..2 rot_0..3 [semantics]``; ``f_rest`` is stored
channel-major ``(3, 15)`` and transposed to ``(15, 3)`` on load; tensors come back shaped as ``load_ply`` leaves
them: ``_features_dc (N,1,3)``, ``_features_rest (N,15,3)``, ``_opacity (N,1,1)`` (the reference's double
unsqueeze, ``:117`` + ``:154``), ``_semantics (N,1)``.
"""
from __future__ import annotations

import numpy as np
import torch


def read_ply(path: str) -> dict:
    """-> {property name: float32 array (N,)} for the vertex element of a binary_little_endian PLY."""
    with open(path, "rb") as f:
        header = []
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            header.append(line.decode("ascii", "replace").strip())
            if header[-1] == "end_header":
                break
        if header[0] != "ply" or not any(h.startswith("format binary_little_endian") for h in header):
            raise ValueError(f"{path}: only binary_little_endian PLY files are supported")
        count, props, in_vertex = 0, [], False
        types = {"float": "<f4", "float32": "<f4", "double": "<f8", "uchar": "u1", "uint8": "u1", "int": "<i4",
                 "int32": "<i4", "uint": "<u4", "short": "<i2", "ushort": "<u2", "char": "i1"}
        for h in header:
            tok = h.split()
            if tok[:1] == ["element"]:
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    count = int(tok[2])
            elif tok[:1] == ["property"] and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties in the vertex element are not supported")
                props.append((tok[2], types[tok[1]]))
        data = np.fromfile(f, dtype=np.dtype(props), count=count)
    return {name: np.ascontiguousarray(data[name]).astype(np.float32) for name, _ in props}


def write_ply(path: str, columns: dict) -> None:
    names = list(columns.keys())
    n = len(next(iter(columns.values())))
    rec = np.empty(n, dtype=np.dtype([(k, "<f4") for k in names]))
    for k in names:
        rec[k] = np.asarray(columns[k], dtype=np.float32).reshape(n)
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\n")
        f.write(f"element vertex {n}\n".encode())
        for k in names:
            f.write(f"property float {k}\n".encode())
        f.write(b"end_header\n")
        rec.tofile(f)


def _sorted(cols: dict, prefix: str):
    names = sorted((k for k in cols if k.startswith(prefix)), key=lambda s: int(s.split("_")[-1]))
    return np.stack([cols[k] for k in names], axis=1) if names else np.zeros((len(cols["x"]), 0), np.float32)


def gaussian_tensors(cols: dict, max_sh_degree: int = 3) -> dict:
    """PLY columns -> numpy arrays in the shapes ``load_ply`` produces."""
    n = len(cols["x"])
    xyz = np.stack((cols["x"], cols["y"], cols["z"]), axis=1)
    dc = np.stack((cols["f_dc_0"], cols["f_dc_1"], cols["f_dc_2"]), axis=1)[:, None, :]  # (N,1,3)
    rest = _sorted(cols, "f_rest_")
    want = 3 * (max_sh_degree + 1) ** 2 - 3
    assert rest.shape[1] == want, f"expected {want} f_rest columns, found {rest.shape[1]}"
    rest = rest.reshape(n, 3, want // 3).transpose(0, 2, 1)  # channel-major (3,15) -> (15,3)
    sem = cols["semantics"][:, None] if "semantics" in cols else np.zeros((n, 1), np.float32)
    return dict(xyz=xyz, features_dc=np.ascontiguousarray(dc), features_rest=np.ascontiguousarray(rest),
                opacity=cols["opacity"][:, None, None], scaling=_sorted(cols, "scale_"), rotation=_sorted(cols, "rot_"),
                semantics=sem)


def read_gaussian_ply(path: str, model, device="cuda", upstream: bool = False) -> None:
    """Fills a GaussianModel-like object (``_xyz``, ``_features_dc`` ... ) from ``path``.

    Default: what GSWorld's ``Semantic3DGSWrapper.load_ply`` leaves behind
    (semantic_3dgs_wrapper.py:151-167) -- frozen plain tensors, ``_opacity (N,1,1)``, ``_semantics (N,1)``.
    ``upstream=True``: what the stock 3DGS ``GaussianModel.load_ply`` leaves behind -- ``nn.Parameter`` s that require
    grad (training resumes from a saved PLY), ``_opacity (N,1)``, no semantics."""
    t = gaussian_tensors(read_ply(path), getattr(model, "max_sh_degree", 3))
    dev = device if (device != "cuda" or torch.cuda.is_available()) else "cpu"
    for attr, key in (("_xyz", "xyz"), ("_features_dc", "features_dc"), ("_features_rest", "features_rest"),
                      ("_opacity", "opacity"), ("_scaling", "scaling"), ("_rotation", "rotation"),
                      ("_semantics", "semantics")):
        if upstream and key == "semantics":
            continue
        v = torch.tensor(t[key], dtype=torch.float32, device=dev)
        if upstream:
            if key == "opacity":
                v = v.reshape(-1, 1)
            v = torch.nn.Parameter(v.contiguous().requires_grad_(True))
        setattr(model, attr, v)


def write_gaussian_ply(path: str, model, with_semantics: bool | None = None) -> None:
    g = lambda a: getattr(model, a).detach().cpu().numpy()  # noqa: E731
    xyz = g("_xyz")
    n = xyz.shape[0]
    cols = {"x": xyz[:, 0], "y": xyz[:, 1], "z": xyz[:, 2], "nx": np.zeros(n), "ny": np.zeros(n), "nz": np.zeros(n)}
    dc = g("_features_dc").reshape(n, -1, 3).transpose(0, 2, 1).reshape(n, -1)
    rest = g("_features_rest").reshape(n, -1, 3).transpose(0, 2, 1).reshape(n, -1)  # back to channel-major
    for i in range(dc.shape[1]):
        cols[f"f_dc_{i}"] = dc[:, i]
    for i in range(rest.shape[1]):
        cols[f"f_rest_{i}"] = rest[:, i]
    cols["opacity"] = g("_opacity").reshape(n)
    for i, c in enumerate(g("_scaling").T):
        cols[f"scale_{i}"] = c
    for i, c in enumerate(g("_rotation").T):
        cols[f"rot_{i}"] = c
    sem = getattr(model, "_semantics", None)
    if with_semantics or (with_semantics is None and sem is not None and sem.numel() == n):
        cols["semantics"] = sem.detach().cpu().numpy().reshape(n)
    write_ply(path, cols)
