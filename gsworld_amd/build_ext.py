"""Builds ``gsworld_amd/_C_ext*.so`` -- the compiled torch binding of libgsr_hip.so (csrc_torch/ext.cpp; SURVEY.md 8b
row B3: the ``_C`` module upstream's ``diff_gaussian_rasterization/__init__.py`` imports) -- in-tree, with g++ against
the installed torch headers.  Host code only; the HIP kernels live in libgsr_hip.so (gsworld_amd/csrc/Makefile).

    python -m gsworld_amd.build_ext
"""
from __future__ import annotations

import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
NAME = "_C_ext"


def target() -> str:
    return os.path.join(HERE, NAME + sysconfig.get_config_var("EXT_SUFFIX"))


def build(force: bool = False) -> str:
    import pybind11  # noqa: F401  (headers ship with torch as well)
    import torch
    from torch.utils import cpp_extension as ce

    src = os.path.join(HERE, "csrc_torch", "ext.cpp")
    out = target()
    deps = [src, os.path.join(os.path.dirname(HERE), "include", "gsr.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    rocm = os.environ.get("ROCM_HOME", "/opt/rocm")
    inc = ce.include_paths() + [sysconfig.get_paths()["include"], os.path.join(rocm, "include")]
    libdir = ce.library_paths()[0]
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", out,
           f"-DTORCH_EXTENSION_NAME={NAME}", "-DTORCH_API_INCLUDE_EXTENSION_H", "-D__HIP_PLATFORM_AMD__=1",
           "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           "-Wno-deprecated-declarations"]
    cmd += [f"-I{p}" for p in inc]
    cmd += [f"-L{libdir}", "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python", "-lc10_hip", "-ltorch_hip",
            f"-L{HERE}", "-l:libgsr_hip.so", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,$ORIGIN"]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
