"""Compares the HIP path with fixtures dumped from the TRUE CUDA reference by tools/dump_reference.py.
No such fixture can be produced in the authoring container (parity unpinned): the test skips
until a maintainer with the upstream CUDA build drops tests/golden/cuda_reference_*.npz into the tree."""
import glob
import os

import numpy as np
import pytest

from gsworld_amd import scenes
from tests import helpers as hp

pytestmark = pytest.mark.gpu
FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "cuda_reference_*.npz")))


@pytest.mark.skipif(not FIXTURES, reason="no CUDA-reference fixture present (see tools/dump_reference.py)")
@pytest.mark.parametrize("path", FIXTURES or ["<none>"])
def test_matches_cuda_reference(cuda_device, path):
    ref = np.load(path)
    if str(ref["config"]) == "config1":
        raw, cam = scenes.random_scene_camera_frame(100_000, seed=0), scenes.identity_camera(256, 256, 60.0)
    else:
        raw, cam = scenes.tabletop_scene("xarm6_align"), scenes.sensor_camera("xarm6_align")
    inp = hp.np_inputs(raw, cam)
    g = hp.gpu_forward(inp, hp.oracle_settings(cam), np.zeros(3, np.float32))
    assert (g["radii"] != ref["radii"]).mean() < 1e-4          # FMA-contraction differences may flip a ceil()
    assert np.abs(g["color"] - ref["color"]).max() <= 2e-2      # a flipped radius changes single pixels
    assert np.median(np.abs(g["color"] - ref["color"])) <= 1e-6
    assert (np.abs(g["color"] - ref["color"]) > 1e-4).mean() < 1e-3
