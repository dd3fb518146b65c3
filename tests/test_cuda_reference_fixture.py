"""Compares the HIP path with fixtures dumped from the TRUE CUDA reference by tools/dump_reference.py.
No such fixture can be produced in the authoring container (parity unpinned): the pin test skips
until a maintainer with the upstream CUDA build drops tests/golden/cuda_reference_*.npz into the tree.  The checker
itself is exercised on every GPU run against a fixture of the SAME format written by the CPU oracle
(``--backend oracle``, labelled ``source = "oracle"``) -- that proves the dump / hash / compare pipeline works; it is
not a CUDA pin and the pin test refuses such a file.

Bars (north_star): RGB and inverse depth <= 1e-4 on every pixel that has no compositing decision inside an exp()-ulp
band of a threshold (``alpha < 1/255``, ``T (1 - alpha) < 1e-4``; the band is found by the oracle on the same inputs);
on the borderline pixels <= 2e-3; radii equal except for at most ``RADII_FLIPS`` Gaussians per frame --
``tools/fma_exposure.py`` measured 0 radius / rect flips between three FMA-contraction variants of the arithmetic on
configs[0] and configs[1] (profiles/fma_exposure.json), so more than a handful means a real difference."""
import glob
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import helpers as hp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "cuda_reference_*.npz")))
INPUT_KEYS = ("means3D", "shs", "opacities", "scales", "rotations", "viewmatrix", "projmatrix", "campos")
RADII_FLIPS = 3
BORDERLINE_TOL = 2e-3


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()


def check_against_fixture(ref, device):
    """-> report dict; raises AssertionError on a parity failure, pytest.fail on unusable inputs."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dump_reference as dr

    raw, cam = dr.scene_of(str(ref["config"]))
    inp = dr.inputs_of(raw, cam)
    drift = [k for k in INPUT_KEYS if _sha(inp[k]) != str(ref[f"sha256.{k}"])]
    if drift:
        # the two machines did not generate the same inputs (torch RNG / version drift): use the embedded arrays, or
        # say plainly that the comparison is impossible -- never compare outputs of different inputs
        missing = [k for k in drift if f"input.{k}" not in ref.files]
        if missing:
            pytest.fail(f"inputs regenerated here differ from the fixture's ({drift}; fixture torch "
                        f"{ref['torch_version']}) and {missing} are not embedded: re-dump with --embed-inputs")
        for k in drift:
            inp[k] = np.asarray(ref[f"input.{k}"])
            assert _sha(inp[k]) == str(ref[f"sha256.{k}"]), f"embedded input {k} does not match its own hash"
    st = hp.oracle_settings(cam)
    bg = np.zeros(3, np.float32)
    g = hp.gpu_forward(inp, st, bg, device=device)
    o = hp.oracle_forward(inp, st, bg)  # only for the borderline set of this frame
    border = o["borderline"] != 0
    flips = int((g["radii"] != ref["radii"]).sum())
    assert flips <= RADII_FLIPS, f"{flips} radii differ from the reference (bar: {RADII_FLIPS})"
    rep = dict(radii_flips=flips, borderline_pixels=int(border.sum()), input_drift=drift)
    for name, got, want in (("color", g["color"], ref["color"]), ("invdepth", g["invdepth"], ref["invdepth"])):
        d = np.abs(got - want).max(0)
        rep[f"{name}_max_off_borderline"] = float(d[~border].max())
        rep[f"{name}_max_borderline"] = float(d[border].max()) if border.any() else 0.0
        if flips == 0:
            assert d[~border].max() <= hp.RGB_TOL, f"{name}: {d[~border].max():.3g} off the borderline set"
        else:  # a flipped radius moves a rect: its pixels are excluded by count, not by position
            assert int((d[~border] > hp.RGB_TOL).sum()) <= 512 * flips, f"{name}: too many pixels beyond 1e-4"
        assert rep[f"{name}_max_borderline"] <= BORDERLINE_TOL, f"{name}: {rep[f'{name}_max_borderline']:.3g} on a borderline pixel"
    return rep


@pytest.mark.skipif(not FIXTURES, reason="no CUDA-reference fixture present (see tools/dump_reference.py)")
@pytest.mark.parametrize("path", FIXTURES or ["<none>"])
def test_matches_cuda_reference(cuda_device, path):
    ref = np.load(path)
    assert str(ref["source"]) == "cuda", "this file was written by --backend oracle: it is not a CUDA pin"
    print(check_against_fixture(ref, cuda_device))


def test_fixture_pipeline_on_an_oracle_written_file(cuda_device, tmp_path):
    """NOT a CUDA pin: the same dump format written by the CPU oracle, to keep the checker above alive."""
    path = str(tmp_path / "oracle_config1.npz")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "dump_reference.py"), "--out", path,
                           "--config", "config1", "--backend", "oracle"])
    ref = np.load(path)
    assert str(ref["source"]) == "oracle"
    rep = check_against_fixture(ref, cuda_device)
    assert rep["radii_flips"] == 0 and not rep["input_drift"]
    # a drifted RNG is detected and repaired from the embedded inputs, not compared blindly
    tampered = dict(ref)
    tampered["sha256.opacities"] = "0" * 64
    np.savez(str(tmp_path / "bad.npz"), **tampered)
    with pytest.raises(AssertionError, match="does not match its own hash"):
        check_against_fixture(np.load(str(tmp_path / "bad.npz")), cuda_device)
