"""A short slice of tools/fuzz_parity.py (random sizes / image shapes / scales / options / parameter spaces, HIP vs
oracle and variant-0 bit-identity) in the GPU suite; the tool itself runs hundreds of cases (400 clean at round 1)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_parity_slice(cuda_device):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "16", "123"], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "fuzz ok: 16 cases" in out.stdout


def test_random_forward_only_slice(cuda_device):
    """A slice of tools/fuzz_forward_only.py: inference frames vs default frames, bit for bit, on random sizes / shapes /
    options (700 cases clean in round 3; the sweep is what found that grids beyond the resident-quadrant compositor
    must keep per-tile lists)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_forward_only.py"), "40", "321"], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "40 cases x 2 frames bit-identical" in out.stdout


def test_random_batch_slice(cuda_device):
    """A slice of tools/fuzz_batch.py: B = 1 ... 19 frames through one gsr_forward_batch call against one gsr_forward call
    each, bit for bit, over random sizes / image shapes / cameras / options / layouts and four consecutive steps (exact,
    capacity path, kept splitters, blind splitters with the halved bucket count); 400 cases = 12 900 frames clean in round 5."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_batch.py"), "24", "77"], cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "batch fuzz: 24 cases" in out.stdout
