"""RCCL under the multi-GPU layer on the one GPU a test box has (BASELINE.json configs[3]).  The N > 1 control
flow is covered on CPU by tests/test_distributed_cpu.py (gloo, world size 2); what those cannot show is RCCL itself executing
``gsworld_amd.distributed`` -- the communicator bound to the device, the collective on the side stream, its ordering against
frames rendered under hipGraph replay on other streams.  A process group of ONE rank over the "nccl" backend does (in a process
of its own: a communicator that hangs must not take the suite with it)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frame_gather_runs_over_rccl_in_a_group_of_one(cuda_device):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_world1.py"), "6", "200000"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and lines, p.stderr[-2000:]
    rec = json.loads(lines[-1])
    print(json.dumps(rec)[:600])
    assert rec["ok"] and rec["backend"] == "nccl" and rec["world_size"] == 1
    for c in ("gather", "all_gather"):
        assert rec[c]["frames_gathered"] == 48 and rec[c]["frames_that_differ_from_the_reference_frame"] == 0
        assert rec[c]["gather_ms"]["count"] == 6  # one collective per batch, timed by HIP events on the side stream
