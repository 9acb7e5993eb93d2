"""GPU: the C ABI used from a plain C++ host program (examples/c_abi_demo.cpp, no PyTorch, no Python
in the compute path): hipMalloc'ed buffers, compute_rnnt_loss, the forward/backward split and the compacted node
order, on the upstream known-answer vector."""
import os
import subprocess

import numpy as np
import pytest

from cases import KAT1

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "c_abi_demo")


def test_c_host_program_reproduces_kat1():
    assert os.path.exists(EXE), "examples/c_abi_demo is built by __graft_entry__.build()"
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "rnnt_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stderr
    lines = {l.split()[0]: l.split()[1:] for l in out.stdout.strip().splitlines()}
    cost = float(lines["cost"][0])
    grads = np.array([float(v) for v in lines["grads"]]).reshape(KAT1["grads"].shape)
    assert abs(cost - KAT1["cost"]) <= 1e-5 * KAT1["cost"]
    assert np.abs(grads - KAT1["grads"]).max() <= 1e-6
    assert abs(float(lines["loss_mean"][0]) - KAT1["cost"]) <= 1e-5 * KAT1["cost"]
    grads_half = np.array([float(v) for v in lines["grads_half"]]).reshape(KAT1["grads"].shape)
    assert np.abs(grads_half - 0.5 * KAT1["grads"]).max() <= 1e-6
    assert lines["bad_blank_status"][0] == "2"
    # the ragged batch in the compacted node order: utterance 0 is KAT-1, found again through one row per node
    assert [int(v) for v in lines["compact_rowmap"]] == [0, 1, 2, 3, 4, 5, 6, 7]
    ccosts = [float(v) for v in lines["compact_costs"]]
    assert abs(ccosts[0] - KAT1["cost"]) <= 1e-5 * KAT1["cost"] and np.isfinite(ccosts[1]) and ccosts[1] > 0
    cgrads = np.array([float(v) for v in lines["compact_grads"]])
    assert cgrads.size == 8 * 5
    assert np.abs(cgrads[:30].reshape(KAT1["grads"].shape) - KAT1["grads"]).max() <= 1e-6
    assert np.abs(cgrads[30:].reshape(2, 5).sum(axis=1)).max() <= 1e-6       # every node's gradient row sums to zero
