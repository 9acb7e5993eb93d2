"""GPU: the C ABI used from a plain C++ host program (examples/c_abi_demo.cpp, no PyTorch, no Python
in the compute path): hipMalloc'ed buffers, compute_rnnt_loss and the forward/backward split, on the
upstream known-answer vector."""
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
