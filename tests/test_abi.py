"""CPU tests: the C-ABI library loads and exports every symbol include/rnnt_mi355.h declares,
size/argument queries behave, and the Python operator mirrors upstream's argument checks.
No compute call is made (no GPU here)."""
import ctypes
import os
import re

import pytest
import torch

import rnnt_amd
from rnnt_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rnnt_mi355.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(?:rnntStatus_t|int|const char \*)\s*\*?\s*(\w+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    decl = declared_symbols()
    assert len(decl) >= 12
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/rnnt_mi355.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == decl  # the ctypes binding covers the whole header


def test_abi_version_and_status_strings():
    lib = _lib.load()
    assert lib.rnnt_mi355_abi_version() == _lib.ABI_VERSION
    assert lib.rnntGetStatusString(0) == b"no error"
    assert lib.rnntGetStatusString(2) == b"invalid value"
    assert lib.rnntGetStatusString(99) == b"unknown error"


def test_shipped_library_does_not_read_the_environment():
    """The RNNT_* tuning knobs exist only in the -DRNNT_TUNING developer build (make tuning)."""
    import subprocess

    syms = subprocess.check_output(["nm", "-D", "--undefined-only", _lib.LIB_PATH], text=True)
    assert "getenv" not in syms


def test_lattices_wider_than_4096_columns_are_a_value_error_before_any_launch():
    acts = torch.zeros(1, 2, 4098, 3)
    with pytest.raises(ValueError, match="at most 4095 labels"):
        rnnt_amd.rnnt_loss(acts, torch.ones(1, 4097, dtype=torch.int32), torch.full((1,), 2, dtype=torch.int32),
                           torch.full((1,), 4097, dtype=torch.int32))


def test_workspace_size():
    # 6 fp32 planes of B*(ND+1)*US (+2 rows slack) + ll[B] + 16-byte node spans [B+1]; config 2: ND=440, US=64
    n = _lib.workspace_bytes(400, 41, 32)
    plane = 32 * 441 * 64 + 2 * 64
    assert n == (6 * plane + 64) * 4 + 768
    assert n % 256 == 0
    lib = _lib.load()
    sz = ctypes.c_size_t(0)
    assert lib.rnnt_get_workspace_size(0, 41, 32, ctypes.byref(sz)) == 2       # invalid value
    assert lib.rnnt_get_workspace_size(400, 600, 32, ctypes.byref(sz)) == 0    # beyond 512 columns: the wide K2
    assert lib.rnnt_get_workspace_size(400, 4097, 32, ctypes.byref(sz)) == 2   # maxU > 4096 unsupported
    assert lib.get_workspace_size(400, 41, 32, 0, ctypes.byref(sz)) == 2       # gpu=false: no CPU path
    assert lib.get_workspace_size(400, 41, 32, 1, ctypes.byref(sz)) == 0 and sz.value == n
    assert lib.rnnt_get_workspace_size(400, 41, 32, None) == 2


def test_invalid_arguments_return_status_not_crash():
    lib = _lib.load()
    opts = _lib.RnntOptions(None, 0, 4, 3, _lib.DTYPE_F32, None)
    # null pointers are rejected before any HIP call is made
    assert lib.rnnt_forward(None, None, None, None, 5, 1, None, None, 1, None, opts) == 2
    assert lib.rnnt_backward(None, None, None, None, None, 5, 1, None, 0, ctypes.c_float(1.0), None, opts) == 2
    assert lib.compute_rnnt_loss(None, None, None, None, None, 5, 1, None, None, opts) == 2
    assert lib.rnnt_reduce_costs(None, None, 4, 1, None) == 2
    # the entry points of the compacted node order and of the joint's hidden layer likewise
    assert lib.rnnt_compact_rows(None, None, 1, 4, 3, 10, None, None, None) == 2
    assert lib.rnnt_forward_compact(None, 10, None, None, None, 5, 1, None, None, 1, None, opts) == 2
    assert lib.rnnt_backward_compact(None, None, 10, None, None, None, 5, 1, None, 0, ctypes.c_float(1.0), None, opts) == 2
    assert lib.rnnt_joint_hidden_forward(None, None, None, 1, 4, 3, 8, 0, None) == 2
    assert lib.rnnt_joint_hidden_backward(None, None, None, None, 1, 4, 3, 8, 0, None, None) == 2
    assert lib.rnnt_joint_hidden_forward_ragged(None, None, None, None, None, 1, 4, 3, 8, 0, None) == 2
    assert lib.rnnt_joint_hidden_backward_ragged(None, None, None, None, None, None, 1, 4, 3, 8, 0, None, None) == 2
    assert lib.rnnt_joint_hidden_forward_compact(None, None, None, 10, None, None, 1, 4, 3, 8, 0, None) == 2
    assert lib.rnnt_joint_hidden_backward_compact(None, None, 10, None, None, None, None, 1, 4, 3, 8, 0, None, None) == 2
    nbytes = ctypes.c_size_t(0)
    assert lib.rnnt_joint_hidden_workspace_size(0, 4, 3, 8, ctypes.byref(nbytes)) == 2
    assert lib.rnnt_joint_hidden_workspace_size(2, 33, 3, 8, ctypes.byref(nbytes)) == 0
    assert nbytes.value == 2 * 3 * 3 * 8 * 4          # B * ceil(T / 16) frame chunks * U1 * H fp32 partial sums
    with pytest.raises(RuntimeError, match="invalid value"):
        _lib.check(2, "x")


def _inputs(B=2, T=4, U=3, V=5):
    acts = torch.zeros(B, T, U + 1, V)
    labels = torch.ones(B, U, dtype=torch.int32)
    al = torch.full((B,), T, dtype=torch.int32)
    ll = torch.full((B,), U, dtype=torch.int32)
    return acts, labels, al, ll


def test_operator_surface_matches_upstream():
    m = rnnt_amd.RNNTLoss()  # the reference constructs it with no arguments (model.py:87)
    assert m.blank == 0 and m.reduction == "mean"
    assert len(m.state_dict()) == 0 and len(list(m.parameters())) == 0
    import warprnnt_pytorch  # the shim src/rnnt/model.py:5 resolves to

    assert warprnnt_pytorch.RNNTLoss is rnnt_amd.RNNTLoss
    with pytest.raises(ValueError):
        rnnt_amd.RNNTLoss(reduction="bogus")


def test_argument_checks_like_upstream():
    acts, labels, al, ll = _inputs()
    f = rnnt_amd.rnnt_loss
    with pytest.raises(TypeError, match="labels must be"):
        f(acts, labels.long(), al, ll)
    with pytest.raises(TypeError, match="lengths must be"):
        f(acts, labels, al.long(), ll)
    with pytest.raises(TypeError, match="label_lengths must be"):
        f(acts, labels, al, ll.long())
    with pytest.raises(TypeError):
        f(acts.double(), labels, al, ll)
    with pytest.raises(ValueError, match="contiguous"):
        f(acts.transpose(1, 2), labels, al, ll)
    with pytest.raises(ValueError, match="4D"):
        f(acts[0], labels, al, ll)
    with pytest.raises(ValueError, match="2D"):
        f(acts, labels[0], al, ll)
    with pytest.raises(ValueError, match="length per example"):
        f(acts, labels, al[:1], ll)
    with pytest.raises(ValueError, match="label length per example"):
        f(acts, labels, al, ll[:1])
    with pytest.raises(ValueError, match="Output length mismatch"):
        f(acts, labels[:, :2].contiguous(), al, ll)
    with pytest.raises(ValueError, match="reduction"):
        f(acts, labels, al, ll, reduction="avg")


def test_cpu_tensors_fail_loudly_no_fallback():
    acts, labels, al, ll = _inputs()
    with pytest.raises(RuntimeError, match="no CPU path"):
        rnnt_amd.RNNTLoss()(acts, labels, al, ll)


def test_length_checks_on_host():
    acts, labels, al, ll = _inputs()
    from rnnt_amd.loss import certify_inputs

    certify_inputs(acts, labels, al, ll)  # fits
    bad = al.clone()
    bad[0] = 5
    with pytest.raises(ValueError, match="Input length mismatch"):
        certify_inputs(acts, labels, bad, ll)
    bad = al.clone()
    bad[1] = 0
    with pytest.raises(ValueError, match="Input length mismatch"):
        certify_inputs(acts, labels, bad, ll)
    bad = ll.clone()
    bad[0] = 4
    with pytest.raises(ValueError, match="Output length mismatch"):
        certify_inputs(acts, labels, al, bad)
    # superset of upstream: shorter-than-tensor lengths are accepted (sharded ragged batches)
    certify_inputs(acts, labels, al - 1, ll - 1)


def test_missing_library_raises(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.RnntLibraryError, match="no CPU or PyTorch fallback"):
        _lib.load()


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "rnnt_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in text and "from oracle" not in text and "rnnt_oracle" not in text, fn


def test_compiled_torch_binding_loads_without_a_gpu():
    """librnnt_amd_torch.so registers torch.ops.rnnt_amd.* on import; nothing is computed here."""
    import torch

    from rnnt_amd import _torch_binding

    ops = _torch_binding.load()
    assert ops.abi_version() == _lib.ABI_VERSION
    for name in ("rnnt_loss", "rnnt_forward", "rnnt_backward"):
        assert hasattr(ops, name)
    a = torch.zeros(1, 2, 3, 5)
    lab = torch.ones(1, 2, dtype=torch.int32)
    one = torch.ones(1, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.rnnt_loss(a, lab, one, one, 0, 1)


def test_default_import_ignores_library_override_in_the_environment():
    """The product loads its own in-tree library and nothing else: RNNT_AMD_LIB (a developer switch until round 2)
    is not read; differently-built libraries go through tools/with_lib.py."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os; from rnnt_amd import _lib; "
            "assert _lib.LIB_PATH == os.path.join(os.path.dirname(_lib.__file__), 'lib', 'librnnt_mi355.so'), _lib.LIB_PATH")
    env = dict(os.environ, RNNT_AMD_LIB="/nonexistent/librnnt_mi355.so")
    subprocess.check_call([sys.executable, "-c", code], cwd=root, env=env)
    src = open(os.path.join(root, "rnnt_amd", "_lib.py")).read()
    assert "os.environ" not in src
