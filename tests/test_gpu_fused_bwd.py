"""GPU parity for the fused BACKWARD of the output projection + loss (backward half):
rnnt_joint_loss_backward through the C ABI.

The operator is K3 (the loss's gradient w.r.t. the logits) as the prologue of the projection's two backward
products; the gradient tensor itself is never written.  It is checked in two steps:

* against the unfused chain ON THE SAME LOGITS AND WORKSPACE: rnnt_backward (K3, itself held to the C loss oracle
  by tests/test_gpu_parity.py and test_gpu_fused.py) writes g, and grad_hidden = g W, grad_weight = g^T hidden,
  grad_bias = sum g are formed in float64 (/root/reference/src/rnnt/model.py:17,59 under autograd).  The fused
  kernels form the same rounded g tile by tile, so what may differ is fp32 accumulation order and the final
  rounding: every element within 0.55 spacings of the dtype at the value + the fp32 accumulation allowance;
* through the Python operator against the C loss oracle's gradient and the float64 chain rule
  (tests/test_gpu_fused.py::test_fused_operator_forward_backward runs on this backward since round 3).

The loss oracle is PARITY UNPINNED (oracle/__init__.py).
"""
import ctypes

import numpy as np
import pytest
import torch

from test_gpu_fused import DEV, EPS, forward_c_abi, make_inputs

pytestmark = pytest.mark.gpu


def backward_c_abi(hidden, weight, logits, labels, al, ll, ws, go, stride, scale, blank=0, want=(True, True, True),
                   ld=None):
    """rnnt_joint_loss_backward as a C caller would use it; returns (status, grad_hidden, grad_weight, grad_bias).
    ``ld``: rnnt_joint_loss_backward_strided on logits whose rows are ``ld`` elements apart."""
    from rnnt_amd import _lib
    from rnnt_amd.loss import _DTYPES

    lib = _lib.load()
    B, T, U1, H = hidden.shape
    V = weight.shape[0]
    h, w = hidden.to(DEV).contiguous(), weight.to(DEV).contiguous()
    lab, ald, lld = (torch.from_numpy(x).to(DEV) for x in (labels, al, ll))
    n = ctypes.c_size_t(0)
    st = lib.rnnt_joint_loss_backward_workspace_size(H, V, B, T, U1, ctypes.byref(n))
    if st != 0:
        return st, None, None, None
    bws = torch.full((int(n.value),), 0xFF, dtype=torch.uint8, device=DEV)     # poisoned scratch
    dh = torch.full_like(h, float("nan")) if want[0] else None
    dw = torch.full_like(w, float("nan")) if want[1] else None
    db = torch.full((V,), float("nan"), dtype=torch.float32, device=DEV) if want[2] else None
    god = torch.as_tensor(go, dtype=torch.float32, device=DEV).reshape(-1)
    opts = _lib.RnntOptions(torch.cuda.current_stream().cuda_stream, blank, T, U1, _DTYPES[hidden.dtype], None)
    outs = (dh.data_ptr() if want[0] else None, dw.data_ptr() if want[1] else None, db.data_ptr() if want[2] else None,
            ws.data_ptr(), bws.data_ptr(), opts)
    if ld is None:
        st = lib.rnnt_joint_loss_backward(h.data_ptr(), w.data_ptr(), logits.data_ptr(), lab.data_ptr(), ald.data_ptr(),
                                          lld.data_ptr(), H, V, B, god.data_ptr(), stride, ctypes.c_float(scale), *outs)
    else:
        st = lib.rnnt_joint_loss_backward_strided(h.data_ptr(), w.data_ptr(), logits.data_ptr(), ld, lab.data_ptr(),
                                                  ald.data_ptr(), lld.data_ptr(), H, V, B, god.data_ptr(), stride,
                                                  ctypes.c_float(scale), *outs)
    torch.cuda.synchronize()
    return st, dh, dw, db


def k3_c_abi(logits, labels, al, ll, ws, go, stride, scale, blank=0):
    """rnnt_backward on the same logits and workspace: the gradient tensor the fused backward never writes."""
    from rnnt_amd import _lib
    from rnnt_amd.loss import _DTYPES

    lib = _lib.load()
    B, T, U1, V = logits.shape
    lab, ald, lld = (torch.from_numpy(x).to(DEV) for x in (labels, al, ll))
    g = torch.full_like(logits, float("nan"))
    god = torch.as_tensor(go, dtype=torch.float32, device=DEV).reshape(-1)
    opts = _lib.RnntOptions(torch.cuda.current_stream().cuda_stream, blank, T, U1, _DTYPES[logits.dtype], None)
    st = lib.rnnt_backward(logits.data_ptr(), g.data_ptr(), lab.data_ptr(), ald.data_ptr(), lld.data_ptr(), V, B,
                           god.data_ptr(), stride, ctypes.c_float(scale), ws.data_ptr(), opts)
    torch.cuda.synchronize()
    assert st == 0
    return g


def spacings_off(got, ref64, allowance, dtype):
    """|got - ref| in units of the dtype's spacing at the value, after the fp32 accumulation allowance."""
    # (below the smallest normal number the spacing stays that of the smallest normal: fp16 2^-14, bf16 2^-126)
    mag = torch.maximum(ref64.abs(), got.double().abs()).clamp_min(2.0 ** -14 if dtype is torch.float16 else 2.0 ** -126)
    spacing = torch.exp2(torch.floor(torch.log2(mag))) * (2 * EPS[dtype])
    return float((((got.double() - ref64).abs() - allowance).clamp_min(0) / spacing).max())


def check_against_k3(hidden, weight, g, dh, dw, db):
    dtype = hidden.dtype
    V, H = weight.shape
    g64 = g.double().reshape(-1, V)
    h64 = hidden.to(DEV).double().reshape(-1, H)
    w64 = weight.to(DEV).double()
    assert bool(torch.isfinite(g64).all())
    if dh is not None:
        ref = (g64 @ w64).reshape(hidden.shape)
        allow = 2e-6 * (g64.abs() @ w64.abs()).reshape(hidden.shape)
        assert bool(torch.isfinite(dh.float()).all()), "an element of grad_hidden was not written"
        assert spacings_off(dh, ref, allow, dtype) <= 0.55
    if dw is not None:
        ref = g64.t() @ h64
        allow = 1e-5 * (g64.abs().t() @ h64.abs())     # up to 10^5 fp32 additions per element
        assert bool(torch.isfinite(dw.float()).all()), "an element of grad_weight was not written"
        assert spacings_off(dw, ref, allow, dtype) <= 0.55
    if db is not None:
        # the kernel sums the gradient tile before it is rounded; the reference sums the rounded tensor
        ref = g64.sum(0)
        # (one rounding step per element; fp16 gradients below 2^-14 are rounded to multiples of 2^-24)
        allow = 2 * EPS[dtype] * g64.abs().sum(0) + g64.shape[0] * (2.0 ** -25 if dtype is torch.float16 else 0.0) + 1e-7
        assert bool(torch.isfinite(db).all())
        assert float(((db.double() - ref).abs() - allow).max()) <= 0.0


CASES = [
    # B, T, U, V, dtype, blank, what
    (2, 9, 4, 4232, torch.bfloat16, 0, "the reference's vocabulary; one row tile, partial"),
    (3, 33, 7, 4240, torch.bfloat16, 0, "the last tile and the last block overlap their predecessors by 16 / 112 entries"),
    (2, 21, 5, 5456, torch.float16, 0, "fp16; 171 tiles"),
    (1, 40, 6, 128, torch.bfloat16, 0, "exactly one vocabulary block, two tiles"),
    (2, 17, 3, 200, torch.bfloat16, 0, "the last block and the last tile overlap their predecessors"),
    (4, 70, 12, 1024, torch.float16, 7, "3 640 nodes: 29 row tiles, 57 steps in several node ranges; blank = 7"),
    (1, 1, 0, 320, torch.bfloat16, 0, "a single node, no labels"),
    (5, 129, 20, 520, torch.float16, 3, "13 545 nodes, ragged; blank inside a chunk"),
]


@pytest.mark.parametrize("B,T,U,V,dtype,blank,what", CASES)
def test_fused_backward_equals_k3_then_two_products(B, T, U, V, dtype, blank, what):
    H = 512
    hidden, weight, bias, labels, al, ll = make_inputs(B, T, U, V, H, dtype, seed=V + T)
    if blank:
        labels = np.where(labels == blank, 0, labels).astype(np.int32)       # labels never equal the blank
    st, logits, costs, ws = forward_c_abi(hidden, weight, bias, labels, al, ll) if blank == 0 else \
        _forward_blank(hidden, weight, bias, labels, al, ll, blank)
    assert st == 0, what
    # padding nodes of the stored logits: whatever they hold must not matter
    for b in range(B):
        logits[b, al[b]:] = float("nan")
        logits[b, :, ll[b] + 1:] = float("inf")
    go = np.linspace(0.25, 1.5, B).astype(np.float32)
    g = k3_c_abi(logits, labels, al, ll, ws, go, 1, 1.0, blank)
    st, dh, dw, db = backward_c_abi(hidden, weight, logits, labels, al, ll, ws, go, 1, 1.0, blank)
    assert st == 0, what
    check_against_k3(hidden, weight, g, dh, dw, db)
    # scalar upstream gradient with the 1/B of 'mean' folded in, and only one output wanted
    g = k3_c_abi(logits, labels, al, ll, ws, [0.5], 0, 1.0 / B, blank)
    st, dh2, _, _ = backward_c_abi(hidden, weight, logits, labels, al, ll, ws, [0.5], 0, 1.0 / B, blank,
                                   want=(True, False, False))
    assert st == 0
    check_against_k3(hidden, weight, g, dh2, None, None)
    st, _, dw2, db2 = backward_c_abi(hidden, weight, logits, labels, al, ll, ws, [0.5], 0, 1.0 / B, blank,
                                     want=(False, True, True))
    assert st == 0
    check_against_k3(hidden, weight, g, None, dw2, db2)


STRIDED = [
    # B, T, U, V, ld, dtype, what
    (2, 9, 4, 4232, 4352, torch.bfloat16, "what rnnt_amd.FusedJointLoss allocates for the reference's vocabulary"),
    (3, 33, 7, 4234, 4352, torch.bfloat16, "the reference's 4234: no multiple of 8; whole tiles, masked from V on"),
    (2, 21, 5, 5455, 5504, torch.float16, "config/magic_*.yaml's 5455, odd; fp16"),
    (2, 17, 3, 130, 256, torch.bfloat16, "two vocabulary blocks, the second holds two entries"),
    (4, 70, 12, 1000, 1024, torch.float16, "several row tiles and node ranges; the last chunk straddles V"),
    (3, 33, 7, 4240, 4248, torch.bfloat16, "a stride without whole-tile padding: overlapping last tile and block"),
    (1, 40, 6, 128, 128, torch.bfloat16, "stride = V through the strided entry"),
]


@pytest.mark.parametrize("B,T,U,V,ld,dtype,what", STRIDED)
def test_fused_backward_on_padded_rows(B, T, U, V, ld, dtype, what):
    """rnnt_joint_loss_backward_strided on the logits rnnt_joint_loss_forward_strided stored (padding: NaN) against
    rnnt_backward on a contiguous copy of the same logits + the float64 products."""
    H = 512
    hidden, weight, bias, labels, al, ll = make_inputs(B, T, U, V, H, dtype, seed=V + T + 1)
    st, logits, costs, ws = forward_c_abi(hidden, weight, bias, labels, al, ll, ld=ld)
    assert st == 0, what
    for b in range(B):                                   # padding nodes, too
        logits[b, al[b]:] = float("nan")
        logits[b, :, ll[b] + 1:] = float("inf")
    dense = logits[..., :V].contiguous()
    go = np.linspace(0.25, 1.5, B).astype(np.float32)
    g = k3_c_abi(dense, labels, al, ll, ws, go, 1, 1.0)
    st, dh, dw, db = backward_c_abi(hidden, weight, logits, labels, al, ll, ws, go, 1, 1.0, ld=ld)
    assert st == 0, what
    check_against_k3(hidden, weight, g, dh, dw, db)
    if ld == V or V % 8 == 0:
        # the contiguous entry point on the contiguous copy: the same bits
        st, dh2, dw2, db2 = backward_c_abi(hidden, weight, dense, labels, al, ll, ws, go, 1, 1.0)
        assert st == 0
        assert torch.equal(dh, dh2) and torch.equal(dw, dw2) and torch.equal(db, db2)


def test_fused_backward_random_shapes():
    """24 seeded random shapes (lattice 1 ... ~25 000 nodes: the shared-vocabulary launch of small and surplus row tiles
    and the plain one; vocabularies on both sides of every tile boundary; contiguous, padded and merely 16-byte-aligned
    rows).  A one-off run of the same loop over 240 further shapes passed on the round-3 build."""
    rng = np.random.default_rng(2024)
    for it in range(24):
        B = int(rng.integers(1, 7))
        T = int(rng.integers(1, 200))
        U = int(rng.integers(0, 24))
        V = int(rng.choice([128, 129, 136, 255, 257, 520, 1027, 2048, 4234, int(rng.integers(128, 3000))]))
        dtype = torch.bfloat16 if rng.random() < 0.6 else torch.float16
        padded = (V + 127) // 128 * 128
        ld = int(rng.choice([padded, padded + 128, V if V % 8 == 0 else padded, max((V + 7) // 8 * 8, padded if V % 8 else 0)]))
        hidden, weight, bias, labels, al, ll = make_inputs(B, T, U, V, 512, dtype, seed=500 + it)
        st, logits, _, ws = forward_c_abi(hidden, weight, bias, labels, al, ll, ld=ld)
        assert st == 0, (B, T, U, V, ld)
        go = rng.uniform(0.2, 1.5, B).astype(np.float32)
        g = k3_c_abi(logits[..., :V].contiguous(), labels, al, ll, ws, go, 1, 1.0)
        st, dh, dw, db = backward_c_abi(hidden, weight, logits, labels, al, ll, ws, go, 1, 1.0, ld=ld)
        assert st == 0, (B, T, U, V, ld)
        check_against_k3(hidden, weight, g, dh, dw, db)


def test_fused_backward_stride_checks():
    hidden, weight, bias, labels, al, ll = make_inputs(1, 4, 2, 4234, 512, torch.bfloat16, seed=1)
    st, logits, _, ws = forward_c_abi(hidden, weight, bias, labels, al, ll, ld=4352)
    assert st == 0
    call = lambda ld: backward_c_abi(hidden, weight, logits, labels, al, ll, ws, [1.0], 0, 1.0, ld=ld)[0]
    assert call(4352) == 0
    assert call(4240) == 2      # rows not padded to whole tiles and V no multiple of 8: overlapping tiles unaligned
    assert call(4354) == 2      # rows must start on 16 bytes
    assert call(4232) == 2      # shorter than a row


def _forward_blank(hidden, weight, bias, labels, al, ll, blank):
    """The unfused forward (library GEMM + rnnt_forward) with a blank other than 0; same workspace layout."""
    from rnnt_amd import _lib
    from rnnt_amd.loss import _DTYPES

    lib = _lib.load()
    B, T, U1, H = hidden.shape
    V = weight.shape[0]
    logits = torch.nn.functional.linear(hidden.to(DEV), weight.to(DEV), bias.to(DEV)).contiguous()
    lab, ald, lld = (torch.from_numpy(x).to(DEV) for x in (labels, al, ll))
    costs = torch.empty(B, dtype=torch.float32, device=DEV)
    ws = torch.empty(_lib.workspace_bytes(T, U1, B), dtype=torch.uint8, device=DEV)
    opts = _lib.RnntOptions(torch.cuda.current_stream().cuda_stream, blank, T, U1, _DTYPES[hidden.dtype], None)
    st = lib.rnnt_forward(logits.data_ptr(), lab.data_ptr(), ald.data_ptr(), lld.data_ptr(), V, B, costs.data_ptr(),
                          None, 0, ws.data_ptr(), opts)
    torch.cuda.synchronize()
    return st, logits, costs, ws


def test_fused_backward_is_bitwise_repeatable():
    """No atomics: 40 launches on the same inputs give the same bits (also the race screen of the LDS pipeline)."""
    hidden, weight, bias, labels, al, ll = make_inputs(6, 90, 14, 1304, 512, torch.bfloat16, seed=31)
    st, logits, _, ws = forward_c_abi(hidden, weight, bias, labels, al, ll)
    assert st == 0
    first = backward_c_abi(hidden, weight, logits, labels, al, ll, ws, [1.0], 0, 1.0)
    assert first[0] == 0
    for i in range(40):
        again = backward_c_abi(hidden, weight, logits, labels, al, ll, ws, [1.0], 0, 1.0)
        assert all(torch.equal(a, b) for a, b in zip(first[1:], again[1:])), f"launch {i} differs"


def test_fused_backward_argument_checks():
    from rnnt_amd import _lib

    lib = _lib.load()
    n = ctypes.c_size_t(0)
    assert lib.rnnt_joint_loss_backward_workspace_size(256, 4232, 2, 10, 5, ctypes.byref(n)) == 2    # H
    assert lib.rnnt_joint_loss_backward_workspace_size(512, 100, 2, 10, 5, ctypes.byref(n)) == 2     # V < 128
    # (the reference's 4234 and 5455 are taken through rnnt_joint_loss_backward_strided on padded rows)
    assert lib.rnnt_joint_loss_backward_workspace_size(512, 4234, 2, 10, 5, ctypes.byref(n)) == 0
    assert lib.rnnt_joint_loss_backward_workspace_size(512, 4232, 2, 10, 5, ctypes.byref(n)) == 0 and n.value > 0
    hidden, weight, bias, labels, al, ll = make_inputs(1, 4, 2, 128, 512, torch.bfloat16, seed=1)
    st, logits, _, ws = forward_c_abi(hidden, weight, bias, labels, al, ll)
    assert st == 0
    h, w = hidden.to(DEV), weight.to(DEV)
    lab, ald, lld = (torch.from_numpy(x).to(DEV) for x in (labels, al, ll))
    bws = torch.empty(int(n.value), dtype=torch.uint8, device=DEV)
    go = torch.ones(1, device=DEV)
    dh = torch.empty_like(h)

    def call(H, V, dtype, bw=bws):
        opts = _lib.RnntOptions(None, 0, 4, 3, dtype, None)
        return lib.rnnt_joint_loss_backward(h.data_ptr(), w.data_ptr(), logits.data_ptr(), lab.data_ptr(),
                                            ald.data_ptr(), lld.data_ptr(), H, V, 1, go.data_ptr(), 0,
                                            ctypes.c_float(1.0), dh.data_ptr(), None, None, ws.data_ptr(),
                                            bw.data_ptr() if bw is not None else None, opts)

    assert call(512, 128, _lib.DTYPE_F32) == 2
    assert call(256, 128, _lib.DTYPE_BF16) == 2
    assert call(512, 64, _lib.DTYPE_BF16) == 2
    assert call(512, 128, _lib.DTYPE_BF16, bw=None) == 2
    assert call(512, 130, _lib.DTYPE_BF16) == 2        # contiguous rows of a vocabulary that is no multiple of 8
    assert call(512, 128, _lib.DTYPE_BF16) == 0
    torch.cuda.synchronize()


def test_fused_backward_config2_shape_full_size():
    """B=32 T=400 U=40 V=4232 H=512 bf16 (BASELINE config 2's lattice), ragged: the three gradients against
    rnnt_backward + the library's products of the same gradient tensor (fp32 accumulation in a different order)."""
    B, T, U, V, H = 32, 400, 40, 4232, 512
    g = torch.Generator(device=DEV).manual_seed(3)
    hidden = torch.tanh(torch.randn(B, T, U + 1, H, device=DEV, generator=g)).to(torch.bfloat16)
    weight = (torch.randn(V, H, device=DEV, generator=g) * 0.09).to(torch.bfloat16)
    bias = (torch.randn(V, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
    rng = np.random.default_rng(3)
    labels = rng.integers(1, V, size=(B, U)).astype(np.int32)
    al = rng.integers(100, T + 1, size=B).astype(np.int32)
    ll = rng.integers(10, U + 1, size=B).astype(np.int32)
    al[0], ll[0] = T, U
    st, logits, costs, ws = forward_c_abi(hidden, weight, bias, labels, al, ll)
    assert st == 0
    gl = k3_c_abi(logits, labels, al, ll, ws, [0.5], 0, 1.0 / B)
    st, dh, dw, db = backward_c_abi(hidden, weight, logits, labels, al, ll, ws, [0.5], 0, 1.0 / B)
    assert st == 0
    g2 = gl.view(-1, V)
    ref_dh = torch.mm(g2.float(), weight.float()).view_as(hidden)          # fp32 library GEMMs of the same g
    ref_dw = torch.mm(g2.float().t(), hidden.view(-1, H).float())
    ref_db = g2.sum(0, dtype=torch.float32)
    for name, got, ref in (("hidden", dh, ref_dh), ("weight", dw, ref_dw), ("bias", db, ref_db)):
        assert bool(torch.isfinite(got.float()).all()), name
        err = float((got.float() - ref).abs().max())
        scale = float(ref.abs().max())
        assert err <= 2 * EPS[torch.bfloat16] * scale, (name, err, scale)
    # rows of padding nodes get exactly zero
    b = int(np.argmin(al))
    assert float(dh[b, int(al[b]):].float().abs().max()) == 0.0
