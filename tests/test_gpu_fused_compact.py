"""GPU parity for the COMPACTED node order of the fused projection + loss (include/rnnt_mi355.h:
rnnt_joint_loss_forward_compact / rnnt_joint_loss_backward_compact), through the C ABI.

The reference computes the logits of all B * maxT * (U+1) rows (/root/reference/src/rnnt/model.py:46-59) and its loss
ignores the rows outside the utterances' lattices (:116).  The compact calls compute, store and read back the nodes
inside the lattices only.  Checked here:

* forward: the stored logits, node by node, are bit for bit the rows the dense call stores; costs equal the dense
  call's and the C loss oracle's (PARITY UNPINNED, oracle/__init__.py) at the loss's tolerance; the node -> row map the
  device built equals the one the host builds from the lengths; the hidden vectors of padding rows are never read (NaN
  there changes nothing);
* backward: grad_hidden / grad_weight / grad_bias against rnnt_backward on the same logits and workspace followed by
  the two products in float64 (the check of tests/test_gpu_fused_bwd.py); rows of grad_hidden outside the lattices are
  exactly zero;
* a logits buffer smaller than the lattices gives NaN costs for the utterances it cuts, nothing else changes;
* batches without any node, with utterances whose lengths do not fit, and the Python operator with and without the
  host's node count.
"""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from test_gpu_fused import DEV, forward_c_abi, make_inputs
from test_gpu_fused_bwd import check_against_k3, k3_c_abi

pytestmark = pytest.mark.gpu


def host_rowmap(al, ll, T, U1):
    """Row of the [B, T, U1] tensor of every node inside the lattices, in the compacted order."""
    rows = []
    for b in range(len(al)):
        if not (1 <= al[b] <= T and 0 <= ll[b] < U1):
            continue
        for t in range(int(al[b])):
            rows.extend((b * T + t) * U1 + u for u in range(int(ll[b]) + 1))
    return np.asarray(rows, dtype=np.int64)


def forward_compact(hidden, weight, bias, labels, al, ll, ld, cap, reduction=0):
    """rnnt_joint_loss_forward_compact as a C caller would use it."""
    from rnnt_amd import _lib
    from rnnt_amd.loss import _DTYPES

    lib = _lib.load()
    B, T, U1, H = hidden.shape
    V = weight.shape[0]
    h, w = hidden.to(DEV).contiguous(), weight.to(DEV).contiguous()
    b32 = bias.to(DEV).float().contiguous()
    lab, ald, lld = (torch.from_numpy(x).to(DEV) for x in (labels, al, ll))
    logits = torch.full((cap, ld), float("nan"), dtype=hidden.dtype, device=DEV)
    costs = torch.empty(B, dtype=torch.float32, device=DEV)
    ws = torch.full((_lib.workspace_bytes(T, U1, B),), 0xFF, dtype=torch.uint8, device=DEV)
    n = ctypes.c_size_t(0)
    assert lib.rnnt_joint_loss_compact_workspace_size(B, T, U1, ctypes.byref(n)) == 0
    cws = torch.full((int(n.value),), 0xFF, dtype=torch.uint8, device=DEV)
    opts = _lib.RnntOptions(torch.cuda.current_stream().cuda_stream, 0, T, U1, _DTYPES[hidden.dtype], None)
    st = lib.rnnt_joint_loss_forward_compact(h.data_ptr(), w.data_ptr(), b32.data_ptr(), logits.data_ptr(), ld, cap,
                                             lab.data_ptr(), ald.data_ptr(), lld.data_ptr(), H, V, B, costs.data_ptr(),
                                             None, reduction, ws.data_ptr(), cws.data_ptr(), opts)
    torch.cuda.synchronize()
    return st, logits, costs, ws, cws


def backward_compact(hidden, weight, logits, ld, cap, labels, al, ll, ws, cws, go, stride, scale, want=(True, True, True),
                     zero_padding_rows=1):
    from rnnt_amd import _lib
    from rnnt_amd.loss import _DTYPES

    lib = _lib.load()
    B, T, U1, H = hidden.shape
    V = weight.shape[0]
    h, w = hidden.to(DEV).contiguous(), weight.to(DEV).contiguous()
    lab, ald, lld = (torch.from_numpy(x).to(DEV) for x in (labels, al, ll))
    n = ctypes.c_size_t(0)
    assert lib.rnnt_joint_loss_backward_compact_workspace_size(H, V, cap, ctypes.byref(n)) == 0
    bws = torch.full((int(n.value),), 0xFF, dtype=torch.uint8, device=DEV)
    dh = torch.full_like(h, float("nan")) if want[0] else None
    dw = torch.full_like(w, float("nan")) if want[1] else None
    db = torch.full((V,), float("nan"), dtype=torch.float32, device=DEV) if want[2] else None
    god = torch.as_tensor(go, dtype=torch.float32, device=DEV).reshape(-1)
    opts = _lib.RnntOptions(torch.cuda.current_stream().cuda_stream, 0, T, U1, _DTYPES[hidden.dtype], None)
    st = lib.rnnt_joint_loss_backward_compact(h.data_ptr(), w.data_ptr(), logits.data_ptr(), ld, cap, lab.data_ptr(),
                                              ald.data_ptr(), lld.data_ptr(), H, V, B, god.data_ptr(), stride,
                                              ctypes.c_float(scale), dh.data_ptr() if want[0] else None, zero_padding_rows,
                                              dw.data_ptr() if want[1] else None, db.data_ptr() if want[2] else None,
                                              ws.data_ptr(), cws.data_ptr(), bws.data_ptr(), opts)
    torch.cuda.synchronize()
    return st, dh, dw, db


def poison_padding(hidden, al, ll):
    """NaN in the hidden vectors of every row outside the lattices: the compact calls must never read them."""
    h = hidden.clone()
    for b in range(h.shape[0]):
        h[b, int(al[b]):] = float("nan")
        h[b, :, int(ll[b]) + 1:] = float("nan")
    return h


CASES = [
    # B, T, U, V, dtype, what
    (5, 60, 9, 4232, torch.bfloat16, "the reference's vocabulary; a few row tiles"),
    (7, 129, 20, 520, torch.float16, "18 963 rows, ragged; fp16"),
    (3, 33, 7, 4234, torch.bfloat16, "the reference's 4234 (padded rows only)"),
    (1, 40, 6, 128, torch.bfloat16, "one utterance: compact order = tensor order"),
    (40, 150, 15, 1000, torch.bfloat16, "96 000 rows: whole rounds of CUs, surplus row tiles, several node ranges"),
    (6, 16, 0, 300, torch.float16, "no labels at all (U = 0)"),
]


@pytest.mark.parametrize("exact_cap", [True, False])
@pytest.mark.parametrize("B,T,U,V,dtype,what", CASES)
def test_compact_forward_and_backward(B, T, U, V, dtype, what, exact_cap):
    H = 512
    hidden, weight, bias, labels, al, ll = make_inputs(B, T, U, V, H, dtype, seed=3 * V + T)
    U1 = U + 1
    ld = (V + 127) // 128 * 128
    rows = host_rowmap(al, ll, T, U1)
    n = len(rows)
    cap = n if exact_cap else B * T * U1
    # the dense call: what the compact one must reproduce node by node
    st, dense, dense_costs, _ = forward_c_abi(hidden, weight, bias, labels, al, ll, ld=ld)
    assert st == 0, what
    st, logits, costs, ws, cws = forward_compact(poison_padding(hidden, al, ll), weight, bias, labels, al, ll, ld, cap)
    assert st == 0, what
    rmap = cws.view(torch.int32)[:n].cpu().numpy()
    np.testing.assert_array_equal(rmap, rows, err_msg=what)
    got = logits[:n, :V].contiguous().view(torch.int16).cpu()
    ref = dense.view(-1, ld)[torch.from_numpy(rows).to(DEV)][:, :V].contiguous().view(torch.int16).cpu()
    assert torch.equal(got, ref), what
    assert bool(torch.isnan(logits[:n, V:].float()).all()) and bool(torch.isnan(logits[n:].float()).all()), \
        "the padding of a row or a row past the last node was written"
    # (the surplus rule may cut the two launches differently: the {max, sum} pairs of a node are then merged in another
    # order)
    np.testing.assert_allclose(costs.cpu().numpy(), dense_costs.cpu().numpy(), rtol=2e-6, atol=0, err_msg=what)
    full = torch.zeros(B * T * U1, V, dtype=dtype, device=DEV)
    full[torch.from_numpy(rows).to(DEV)] = logits[:n, :V]
    full = full.view(B, T, U1, V)
    c, _, _ = oracle.rnnt_loss_c(full.float().cpu().numpy(), labels, al, ll, want_grads=False)
    np.testing.assert_allclose(costs.cpu().numpy(), c, rtol=1e-5, atol=0, err_msg=what)

    # ---- backward: K3 on the same logits (scattered back into the tensor's order) and the same workspace
    go = np.linspace(0.25, 1.5, B).astype(np.float32)
    g = k3_c_abi(full.contiguous(), labels, al, ll, ws, go, 1, 1.0)
    st, dh, dw, db = backward_compact(poison_padding(hidden, al, ll), weight, logits, ld, cap, labels, al, ll, ws, cws,
                                      go, 1, 1.0)
    assert st == 0, what
    inside = torch.zeros(B * T * U1, dtype=torch.bool, device=DEV)
    inside[torch.from_numpy(rows).to(DEV)] = True
    assert float(dh.view(-1, H)[~inside].float().abs().sum()) == 0.0, "a row outside the lattices is not zero"
    check_against_k3(hidden, weight, g, dh, dw, db)
    # one output at a time, scalar upstream gradient with the 1/B of 'mean'
    g = k3_c_abi(full.contiguous(), labels, al, ll, ws, [0.5], 0, 1.0 / B)
    st, dh2, _, _ = backward_compact(hidden, weight, logits, ld, cap, labels, al, ll, ws, cws, [0.5], 0, 1.0 / B,
                                     want=(True, False, False))
    assert st == 0
    check_against_k3(hidden, weight, g, dh2, None, None)
    st, _, dw2, db2 = backward_compact(hidden, weight, logits, ld, cap, labels, al, ll, ws, cws, [0.5], 0, 1.0 / B,
                                       want=(False, True, True))
    assert st == 0
    check_against_k3(hidden, weight, g, None, dw2, db2)
    # zero_padding_rows = 0: the rows outside the lattices keep what they held, the others are the same bits
    st, dh3, _, _ = backward_compact(hidden, weight, logits, ld, cap, labels, al, ll, ws, cws, [0.5], 0, 1.0 / B,
                                     want=(True, False, False), zero_padding_rows=0)
    assert st == 0
    assert torch.equal(dh3.view(-1, H)[inside], dh2.view(-1, H)[inside])
    assert bool(torch.isnan(dh3.view(-1, H)[~inside].float()).all())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_compact_on_peaked_logits(dtype):
    """SURVEY §8d's "peaked" distribution through the fused kernels: logits of a wide dynamic range (weights four times
    the other tests', sigma of a logit about 6, extremes beyond +-30) and a blank logit 8 above the rest — what a trained
    model emits.  Costs against the C oracle on the stored logits, gradients against K3 on the same logits."""
    B, T, U, V, H = 4, 70, 12, 1304, 512
    hidden, weight, bias, labels, al, ll = make_inputs(B, T, U, V, H, dtype, seed=77)
    weight = (weight.float() * 4.0).to(dtype)
    bias = bias.float()
    bias[0] += 8.0
    bias = bias.to(dtype)
    U1, ld = U + 1, (V + 127) // 128 * 128
    rows = host_rowmap(al, ll, T, U1)
    n = len(rows)
    st, logits, costs, ws, cws = forward_compact(poison_padding(hidden, al, ll), weight, bias, labels, al, ll, ld, n)
    assert st == 0
    stored = logits[:n, :V].float()
    assert float(stored.max()) > 25.0 and float(stored.min()) < -25.0 and float(stored.std()) > 4.0
    full = torch.zeros(B * T * U1, V, dtype=dtype, device=DEV)
    full[torch.from_numpy(rows).to(DEV)] = logits[:n, :V]
    full = full.view(B, T, U1, V)
    c, _, _ = oracle.rnnt_loss_c(full.float().cpu().numpy(), labels, al, ll, want_grads=False)
    np.testing.assert_allclose(costs.cpu().numpy(), c, rtol=1e-5, atol=0)
    go = np.linspace(0.25, 1.5, B).astype(np.float32)
    g = k3_c_abi(full.contiguous(), labels, al, ll, ws, go, 1, 1.0)
    st, dh, dw, db = backward_compact(poison_padding(hidden, al, ll), weight, logits, ld, n, labels, al, ll, ws, cws,
                                      go, 1, 1.0)
    assert st == 0
    check_against_k3(hidden, weight, g, dh, dw, db)


def test_compact_on_a_batch_without_padding_equals_the_dense_calls():
    """Uniform lengths: the compacted order is the tensor's order; logits, costs and all three gradients are the dense
    entry points' bit for bit (same tiles, same node ranges, same order of every sum)."""
    from test_gpu_fused_bwd import backward_c_abi

    B, T, U, V, H = 4, 70, 12, 1304, 512
    hidden, weight, bias, labels, al, ll = make_inputs(B, T, U, V, H, torch.bfloat16, seed=77, ragged=False)
    ld = 1408
    st, dense, dense_costs, dws = forward_c_abi(hidden, weight, bias, labels, al, ll, ld=ld)
    assert st == 0
    cap = B * T * (U + 1)
    st, logits, costs, ws, cws = forward_compact(hidden, weight, bias, labels, al, ll, ld, cap)
    assert st == 0
    assert torch.equal(logits[:, :V].contiguous().view(torch.int16), dense.view(-1, ld)[:, :V].contiguous().view(torch.int16))
    assert torch.equal(costs, dense_costs)
    st, dh0, dw0, db0 = backward_c_abi(hidden, weight, dense, labels, al, ll, dws, [0.5], 0, 0.25, ld=ld)
    assert st == 0
    st, dh, dw, db = backward_compact(hidden, weight, logits, ld, cap, labels, al, ll, ws, cws, [0.5], 0, 0.25)
    assert st == 0
    assert torch.equal(dh, dh0) and torch.equal(dw, dw0) and torch.equal(db, db0)


def test_compact_logits_buffer_smaller_than_the_lattices():
    """logits_rows below the number of nodes: the nodes past it are not computed, the costs of their utterances are
    NaN (not garbage), the other utterances' costs and gradients are what they are with room for everything."""
    B, T, U, V, H = 4, 50, 8, 640, 512
    hidden, weight, bias, labels, al, ll = make_inputs(B, T, U, V, H, torch.bfloat16, seed=5)
    rows = host_rowmap(al, ll, T, U + 1)
    n = len(rows)
    ld = 640
    st, _, costs_all, ws0, cws0 = forward_compact(hidden, weight, bias, labels, al, ll, ld, n)
    assert st == 0 and bool(torch.isfinite(costs_all).all())
    cut = n - 7                                    # cuts into the last utterance only
    assert cut > n - int(al[-1]) * (int(ll[-1]) + 1)
    st, logits, costs, ws, cws = forward_compact(hidden, weight, bias, labels, al, ll, ld, cut)
    assert st == 0
    assert torch.equal(costs[:-1], costs_all[:-1]) and bool(torch.isnan(costs[-1]))
    st, dh, dw, db = backward_compact(hidden, weight, logits, ld, cut, labels, al, ll, ws, cws, [1.0, 1.0, 1.0, 0.0], 1, 1.0)
    assert st == 0
    # (a NaN cost comes with NaN gradients, as in rnnt_backward: the reference's loop drops such a batch, bin/train.py:36-41)
    assert bool(torch.isfinite(dh[:-1].float()).all())


def test_compact_batches_with_nothing_or_with_misfits():
    """An utterance whose lengths do not fit the tensor has no nodes (NaN cost, zero gradient, no address formed from
    its lengths); a batch of such utterances only launches workgroups that leave at once."""
    B, T, U, V, H = 3, 20, 5, 256, 512
    hidden, weight, bias, labels, al, ll = make_inputs(B, T, U, V, H, torch.float16, seed=9)
    bad_al = al.copy()
    bad_al[1] = T + 3
    rows = host_rowmap(bad_al, ll, T, U + 1)
    st, logits, costs, ws, cws = forward_compact(hidden, weight, bias, labels, bad_al, ll, 256, B * T * (U + 1))
    assert st == 0
    c = costs.cpu().numpy()
    assert np.isnan(c[1]) and np.isfinite(c[[0, 2]]).all()
    st, dh, dw, db = backward_compact(hidden, weight, logits, 256, B * T * (U + 1), labels, bad_al, ll, ws, cws,
                                      [1.0, 0.0, 1.0], 1, 1.0)
    assert st == 0
    assert float(dh[1].float().abs().sum()) == 0.0 and bool(torch.isfinite(dw.float()).all())
    assert len(rows) == int(bad_al[0]) * (int(ll[0]) + 1) + int(bad_al[2]) * (int(ll[2]) + 1)
    none_al = np.full(B, T + 1, dtype=np.int32)
    st, logits, costs, ws, cws = forward_compact(hidden, weight, bias, labels, none_al, ll, 256, 64)
    assert st == 0 and bool(torch.isnan(costs).all())
    st, dh, dw, db = backward_compact(hidden, weight, logits, 256, 64, labels, none_al, ll, ws, cws, [0.0], 0, 1.0)
    assert st == 0
    assert float(dh.float().abs().sum()) == 0.0 and float(dw.float().abs().sum()) == 0.0 and float(db.abs().sum()) == 0.0


def test_compact_argument_checks():
    from rnnt_amd import _lib

    lib = _lib.load()
    n = ctypes.c_size_t(0)
    assert lib.rnnt_joint_loss_compact_workspace_size(2, 10, 5, ctypes.byref(n)) == 0 and n.value == 512
    assert lib.rnnt_joint_loss_compact_workspace_size(0, 10, 5, ctypes.byref(n)) == 2
    assert lib.rnnt_joint_loss_compact_workspace_size(1 << 15, 1 << 10, 1 << 7, ctypes.byref(n)) == 2    # rows beyond 31 bits
    assert lib.rnnt_joint_loss_backward_compact_workspace_size(512, 4234, 1000, ctypes.byref(n)) == 0 and n.value > 0
    assert lib.rnnt_joint_loss_backward_compact_workspace_size(256, 4234, 1000, ctypes.byref(n)) == 2
    assert lib.rnnt_joint_loss_backward_compact_workspace_size(512, 4234, 0, ctypes.byref(n)) == 2
    hidden, weight, bias, labels, al, ll = make_inputs(1, 4, 2, 128, 512, torch.bfloat16, seed=1)
    h, w, b32 = hidden.to(DEV), weight.to(DEV), bias.to(DEV).float()
    lab, ald, lld = (torch.from_numpy(x).to(DEV) for x in (labels, al, ll))
    logits = torch.empty(12, 128, dtype=torch.bfloat16, device=DEV)
    costs = torch.empty(1, device=DEV)
    ws = torch.empty(_lib.workspace_bytes(4, 3, 1), dtype=torch.uint8, device=DEV)
    cws = torch.empty(256, dtype=torch.uint8, device=DEV)
    opts = _lib.RnntOptions(None, 0, 4, 3, _lib.DTYPE_BF16, None)

    def fwd(cap, cw):
        return lib.rnnt_joint_loss_forward_compact(h.data_ptr(), w.data_ptr(), b32.data_ptr(), logits.data_ptr(), 128, cap,
                                                   lab.data_ptr(), ald.data_ptr(), lld.data_ptr(), 512, 128, 1,
                                                   costs.data_ptr(), None, 0, ws.data_ptr(), cw, opts)

    assert fwd(12, None) == 2            # no compact workspace
    assert fwd(0, cws.data_ptr()) == 2   # a buffer of no rows
    assert fwd(12, cws.data_ptr() + 4) == 2
    assert fwd(12, cws.data_ptr()) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_operator_with_and_without_the_hosts_node_count(dtype):
    """rnnt_amd.fused_joint_loss on a ragged batch: with valid_nodes (logits buffer and grids cut for the nodes), without
    (cut for the tensor, the surplus workgroups leave), and the round-2 backward (K3 + two library GEMMs on the dense
    logits) agree; JointNet.transducer_loss counts the nodes itself when the lengths arrive on the host."""
    import rnnt_amd

    B, T, U, V, H = 6, 90, 14, 1304, 512
    hidden, weight, bias, labels, al, ll = make_inputs(B, T, U, V, H, dtype, seed=31)
    lab, ald, lld = (torch.from_numpy(x).to(DEV) for x in (labels, al, ll))
    n = rnnt_amd.count_lattice_nodes(al, ll)
    assert n == len(host_rowmap(al, ll, T, U + 1))
    results = []
    for kw in (dict(valid_nodes=n), dict(), dict(fused_backward=False)):
        h = hidden.to(DEV).requires_grad_(True)
        w = weight.to(DEV).requires_grad_(True)
        b = bias.to(DEV).requires_grad_(True)
        loss = rnnt_amd.fused_joint_loss(h, w, b, lab, ald, lld, min_nodes=0, **kw)
        loss.backward()
        results.append((loss.detach(), h.grad, w.grad, b.grad))
    eps = 2.0 ** -8 if dtype is torch.bfloat16 else 2.0 ** -11
    for other in results[1:]:
        torch.testing.assert_close(other[0], results[0][0], rtol=2e-6, atol=0)
        for a, c in zip(other[1:], results[0][1:]):
            scale = float(c.float().abs().max())
            assert float((a.float() - c.float()).abs().max()) <= 2 * eps * scale
    # the same two results bit for bit: the bound only sizes buffers and grids when no surplus rule applies ... which it
    # may; what must hold exactly is that no row outside the lattices got a gradient
    inside = torch.zeros(B * T * (U + 1), dtype=torch.bool)
    inside[torch.from_numpy(host_rowmap(al, ll, T, U + 1))] = True
    for r in results:
        assert float(r[1].view(-1, H)[~inside.to(DEV)].float().abs().sum()) == 0.0

    # JointNet.transducer_loss: lengths on the host (as the reference has them) or on the device
    torch.manual_seed(0)
    joint = rnnt_amd.JointNet(640, 512, V).to(DEV).to(dtype)
    enc = torch.randn(B, T, 320, device=DEV).to(dtype)
    dec = torch.randn(B, U + 1, 320, device=DEV).to(dtype)
    on_host = joint.transducer_loss(enc, dec, lab, torch.from_numpy(al), torch.from_numpy(ll))
    on_device = joint.transducer_loss(enc, dec, lab, ald, lld)
    torch.testing.assert_close(on_host, on_device, rtol=2e-6, atol=0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_ragged_hidden_layer(dtype):
    """rnnt_joint_hidden_forward_ragged / _backward_ragged (the producer of `hidden` for the compact calls): inside the
    lattices the same bits as the dense pair; outside, forward writes nothing and backward reads nothing."""
    import rnnt_amd

    B, T, U1, H = 5, 37, 9, 512
    g = torch.Generator().manual_seed(11)
    enc = torch.randn(B, T, H, generator=g).to(dtype).to(DEV)
    dec = torch.randn(B, U1, H, generator=g).to(dtype).to(DEV)
    al = torch.tensor([37, 5, 16, 17, 40], dtype=torch.int32)       # the last one does not fit: no rows
    ll = torch.tensor([8, 0, 3, 8, 2], dtype=torch.int32)
    inside = torch.zeros(B, T, U1, dtype=torch.bool)
    for b in range(B - 1):
        inside[b, :al[b], :ll[b] + 1] = True
    inside = inside.to(DEV)
    e1, d1 = enc.clone().requires_grad_(True), dec.clone().requires_grad_(True)
    dense = rnnt_amd.joint_hidden(e1, d1)
    e2, d2 = enc.clone().requires_grad_(True), dec.clone().requires_grad_(True)
    ragged = rnnt_amd.joint_hidden(e2, d2, al.to(DEV), ll.to(DEV))
    assert torch.equal(ragged[inside], dense[inside])
    go = torch.randn(B, T, U1, H, generator=g).to(dtype).to(DEV)
    dense.backward(torch.where(inside[..., None], go, torch.zeros_like(go)))
    poisoned = torch.where(inside[..., None], go, torch.full_like(go, float("nan")))
    ragged.data[~inside] = float("nan")         # (the saved tensor: backward must not read these rows either)
    ragged.backward(poisoned)
    assert torch.equal(e2.grad, e1.grad) and torch.equal(d2.grad, d1.grad)
    assert float(e2.grad[1, 5:].float().abs().sum()) == 0.0 and float(d2.grad[2, 4:].float().abs().sum()) == 0.0
    assert float(e2.grad[4].float().abs().sum()) == 0.0
    with pytest.raises(ValueError):
        rnnt_amd.joint_hidden(enc, dec, al.to(DEV), None)
    with pytest.raises(ValueError):
        rnnt_amd.joint_hidden(enc, dec, al.to(DEV).long(), ll.to(DEV).long())


@pytest.mark.parametrize("dtype,H", [(torch.float32, 512), (torch.bfloat16, 512), (torch.float16, 330), (torch.float32, 37)])
def test_hidden_layer_in_the_compacted_node_order(dtype, H):
    """rnnt_joint_hidden_forward_compact / _backward_compact: row k of the [n, H] result is node k of
    rnnt_compact_rows, with the bits of the dense layer's row; backward gives the dense pair's gradients for a gradient
    that is zero outside the lattices; a buffer with slack keeps its surplus rows, one that is too small drops whole
    frames."""
    import rnnt_amd

    B, T, U1 = 6, 45, 9
    g = torch.Generator().manual_seed(5)
    enc = torch.randn(B, T, H, generator=g).to(dtype).to(DEV)
    dec = torch.randn(B, U1, H, generator=g).to(dtype).to(DEV)
    al = np.array([45, 5, 16, 17, 50, 33], dtype=np.int32)            # utterance 4 does not fit: no nodes
    ll = np.array([8, 0, 3, 8, 2, 5], dtype=np.int32)
    ald, lld = torch.from_numpy(al).to(DEV), torch.from_numpy(ll).to(DEV)
    rows = torch.from_numpy(host_rowmap(al, ll, T, U1)).to(DEV)
    n = rows.numel()
    assert n == rnnt_amd.count_lattice_nodes(al, ll, T, U1)
    e1, d1 = enc.clone().requires_grad_(True), dec.clone().requires_grad_(True)
    dense = rnnt_amd.joint_hidden(e1, d1)
    e2, d2 = enc.clone().requires_grad_(True), dec.clone().requires_grad_(True)
    nodes = rnnt_amd.joint_hidden(e2, d2, ald, lld, nodes=n)
    assert nodes.shape == (n, H)
    assert torch.equal(nodes, dense.view(-1, H)[rows])
    go = torch.randn(n, H, generator=g).to(dtype).to(DEV)
    full = torch.zeros(B * T * U1, H, dtype=dtype, device=DEV)
    full[rows] = go
    dense.backward(full.view(B, T, U1, H))
    nodes.backward(go)
    assert torch.equal(e2.grad, e1.grad) and torch.equal(d2.grad, d1.grad)
    assert float(e2.grad[4].float().abs().sum()) == 0.0 and float(d2.grad[4].float().abs().sum()) == 0.0

    # slack: rows past the last node are not written; too small: the frames that do not fit are dropped whole
    lib = rnnt_amd._lib.load()
    dt = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[dtype]
    stream = torch.cuda.current_stream().cuda_stream
    buf = torch.full((n + 7, H), 3.0, dtype=dtype, device=DEV)
    assert lib.rnnt_joint_hidden_forward_compact(enc.data_ptr(), dec.data_ptr(), buf.data_ptr(), n + 7, ald.data_ptr(),
                                                 lld.data_ptr(), B, T, U1, H, dt, stream) == 0
    assert torch.equal(buf[:n], nodes) and bool((buf[n:] == 3.0).all())
    cut = n - 4                                                       # the last utterance has 6 rows per frame
    buf = torch.full((cut, H), 3.0, dtype=dtype, device=DEV)
    assert lib.rnnt_joint_hidden_forward_compact(enc.data_ptr(), dec.data_ptr(), buf.data_ptr(), cut, ald.data_ptr(),
                                                 lld.data_ptr(), B, T, U1, H, dt, stream) == 0
    assert torch.equal(buf[:n - 6], nodes[:n - 6]) and bool((buf[n - 6:] == 3.0).all())
    ge, gd = torch.empty_like(enc), torch.empty_like(dec)
    nbytes = ctypes.c_size_t(0)
    assert lib.rnnt_joint_hidden_workspace_size(B, T, U1, H, ctypes.byref(nbytes)) == 0
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=DEV)
    hid_cut, go_cut = nodes.detach()[:cut].contiguous(), go[:cut].contiguous()
    assert lib.rnnt_joint_hidden_backward_compact(hid_cut.data_ptr(), go_cut.data_ptr(), cut, ge.data_ptr(), gd.data_ptr(),
                                                  ald.data_ptr(), lld.data_ptr(), B, T, U1, H, dt, ws.data_ptr(), stream) == 0
    e3, d3 = enc.clone().requires_grad_(True), dec.clone().requires_grad_(True)
    full = torch.zeros(B * T * U1, H, dtype=dtype, device=DEV)
    full[rows[:n - 6]] = go[:n - 6]                                    # the last frame of the last utterance: dropped
    rnnt_amd.joint_hidden(e3, d3).backward(full.view(B, T, U1, H))
    assert torch.equal(ge, e3.grad) and torch.equal(gd, d3.grad)
    # argument checks
    assert lib.rnnt_joint_hidden_forward_compact(enc.data_ptr(), dec.data_ptr(), buf.data_ptr(), 0, ald.data_ptr(),
                                                 lld.data_ptr(), B, T, U1, H, dt, stream) != 0
    assert lib.rnnt_joint_hidden_forward_compact(enc.data_ptr(), dec.data_ptr(), buf.data_ptr(), cut, None,
                                                 lld.data_ptr(), B, T, U1, H, dt, stream) != 0
    with pytest.raises(ValueError):
        rnnt_amd.joint_hidden(enc, dec, nodes=n)


def _random_compact_case(rng, it):
    B = int(rng.integers(1, 9))
    T = int(rng.integers(1, 160))
    U = int(rng.integers(0, 24))
    V = int(rng.choice([128, 129, 255, 520, 1027, 2048, 4234, int(rng.integers(128, 3000))]))
    dtype = torch.bfloat16 if rng.random() < 0.6 else torch.float16
    hidden, weight, bias, labels, al, ll = make_inputs(B, T, U, V, 512, dtype, seed=900 + it)
    if rng.random() < 0.3:                      # an utterance without nodes among the others
        al = al.copy()
        al[int(rng.integers(0, B))] = T + 1
    rows = host_rowmap(al, ll, T, U + 1)
    n = len(rows)
    slack = int(rng.choice([0, 1, 37, 300]))
    cap = max(1, min(n + slack, B * T * (U + 1)))
    ld = (V + 127) // 128 * 128 + int(rng.choice([0, 128]))
    return hidden, weight, bias, labels, al, ll, rows, n, cap, ld, V, dtype


def test_compact_random_shapes():
    """20 seeded random ragged batches (1 ... ~25 000 rows; vocabularies on both sides of the tile boundaries; logits
    buffers with and without slack; utterances without nodes): costs against the dense call, gradients against
    rnnt_backward + float64 products.  A one-off run of the same loop over 300 further seeds passed on this build."""
    rng = np.random.default_rng(77)
    for it in range(20):
        hidden, weight, bias, labels, al, ll, rows, n, cap, ld, V, dtype = _random_compact_case(rng, it)
        B, T, U1, H = hidden.shape
        what = (it, B, T, U1 - 1, V, n, cap, ld)
        st, dense, dense_costs, _ = forward_c_abi(hidden, weight, bias, labels, al, ll, ld=ld)
        assert st == 0, what
        st, logits, costs, ws, cws = forward_compact(poison_padding(hidden, al, ll), weight, bias, labels, al, ll, ld, cap)
        assert st == 0, what
        idx = torch.from_numpy(rows).to(DEV)
        assert torch.equal(logits[:n, :V].contiguous().view(torch.int16),
                           dense.view(-1, ld)[idx][:, :V].contiguous().view(torch.int16)), what
        np.testing.assert_allclose(costs.cpu().numpy(), dense_costs.cpu().numpy(), rtol=2e-6, atol=0, equal_nan=True,
                                   err_msg=str(what))
        if n == 0:
            continue
        full = torch.zeros(B * T * U1, V, dtype=dtype, device=DEV)
        full[idx] = logits[:n, :V]
        go = rng.uniform(0.2, 1.5, B).astype(np.float32)
        fit = (al >= 1) & (al <= T)
        go[~fit] = 0.0                           # (a NaN cost comes with NaN gradients; the loop drops such a batch)
        g = k3_c_abi(full.view(B, T, U1, V).contiguous(), labels, al, ll, ws, go, 1, 1.0)
        g = torch.where(torch.from_numpy(fit).to(DEV)[:, None, None, None], g, torch.zeros_like(g))
        st, dh, dw, db = backward_compact(poison_padding(hidden, al, ll), weight, logits, ld, cap, labels, al, ll, ws, cws,
                                          go, 1, 1.0)
        assert st == 0, what
        check_against_k3(hidden, weight, g, dh, dw, db)


def test_compact_is_bitwise_repeatable():
    """No atomics, and the race screen of the re-timed dW pipeline (node records three steps ahead, row numbers read
    back from LDS): 40 launches of forward + backward on one ragged batch give the same bits."""
    hidden, weight, bias, labels, al, ll = make_inputs(6, 90, 14, 1304, 512, torch.bfloat16, seed=31)
    n = len(host_rowmap(al, ll, 90, 15))
    first = None
    for i in range(40):
        st, logits, costs, ws, cws = forward_compact(hidden, weight, bias, labels, al, ll, 1408, n + 11)
        assert st == 0
        st, dh, dw, db = backward_compact(hidden, weight, logits, 1408, n + 11, labels, al, ll, ws, cws, [1.0], 0, 1.0)
        assert st == 0
        got = (logits[:n, :1304].clone(), costs, dh, dw, db)
        if first is None:
            first = got
        else:
            assert all(torch.equal(a, b) for a, b in zip(first, got)), f"launch {i} differs"
