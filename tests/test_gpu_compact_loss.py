"""GPU parity for the loss on logits kept in the compacted node order (include/rnnt_mi355.h: rnnt_compact_rows,
rnnt_forward_compact, rnnt_backward_compact; rnnt_amd/compact.py), through the C ABI.

The three passes run with one row of logits per lattice NODE instead of one per row of the [B, T, U+1] tensor the
reference hands its loss (/root/reference/src/rnnt/model.py:116).  The per-node arithmetic is that of rnnt_forward /
rnnt_backward, which tests/test_gpu_parity.py holds to the C oracle (PARITY UNPINNED, oracle/__init__.py), so the bar
here is exact where it can be: the same costs and the node rows of the same gradient bit for bit when a row of logits
is a whole number of 128-byte lines (and longer than 2 KiB: shorter rows take the several-nodes-per-wave kernels on the
padded tensor and one wave per node in the compacted layout).  Any other row starts at varying offsets inside a line; the kernels shift their
lane assignment by that offset (every wave instruction covers one aligned KiB), the offset decides which lane adds
which exponentials, and a node's row starts elsewhere in the two layouts: there the log-sum may differ in its last bit
(as it does between two nodes of ONE layout); one such bit moves alpha and beta of the nodes behind it by an ulp of
THEIR magnitude, so the bar there is the loss's own: costs 1e-5 relative, gradients 1e-4 absolute.
"""
import numpy as np
import pytest
import torch

import rnnt_amd
from cases import make_case

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def same(a, b, exact):
    if exact:
        return torch.equal(a, b)
    # the loss's own bar (north_star; tests/test_gpu_parity.py): costs 1e-5 relative, gradients 1e-4 absolute
    # (16-bit gradients: a last-bit difference before the rounding can land on the other side of a rounding boundary)
    step = {torch.float32: 0.0, torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}[a.dtype]
    a, b = a.double(), b.double()
    return bool(((a - b).abs() <= (1e-5 + step) * b.abs() + 1e-4).all())


def host_rowmap(al, ll, T, U1):
    rows = []
    for b in range(len(al)):
        if not (1 <= al[b] <= T and 0 <= ll[b] < U1):
            continue
        for t in range(int(al[b])):
            rows.extend((b * T + t) * U1 + u for u in range(int(ll[b]) + 1))
    return np.asarray(rows, dtype=np.int64)


CASES = [
    # B, T, U, V, dtype, what
    (5, 60, 9, 4232, torch.float32, "the reference's vocabulary, fp32: one wave per node, 17 chunks per lane"),
    (4, 40, 8, 4224, torch.float32, "rows of whole 128-byte lines: bit for bit"),
    (4, 50, 12, 2048, torch.float16, "fp16, rows of whole lines: bit for bit"),
    (4, 50, 12, 1024, torch.float16, "rows of 2 KiB: the padded tensor takes the small-row K1, the compacted layout one wave per node"),
    (3, 33, 7, 4234, torch.float32, "the reference's 4234: rows 8-byte aligned (the misaligned kernels)"),
    (4, 40, 8, 5455, torch.float32, "config/magic_*.yaml's 5455: rows 4-byte aligned"),
    (5, 60, 9, 4232, torch.bfloat16, "bf16"),
    (4, 50, 12, 1000, torch.float16, "fp16, short rows"),
    (6, 30, 5, 29, torch.float32, "a small vocabulary: the compacted layout takes the one-wave-per-node kernels"),
    (2, 9, 3, 20000, torch.float32, "rows of two waves"),
    (1, 7, 2, 70001, torch.float32, "rows beyond four waves' registers, odd: the element-granular kernels"),
    (2, 6, 2, 131072, torch.bfloat16, "rows beyond four waves' registers: the streaming K1"),
    (6, 16, 0, 300, torch.float16, "no labels at all (U = 0)"),
]


@pytest.mark.parametrize("B,T,U,V,dtype,what", CASES)
def test_compact_loss_equals_the_loss_on_the_padded_tensor(B, T, U, V, dtype, what):
    acts_np, labels, al, ll = make_case(B, T, U, V, seed=V + T, ragged=True)
    acts = torch.from_numpy(acts_np).to(dtype).to(DEV).requires_grad_(True)
    lab, ald, lld = (torch.from_numpy(x).to(DEV) for x in (labels, al, ll))
    go = torch.linspace(0.25, 1.5, B, device=DEV)
    costs = rnnt_amd.rnnt_loss(acts, lab, ald, lld, reduction="none", check_lengths=False)
    costs.backward(go)

    rows = host_rowmap(al, ll, T, U + 1)
    n = len(rows)
    assert n == rnnt_amd.count_lattice_nodes(al, ll, T, U + 1)
    rowmap = rnnt_amd.compact_rowmap(ald, lld, T, U + 1, n)
    np.testing.assert_array_equal(rowmap.cpu().numpy(), rows, err_msg=what)
    idx = rowmap.long()
    nodes = acts.detach().view(-1, V).index_select(0, idx).contiguous().requires_grad_(True)
    ccosts = rnnt_amd.rnnt_loss_compact(nodes, lab, ald, lld, T, reduction="none", exact_rows=True)
    # rows are whole 128-byte lines, and long enough for the same kernels in both layouts (module docstring)
    exact = (V * acts.element_size()) % 128 == 0 and V * acts.element_size() > 2048
    assert same(ccosts, costs, exact), what
    ccosts.backward(go)
    ref = acts.grad.view(-1, V).index_select(0, idx)
    assert same(nodes.grad, ref, exact), what

    # 'mean' with a scalar upstream gradient, a buffer with slack (surplus rows: zero gradient)
    slack = torch.cat((nodes.detach(), torch.full((5, V), float("nan"), dtype=dtype, device=DEV))).requires_grad_(True)
    loss = rnnt_amd.rnnt_loss_compact(slack, lab, ald, lld, T, reduction="mean")
    dense = acts.detach().clone().requires_grad_(True)
    ref_loss = rnnt_amd.rnnt_loss(dense, lab, ald, lld, reduction="mean", check_lengths=False)
    assert same(loss, ref_loss, exact), what
    (loss * 0.5).backward()
    (ref_loss * 0.5).backward()
    assert same(slack.grad[:n], dense.grad.view(-1, V).index_select(0, idx), exact), what
    assert float(slack.grad[n:].float().abs().sum()) == 0.0


def test_compact_loss_buffer_smaller_than_the_lattices_and_misfits():
    B, T, U, V = 4, 50, 8, 640
    acts_np, labels, al, ll = make_case(B, T, U, V, seed=5, ragged=True)
    acts = torch.from_numpy(acts_np).to(DEV)
    lab, ald, lld = (torch.from_numpy(x).to(DEV) for x in (labels, al, ll))
    rows = host_rowmap(al, ll, T, U + 1)
    n = len(rows)
    nodes = acts.view(-1, V)[torch.from_numpy(rows).to(DEV)].contiguous()
    full = rnnt_amd.rnnt_loss_compact(nodes, lab, ald, lld, T, reduction="none")
    assert bool(torch.isfinite(full).all())
    cut = nodes[: n - 7].contiguous()                              # cuts into the last utterance only
    assert n - 7 > n - int(al[-1]) * (int(ll[-1]) + 1)
    costs = rnnt_amd.rnnt_loss_compact(cut, lab, ald, lld, T, reduction="none")
    assert torch.equal(costs[:-1], full[:-1]) and bool(torch.isnan(costs[-1]))
    # an utterance whose lengths do not fit has no nodes: NaN cost, the others unchanged
    bad = al.copy()
    bad[1] = T + 2
    rows_bad = host_rowmap(bad, ll, T, U + 1)
    assert len(rows_bad) == rnnt_amd.count_lattice_nodes(bad, ll, T, U + 1) < rnnt_amd.count_lattice_nodes(bad, ll)
    nodes_bad = acts.view(-1, V)[torch.from_numpy(rows_bad).to(DEV)].contiguous().requires_grad_(True)
    costs = rnnt_amd.rnnt_loss_compact(nodes_bad, lab, torch.from_numpy(bad).to(DEV), lld, T, reduction="none",
                                       exact_rows=True)
    c = costs.detach().cpu().numpy()
    assert np.isnan(c[1]) and np.array_equal(c[[0, 2, 3]], full.cpu().numpy()[[0, 2, 3]])
    costs.backward(torch.tensor([1.0, 0.0, 1.0, 1.0], device=DEV))
    assert bool(torch.isfinite(nodes_bad.grad).all())


def test_compact_loss_argument_checks():
    acts = torch.randn(10, 40, device=DEV)
    lab = torch.ones(1, 1, dtype=torch.int32, device=DEV)
    al = torch.tensor([5], dtype=torch.int32, device=DEV)
    ll = torch.tensor([1], dtype=torch.int32, device=DEV)
    assert rnnt_amd.rnnt_loss_compact(acts, lab, al, ll, 5).shape == (1,)
    with pytest.raises(RuntimeError):
        rnnt_amd.rnnt_loss_compact(acts.cpu(), lab, al, ll, 5)
    with pytest.raises(ValueError):
        rnnt_amd.rnnt_loss_compact(acts.view(1, 10, 40), lab, al, ll, 5)
    with pytest.raises(TypeError):
        rnnt_amd.rnnt_loss_compact(acts, lab.long(), al, ll, 5)
    with pytest.raises(ValueError):
        rnnt_amd.rnnt_loss_compact(acts, lab, al, ll, 5, blank=40)
    with pytest.raises(RuntimeError):
        rnnt_amd.compact_rowmap(al.cpu(), ll.cpu(), 5, 2, 10)
    with pytest.raises(TypeError):
        rnnt_amd.compact_rowmap(al.long(), ll.long(), 5, 2, 10)


@pytest.mark.parametrize("dtype,H", [(torch.float32, 512), (torch.bfloat16, 320)])
def test_transducer_loss_takes_the_compacted_route_for_library_projections(dtype, H):
    """JointNet.transducer_loss with the lengths on the host, as the reference has them: fp32 states (the reference's
    precision) and the shapes the fused kernels leave to the library gather the nodes' hidden vectors, project only
    those and run rnnt_loss_compact.  Same loss and parameter / state gradients as the route over the padded tensor
    (lengths on the device), up to the library GEMM's summation order on a different number of rows."""
    B, T, U, V = 6, 70, 11, 1304
    torch.manual_seed(1)
    joint = rnnt_amd.JointNet(640, H, V).to(DEV).to(dtype)
    _, labels, al, ll = make_case(B, T, U, V, seed=3, ragged=True)
    lab = torch.from_numpy(labels).to(DEV)
    enc0 = torch.randn(B, T, 320, device=DEV).to(dtype)
    dec0 = torch.randn(B, U + 1, 320, device=DEV).to(dtype)
    out = []
    for lens in ((torch.from_numpy(al), torch.from_numpy(ll)), (torch.from_numpy(al).to(DEV), torch.from_numpy(ll).to(DEV))):
        joint.zero_grad(set_to_none=True)
        enc, dec = enc0.clone().requires_grad_(True), dec0.clone().requires_grad_(True)
        loss = joint.transducer_loss(enc, dec, lab, *lens)
        loss.backward()
        out.append((loss.detach(), enc.grad, dec.grad, *[p.grad.clone() for p in joint.parameters()]))
    tol = 2e-5 if dtype is torch.float32 else 2e-2
    torch.testing.assert_close(out[0][0], out[1][0], rtol=tol, atol=0)
    for a, b in zip(out[0][1:], out[1][1:]):
        scale = float(b.float().abs().max())
        assert float((a.float() - b.float()).abs().max()) <= tol * max(scale, 1e-6)
