"""The compiled PyTorch binding (rnnt_amd/csrc/torch_binding.cpp, torch.ops.rnnt_amd.*) against the ctypes binding
of the same C ABI: bit-identical losses and gradients (same kernels, same launch arguments), the raw
forward/backward ops, non-default streams and devices' error behaviour.  Parity with the oracle is what every other
GPU test checks THROUGH this binding (it is RNNTLoss's default); test_default_binding_is_the_compiled_one pins that."""
import numpy as np
import pytest
import torch

import oracle
from cases import make_case

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def d(x):
    return torch.from_numpy(x).to(DEV)


def run(binding, acts, labels, al, ll, reduction, go, dtype=torch.float32, blank=0):
    import rnnt_amd

    a = d(acts).to(dtype).requires_grad_(True)
    loss = rnnt_amd.RNNTLoss(blank=blank, reduction=reduction, binding=binding)(a, d(labels), d(al), d(ll))
    loss.backward(go.to(DEV))
    return loss.detach(), a.grad


def test_default_binding_is_the_compiled_one():
    import rnnt_amd
    from rnnt_amd import _torch_binding, loss

    assert loss.get_binding() == "torch"
    assert rnnt_amd.RNNTLoss().binding is None
    ops = _torch_binding.load()
    assert ops.abi_version() == rnnt_amd._lib.ABI_VERSION
    with open("/proc/self/maps") as fh:
        maps = fh.read()
    assert "librnnt_amd_torch.so" in maps and maps.count("librnnt_mi355.so") >= 1
    # one copy of the kernels library in the process: the binding resolved to the copy ctypes loaded
    paths = {line.split()[-1] for line in maps.splitlines() if "librnnt_mi355" in line}
    assert len(paths) == 1, paths


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("reduction", ["mean", "sum", "none"])
@pytest.mark.parametrize("B,T,U,V", [(3, 9, 4, 29), (2, 12, 5, 520), (4, 20, 7, 4232), (1, 3, 0, 17412)])
def test_bindings_agree_bit_for_bit(B, T, U, V, reduction, dtype):
    acts, labels, al, ll = make_case(B, T, U, V, seed=B + T + V, ragged=True)
    go = torch.linspace(0.25, 1.5, B) if reduction == "none" else torch.tensor([0.5])
    l0, g0 = run("ctypes", acts, labels, al, ll, reduction, go, dtype)
    l1, g1 = run("torch", acts, labels, al, ll, reduction, go, dtype)
    assert l1.shape == l0.shape and l1.dtype is torch.float32
    assert torch.equal(l0, l1) and torch.equal(g0, g1)
    assert g1.dtype is dtype and g1.is_contiguous()


def test_compiled_binding_against_oracle_with_nonzero_blank():
    acts, labels, al, ll = make_case(3, 10, 4, 37, seed=5, ragged=True, blank=2)
    loss, grads = run("torch", acts, labels, al, ll, "mean", torch.tensor([1.0]), blank=2)
    c, g, _ = oracle.rnnt_loss_c(acts, labels, al, ll, blank=2)
    ref_loss, ref_g = oracle.reduce_like_upstream(c, g, "mean", 1.0)
    np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, rtol=1e-5)
    assert np.abs(grads.cpu().numpy() - ref_g).max() <= 1e-4


def test_raw_forward_backward_ops():
    from rnnt_amd import _lib, _torch_binding

    ops = _torch_binding.load()
    acts, labels, al, ll = make_case(4, 11, 5, 300, seed=3, ragged=True)
    a, lab, ald, lld = d(acts), d(labels), d(al), d(ll)
    loss, costs, ws = ops.rnnt_forward(a, lab, ald, lld, 0, _lib.REDUCTION_NONE)
    assert ws.dtype is torch.uint8 and ws.numel() == _lib.workspace_bytes(11, 6, 4) and ws.data_ptr() % 256 == 0
    assert torch.equal(loss, costs)
    c, g, _ = oracle.rnnt_loss_c(acts, labels, al, ll)
    np.testing.assert_allclose(costs.cpu().numpy(), c, rtol=1e-5)
    go = torch.tensor([1.0, 0.5, 2.0, 0.0], device=DEV)
    grads = ops.rnnt_backward(a, lab, ald, lld, go, ws, 0, _lib.REDUCTION_NONE)
    assert np.abs(grads.cpu().numpy() - g * go.cpu().numpy()[:, None, None, None]).max() <= 1e-4
    mean, _, ws2 = ops.rnnt_forward(a, lab, ald, lld, 0, _lib.REDUCTION_MEAN)
    assert mean.shape == (1,) and abs(mean.item() - c.mean()) <= 1e-5 * abs(c.mean())
    gm = ops.rnnt_backward(a, lab, ald, lld, torch.ones(1, device=DEV), ws2, 0, _lib.REDUCTION_MEAN)
    assert np.abs(gm.cpu().numpy() - g / 4).max() <= 1e-4
    with pytest.raises(RuntimeError, match="workspace"):
        ops.rnnt_backward(a, lab, ald, lld, go, ws[:100], 0, _lib.REDUCTION_NONE)
    with pytest.raises(RuntimeError, match="one element per utterance"):
        ops.rnnt_backward(a, lab, ald, lld, go[:2], ws, 0, _lib.REDUCTION_NONE)


def test_runs_on_the_current_stream_and_allocates_from_torch():
    import rnnt_amd

    acts, labels, al, ll = make_case(2, 30, 6, 4232, seed=9, ragged=False)
    ref_l, ref_g = run("torch", acts, labels, al, ll, "mean", torch.tensor([1.0]))
    side = torch.cuda.Stream()
    a = d(acts).requires_grad_(True)
    lab, ald, lld = d(labels), d(al), d(ll)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        torch.cuda._sleep(20_000_000)            # the side stream is busy: a launch on another stream would overtake it
        a2 = a * 1.0                             # produced on the side stream, after the sleep
        loss = rnnt_amd.RNNTLoss()(a2, lab, ald, lld)
        loss.backward()
    side.synchronize()
    assert torch.equal(loss.detach(), ref_l) and torch.equal(a.grad, ref_g)
    before = torch.cuda.memory_allocated()
    loss = rnnt_amd.RNNTLoss(check_lengths=False)(a, lab, ald, lld)
    held = torch.cuda.memory_allocated() - before
    assert held >= rnnt_amd._lib.workspace_bytes(30, 7, 2)   # the lattice workspace is a torch allocation kept for backward
    del loss
    assert torch.cuda.memory_allocated() - before < 1024


def test_argument_errors_keep_their_types():
    import rnnt_amd
    from rnnt_amd import _torch_binding

    ops = _torch_binding.load()
    a = torch.zeros(1, 2, 3, 5, device=DEV)
    lab = torch.ones(1, 2, dtype=torch.int32, device=DEV)
    one = torch.ones(1, dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.rnnt_loss(a.cpu(), lab.cpu(), one.cpu(), one.cpu(), 0, 1)
    with pytest.raises(TypeError):
        ops.rnnt_loss(a.double(), lab, one, one, 0, 1)
    with pytest.raises(TypeError):
        ops.rnnt_loss(a, lab.long(), one, one, 0, 1)
    with pytest.raises(ValueError):
        ops.rnnt_loss(a, lab[:, :1], one, one, 0, 1)
    with pytest.raises(ValueError):
        ops.rnnt_loss(a, lab, one, one, 0, 7)
    with pytest.raises(RuntimeError, match="invalid value"):      # blank outside the vocabulary: the C ABI's status
        ops.rnnt_loss(a, lab, one, one, 9, 1)
    with pytest.raises(ValueError, match="binding"):
        rnnt_amd.RNNTLoss(binding="jit")
    # the operator's own checks come first and are upstream's
    with pytest.raises(ValueError, match="Input length mismatch"):
        rnnt_amd.RNNTLoss()(a, lab, one * 3, one)


def test_double_use_of_one_graph_and_no_grad():
    import rnnt_amd

    acts, labels, al, ll = make_case(2, 8, 3, 64, seed=1, ragged=True)
    a = d(acts).requires_grad_(True)
    crit = rnnt_amd.RNNTLoss(reduction="sum")
    loss = crit(a, d(labels), d(al), d(ll))
    g1, = torch.autograd.grad(loss, a, retain_graph=True)
    g2, = torch.autograd.grad(loss, a, grad_outputs=torch.full((1,), 2.0, device=DEV))
    assert torch.equal(g2, g1 * 2)
    with torch.no_grad():
        l2 = crit(a, d(labels), d(al), d(ll))
    assert not l2.requires_grad and torch.equal(l2, loss.detach())
