"""GPU parity for the fused output projection + loss: rnnt_joint_loss_forward /
rnnt_amd.fused_joint_loss through the C ABI.

The operator is checked in its two halves, with the intermediate (the logits it stores) exposed:

* the logits against the CPU restatement of the reference's ``Linear(inner -> vocab)``
  (/root/reference/src/rnnt/model.py:17,59; the last ``F.linear`` of oracle/joint_oracle.py) evaluated in
  float64 on the SAME 16-bit inputs: the stored value must be the float64 answer rounded to the dtype,
  give or take one unit in the last place where fp32 accumulation lands on the other side of a rounding
  boundary (>= 99.9 % exact, every element within 0.55 spacings of the float64 value);
* costs and gradients against the CPU loss oracle (oracle/rnnt_oracle.c) run on the logits THE KERNEL
  STORED, at the loss's own tolerances (north_star): costs <= 1e-5 relative, gradients within
  1e-4 + one rounding step of the dtype.

The loss oracle is PARITY UNPINNED (oracle/__init__.py); the joint's restatement is pinned by tests/golden/ref_joint_*.npz.
"""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from cases import make_case

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
EPS = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}


def make_inputs(B, T, U, V, H, dtype, seed, ragged=True):
    g = torch.Generator().manual_seed(seed)
    hidden = torch.tanh(torch.randn(B, T, U + 1, H, generator=g) * 1.5).to(dtype)
    weight = (torch.randn(V, H, generator=g) * (2.0 / H ** 0.5)).to(dtype)
    bias = (torch.randn(V, generator=g) * 0.5).to(dtype)
    _, labels, al, ll = make_case(B, T, U, max(V, 2), seed=seed, ragged=ragged)
    return hidden, weight, bias, labels, al, ll


def forward_c_abi(hidden, weight, bias, labels, al, ll, reduction=0, ld=None):
    """rnnt_joint_loss_forward as a C caller would use it; returns (status, logits_out, costs, workspace).
    ``ld``: rnnt_joint_loss_forward_strided with that row stride; logits_out is then [B, T, U+1, ld], NaN-filled
    before the call (columns V .. ld-1 must come back untouched)."""
    from rnnt_amd import _lib
    from rnnt_amd.loss import _DTYPES

    lib = _lib.load()
    B, T, U1, H = hidden.shape
    V = weight.shape[0]
    h, w = hidden.to(DEV).contiguous(), weight.to(DEV).contiguous()
    b32 = bias.to(DEV).float().contiguous()
    lab, ald, lld = (torch.from_numpy(x).to(DEV) for x in (labels, al, ll))
    logits = torch.full((B, T, U1, V if ld is None else ld), float("nan"), dtype=hidden.dtype, device=DEV)
    costs = torch.empty(B, dtype=torch.float32, device=DEV)
    ws = torch.empty(_lib.workspace_bytes(T, U1, B), dtype=torch.uint8, device=DEV)
    opts = _lib.RnntOptions(torch.cuda.current_stream().cuda_stream, 0, T, U1, _DTYPES[hidden.dtype], None)
    if ld is None:
        st = lib.rnnt_joint_loss_forward(h.data_ptr(), w.data_ptr(), b32.data_ptr(), logits.data_ptr(), lab.data_ptr(),
                                         ald.data_ptr(), lld.data_ptr(), H, V, B, costs.data_ptr(), None, reduction,
                                         ws.data_ptr(), opts)
    else:
        st = lib.rnnt_joint_loss_forward_strided(h.data_ptr(), w.data_ptr(), b32.data_ptr(), logits.data_ptr(), ld,
                                                 lab.data_ptr(), ald.data_ptr(), lld.data_ptr(), H, V, B,
                                                 costs.data_ptr(), None, reduction, ws.data_ptr(), opts)
    torch.cuda.synchronize()
    return st, logits, costs, ws


def check_logits(logits, hidden, weight, bias, min_exact=0.999):
    ref64 = torch.nn.functional.linear(hidden.double(), weight.double(), bias.double())   # model.py:17,59
    ref = ref64.to(logits.dtype)
    got = logits.cpu()
    assert bool(torch.isfinite(got.float()).all()), "an element of logits_out was not written"
    exact = (got == ref).float().mean().item()
    assert exact >= min_exact, exact
    # spacing of the dtype at the value: 2^(exponent - mantissa bits); a result on the wrong side of a rounding
    # boundary is just over half a spacing from the float64 value, anything else at most half
    mag = torch.maximum(ref64.abs(), got.double().abs()).clamp_min(2.0 ** -14)
    spacing = torch.exp2(torch.floor(torch.log2(mag))) * (2 * EPS[logits.dtype])
    # ... plus what fp32 accumulation of 128-512 products can be off by before the rounding: a few 2^-24 of
    # sum |h w| (matters where the terms cancel and the logit is tiny against them)
    acc_err = 2e-6 * torch.nn.functional.linear(hidden.double().abs(), weight.double().abs(), bias.double().abs())
    worst = float((((got.double() - ref64).abs() - acc_err).clamp_min(0) / spacing).max())
    assert worst <= 0.55, worst


@pytest.mark.parametrize("B,T,U,V,ld,H,dtype", [
    (2, 9, 4, 4232, 4352, 512, torch.bfloat16),     # what rnnt_amd.FusedJointLoss allocates: rows on 256-byte boundaries
    (3, 33, 7, 4234, 4288, 512, torch.bfloat16),    # odd vocabulary, rows on 128-byte lines
    (2, 21, 5, 5455, 5457, 512, torch.float16),     # a stride that aligns nothing
    (4, 70, 12, 1000, 1024, 128, torch.float16),    # several workgroups, the surplus rule, H = 128
    (1, 40, 6, 64, 200, 256, torch.bfloat16),
])
def test_fused_forward_with_padded_rows(B, T, U, V, ld, H, dtype):
    """rnnt_joint_loss_forward_strided: the same logits (bit for bit), costs and lattice planes as the contiguous
    call, written at the row stride; the padding of every row stays untouched."""
    hidden, weight, bias, labels, al, ll = make_inputs(B, T, U, V, H, dtype, seed=V + H + T)
    st, ref_logits, ref_costs, ref_ws = forward_c_abi(hidden, weight, bias, labels, al, ll)
    assert st == 0
    st, logits, costs, ws = forward_c_abi(hidden, weight, bias, labels, al, ll, ld=ld)
    assert st == 0
    assert torch.equal(logits[..., :V].contiguous().view(torch.int16), ref_logits.view(torch.int16))
    assert bool(torch.isnan(logits[..., V:].float()).all())
    assert torch.equal(costs, ref_costs)
    from rnnt_amd import _lib

    lib = _lib.load()
    opts = _lib.RnntOptions(None, 0, T, U + 1, 1, None)
    args = (hidden.to(DEV).data_ptr(), weight.to(DEV).data_ptr(), bias.to(DEV).float().data_ptr(), logits.data_ptr())
    tail = (torch.from_numpy(labels).to(DEV).data_ptr(), torch.from_numpy(al).to(DEV).data_ptr(),
            torch.from_numpy(ll).to(DEV).data_ptr(), H, V, B, costs.data_ptr(), None, 0, ws.data_ptr(), opts)
    assert lib.rnnt_joint_loss_forward_strided(*args, V - 1, *tail) == 2     # stride shorter than a row


@pytest.mark.parametrize("B,T,U,V,H,dtype", [
    (2, 9, 4, 4232, 512, torch.bfloat16),      # the reference's vocabulary and inner size
    (3, 33, 7, 4234, 512, torch.bfloat16),     # rows not 16-byte aligned: 16-byte stores at 2-byte aligned addresses
    (2, 21, 5, 5455, 512, torch.float16),      # config/magic_*.yaml's vocabulary, odd
    (1, 40, 6, 64, 512, torch.bfloat16),       # exactly one tile
    (2, 17, 3, 100, 256, torch.bfloat16),      # last tile overlaps its predecessor; H = 256
    (2, 50, 9, 1000, 128, torch.float16),      # H = 128
    (4, 70, 12, 1024, 512, torch.float16),     # more than one workgroup (3 640 nodes), last one partial
    (1, 1, 0, 320, 512, torch.bfloat16),       # a single node, no labels
])
def test_fused_forward_two_halves(B, T, U, V, H, dtype):
    hidden, weight, bias, labels, al, ll = make_inputs(B, T, U, V, H, dtype, seed=V + H + T)
    st, logits, costs, _ = forward_c_abi(hidden, weight, bias, labels, al, ll)
    assert st == 0
    check_logits(logits, hidden, weight, bias)
    c, _, _ = oracle.rnnt_loss_c(logits.float().cpu().numpy(), labels, al, ll, want_grads=False)
    np.testing.assert_allclose(costs.cpu().numpy(), c, rtol=1e-5, atol=0)


@pytest.mark.parametrize("B,T,U,V,H,what", [
    (50, 64, 15, 320, 128, "200 row tiles: less than a round of 256 CUs but more than half, every workgroup walks the whole vocabulary"),
    (37, 112, 15, 320, 128, "259 row tiles: one full round + 3 surplus tiles whose 5 vocabulary tiles are shared by 5 workgroups each"),
    (33, 31, 7, 4232, 256, "32 row tiles (the last one partial), each shared by 8 workgroups of 8-9 vocabulary tiles"),
    (3, 43, 5, 16390, 128, "4 row tiles x 64 shares of a 257-tile vocabulary whose last tile overlaps"),
])
def test_surplus_row_tiles_share_their_vocabulary(B, T, U, V, H, what):
    """A launch's row tiles beyond whole rounds of CUs are split along the vocabulary (rnnt_fused_merge combines the
    {max, sum} pairs): same logits, same costs, on either side of every boundary of that rule."""
    hidden, weight, bias, labels, al, ll = make_inputs(B, T, U, V, H, torch.bfloat16, seed=B + V)
    st, logits, costs, _ = forward_c_abi(hidden, weight, bias, labels, al, ll)
    assert st == 0, what
    check_logits(logits, hidden, weight, bias)
    c, _, _ = oracle.rnnt_loss_c(logits.float().cpu().numpy(), labels, al, ll, want_grads=False)
    np.testing.assert_allclose(costs.cpu().numpy(), c, rtol=1e-5, atol=0, err_msg=what)


@pytest.mark.parametrize("B,T,U,V,H,dtype", [(3, 1, 6, 518, 512, torch.float16), (5, 11, 5, 880, 128, torch.bfloat16),
                                             (7, 39, 7, 2584, 128, torch.float16)])
def test_repeated_small_launches_give_the_same_bits(B, T, U, V, H, dtype):
    """Regression: the first step of a vocabulary walk once let its last four weight-tile DMAs stay in flight past the
    barrier (no epilogue stores had been issued yet to be the youngest operations), and about one small launch in
    150 — many workgroups starting cold at once — multiplied a half-landed second tile.  300 launches each."""
    hidden, weight, bias, labels, al, ll = make_inputs(B, T, U, V, H, dtype, seed=V)
    st, first, costs0, _ = forward_c_abi(hidden, weight, bias, labels, al, ll)
    assert st == 0
    check_logits(first, hidden, weight, bias, min_exact=0.995)   # (10 878 fp16 logits: a dozen sit on rounding boundaries)
    for i in range(300):
        st, logits, costs, _ = forward_c_abi(hidden, weight, bias, labels, al, ll)
        assert st == 0 and torch.equal(logits, first) and torch.equal(costs, costs0), f"launch {i} differs"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("reduction", ["mean", "sum", "none"])
def test_fused_operator_forward_backward(dtype, reduction):
    """rnnt_amd.fused_joint_loss: loss and every gradient against the unfused composition evaluated by the
    oracles on the same 16-bit inputs (joint projection in float64, loss by the C oracle on the stored
    logits, then the chain rule through the Linear in float64)."""
    import rnnt_amd

    B, T, U, V, H = 3, 30, 6, 515 if dtype is torch.float16 else 520, 512
    hidden, weight, bias, labels, al, ll = make_inputs(B, T, U, V, H, dtype, seed=77)
    h = hidden.to(DEV).requires_grad_(True)
    w = weight.to(DEV).requires_grad_(True)
    b = bias.to(DEV).requires_grad_(True)
    lab, ald, lld = (torch.from_numpy(x).to(DEV) for x in (labels, al, ll))
    loss = rnnt_amd.fused_joint_loss(h, w, b, lab, ald, lld, reduction=reduction, min_nodes=0)   # the kernel, whatever the size
    go = 0.5 if reduction != "none" else np.linspace(0.25, 1.5, B).astype(np.float32)
    loss.backward(torch.as_tensor(go, dtype=torch.float32, device=DEV).reshape(loss.shape))

    # what the kernel stored (same call through the C ABI), then the oracle on exactly those logits
    _, logits, _, _ = forward_c_abi(hidden, weight, bias, labels, al, ll)
    lg = logits.float().cpu().numpy()
    c, g, _ = oracle.rnnt_loss_c(lg, labels, al, ll)
    ref_loss, ref_g = oracle.reduce_like_upstream(c, g, reduction, go)
    np.testing.assert_allclose(loss.detach().cpu().numpy(), ref_loss, rtol=1e-5)
    # chain rule through Linear in float64, from the dtype-rounded gradient of the logits (what K3 writes)
    g_logits = torch.from_numpy(ref_g).to(dtype).double().reshape(-1, V)
    ref_dh = (g_logits @ weight.double()).reshape(hidden.shape)
    ref_dw = g_logits.t() @ hidden.double().reshape(-1, H)
    ref_db = g_logits.sum(0)
    for name, got, ref in (("hidden", h.grad, ref_dh), ("weight", w.grad, ref_dw), ("bias", b.grad, ref_db)):
        got = got.double().cpu()
        scale = float(ref.abs().max())
        # K3's own rounding of every logit gradient (one step of the dtype each) enters these sums
        assert float((got - ref).abs().max()) <= 4 * EPS[dtype] * scale + 1e-6, name
    assert h.grad.dtype is dtype and w.grad.dtype is dtype and b.grad.dtype is dtype


def test_fused_equals_unfused_operator_on_device():
    """Same inputs through rnnt_amd.JointNet-style Linear + RNNTLoss: the two forward values agree to fp32
    summation order (the unfused path's GEMM rounds the same fp32 sums to bf16 except at rounding boundaries)."""
    import rnnt_amd

    hidden, weight, bias, labels, al, ll = make_inputs(4, 60, 10, 4232, 512, torch.bfloat16, seed=5)
    h, w, b = hidden.to(DEV), weight.to(DEV), bias.to(DEV)
    lab, ald, lld = (torch.from_numpy(x).to(DEV) for x in (labels, al, ll))
    fused = rnnt_amd.FusedJointLoss(min_nodes=0)(h, _Proj(w, b), lab, ald, lld)
    unfused = rnnt_amd.RNNTLoss()(torch.nn.functional.linear(h, w, b), lab, ald, lld)
    assert abs(float(fused) - float(unfused)) <= 2e-4 * abs(float(unfused))


def test_small_lattices_take_the_unfused_composition():
    """Below min_nodes the operator is Linear + RNNTLoss on the device (the faster forward there): same loss and
    gradients as the fused kernel within the 16-bit rounding of the logits, and bitwise the composition itself."""
    import rnnt_amd

    hidden, weight, bias, labels, al, ll = make_inputs(2, 20, 4, 300, 256, torch.bfloat16, seed=21)
    lab, ald, lld = (torch.from_numpy(x).to(DEV) for x in (labels, al, ll))
    outs = []
    for min_nodes in (rnnt_amd.fused.DEFAULT_MIN_NODES, 0):
        h, w, b = (x.to(DEV).requires_grad_(True) for x in (hidden, weight, bias))
        loss = rnnt_amd.FusedJointLoss(min_nodes=min_nodes)(h, _Proj(w, b), lab, ald, lld)
        loss.backward()
        outs.append((loss.detach(), h.grad, w.grad, b.grad))
    assert 2 * 20 * 5 < rnnt_amd.fused.DEFAULT_MIN_NODES
    h, w, b = (x.to(DEV).requires_grad_(True) for x in (hidden, weight, bias))
    # (the composition as the kernel forms it: fp32 accumulation, fp32 bias, one rounding to the 16-bit type)
    ref = rnnt_amd.RNNTLoss(check_lengths=False)(
        torch.nn.functional.linear(h.float(), w.float(), b.float()).to(h.dtype), lab, ald, lld)
    ref.backward()
    assert torch.equal(outs[0][0], ref.detach()) and torch.equal(outs[0][1], h.grad) and torch.equal(outs[0][2], w.grad)
    assert abs(float(outs[0][0]) - float(outs[1][0])) <= 2e-4 * abs(float(outs[1][0]))
    for a, c in zip(outs[0][1:], outs[1][1:]):
        assert float((a.float() - c.float()).abs().max()) <= 4 * EPS[torch.bfloat16] * float(c.float().abs().max()) + 1e-6


def test_fp32_bias_gives_the_same_logits_on_both_sides_of_min_nodes():
    """An fp32 bias parameter is added in fp32 by the kernel and by the small-lattice composition alike: the two
    paths see the same logits (loss within fp32 summation order), not logits that differ by the bias's rounding."""
    import rnnt_amd

    hidden, weight, bias, labels, al, ll = make_inputs(2, 20, 4, 300, 256, torch.bfloat16, seed=23)
    bias32 = (bias.float() * 3.0 + 0.001234)        # not representable in bf16
    lab, ald, lld = (torch.from_numpy(x).to(DEV) for x in (labels, al, ll))
    a = rnnt_amd.fused_joint_loss(hidden.to(DEV), weight.to(DEV), bias32.to(DEV), lab, ald, lld, min_nodes=0)
    b = rnnt_amd.fused_joint_loss(hidden.to(DEV), weight.to(DEV), bias32.to(DEV), lab, ald, lld, min_nodes=10 ** 9)
    assert abs(float(a) - float(b)) <= 2e-4 * abs(float(a))   # (logits on a rounding boundary may differ by one step)


class _Proj:
    """Stands in for nn.Linear: FusedJointLoss only reads .weight and .bias."""

    def __init__(self, weight, bias):
        self.weight, self.bias = weight, bias


def test_fused_nonfinite_inputs_give_nan_cost_not_an_error():
    hidden, weight, bias, labels, al, ll = make_inputs(2, 12, 3, 256, 512, torch.bfloat16, seed=9, ragged=False)
    hidden[1, 3, 2, 17] = float("inf")       # one hidden unit of one node of utterance 1
    st, logits, costs, _ = forward_c_abi(hidden, weight, bias, labels, al, ll)
    assert st == 0
    c = costs.cpu().numpy()
    assert np.isfinite(c[0]) and np.isnan(c[1])   # bin/train.py:36-41 must SEE the NaN


def test_fused_padding_nodes_do_not_matter_and_costs_are_bitwise_repeatable():
    hidden, weight, bias, labels, al, ll = make_inputs(3, 25, 6, 777, 512, torch.float16, seed=13)
    _, _, costs1, _ = forward_c_abi(hidden, weight, bias, labels, al, ll)
    poisoned = hidden.clone()
    for bb in range(3):
        poisoned[bb, al[bb]:] = float("nan")
        poisoned[bb, :, ll[bb] + 1:] = float("nan")
    _, _, costs2, _ = forward_c_abi(poisoned, weight, bias, labels, al, ll)
    assert torch.equal(costs1, costs2) and bool(torch.isfinite(costs1).all())


def test_fused_argument_checks():
    import rnnt_amd
    from rnnt_amd import _lib

    hidden, weight, bias, labels, al, ll = make_inputs(1, 4, 2, 128, 512, torch.bfloat16, seed=1)
    lab, ald, lld = (torch.from_numpy(x).to(DEV) for x in (labels, al, ll))
    h, w, b = hidden.to(DEV), weight.to(DEV), bias.to(DEV)
    with pytest.raises(TypeError, match="bfloat16 or float16"):
        rnnt_amd.fused_joint_loss(h.float(), w.float(), b.float(), lab, ald, lld)
    with pytest.raises(ValueError, match="inner sizes"):
        rnnt_amd.fused_joint_loss(h[..., :320].contiguous(), w[:, :320].contiguous(), b, lab, ald, lld)
    with pytest.raises(ValueError, match="needs the projection's bias"):
        rnnt_amd.fused_joint_loss(h, w, None, lab, ald, lld)
    with pytest.raises(ValueError, match="at least 64"):
        rnnt_amd.fused_joint_loss(h, w[:40].contiguous(), b[:40].contiguous(), lab, ald, lld)
    with pytest.raises(RuntimeError, match="no CPU path"):
        rnnt_amd.fused_joint_loss(hidden, weight, bias, torch.from_numpy(labels), torch.from_numpy(al),
                                  torch.from_numpy(ll))
    # the C ABI refuses the same things with a status, before any launch
    lib = _lib.load()
    opts = _lib.RnntOptions(None, 0, 4, 3, _lib.DTYPE_F32, None)
    args = [h.data_ptr(), w.data_ptr(), b.float().data_ptr(), h.data_ptr(), lab.data_ptr(), ald.data_ptr(),
            lld.data_ptr()]
    costs = torch.empty(1, device=DEV)
    ws = torch.empty(_lib.workspace_bytes(4, 3, 1), dtype=torch.uint8, device=DEV)
    assert lib.rnnt_joint_loss_forward(*args, 512, 128, 1, costs.data_ptr(), None, 0, ws.data_ptr(), opts) == 2  # fp32
    opts = _lib.RnntOptions(None, 0, 4, 3, _lib.DTYPE_BF16, None)
    assert lib.rnnt_joint_loss_forward(*args, 320, 128, 1, costs.data_ptr(), None, 0, ws.data_ptr(), opts) == 2  # H
    assert lib.rnnt_joint_loss_forward(*args, 512, 40, 1, costs.data_ptr(), None, 0, ws.data_ptr(), opts) == 2   # V < 64


def test_fused_config2_shape_full_size():
    """B=32 T=400 U=40 V=4232 H=512 bf16 (BASELINE config 2's lattice): 524 800 nodes, 2 050 workgroups, the
    last one partial.  Costs against the unfused loss on the stored logits (same K2, statistics from two
    different kernels), a sampled block of logits against float64, every element written."""
    import rnnt_amd

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
    st, logits, costs, _ = forward_c_abi(hidden, weight, bias, labels, al, ll)
    assert st == 0 and bool(torch.isfinite(logits.view(-1, V)[::97].float()).all())
    assert not bool(torch.isnan(logits.float().sum(-1)).any())          # every row fully written
    lab, ald, lld = (torch.from_numpy(x).to(DEV) for x in (labels, al, ll))
    ref_costs = rnnt_amd.rnnt_loss(logits, lab, ald, lld, reduction="none")
    assert float(((costs - ref_costs).abs() / ref_costs.abs()).max()) <= 1e-6
    rows = slice(123 * 41 * 7, 123 * 41 * 7 + 300)
    check_logits(logits.view(-1, V)[rows].cpu(), hidden.view(-1, H)[rows].cpu(), weight.cpu(), bias.cpu())
    b = int(np.argmin(al.astype(np.int64) * (ll + 1)))
    Tb, Ub = int(al[b]), int(ll[b])
    c, _, _ = oracle.rnnt_loss_c(logits[b:b + 1, :Tb, :Ub + 1].float().cpu().numpy(), labels[b:b + 1, :Ub], [Tb], [Ub],
                                 want_grads=False)
    assert abs(float(costs[b]) - c[0]) <= 1e-5 * abs(c[0])


def test_module_under_autocast_with_fp32_projection():
    """`FusedJointLoss` inside torch.autocast with the reference's fp32 `Linear(512 -> V)`: the weight is cast to the
    hidden layer's 16-bit type, its gradient comes back in fp32 and equals the bf16-weight call's (rounded once)."""
    import rnnt_amd

    B, T, U, V, H = 3, 50, 9, 333, 256
    torch.manual_seed(4)
    proj = torch.nn.Linear(H, V).to(DEV)
    hidden = torch.tanh(torch.randn(B, T, U + 1, H, device=DEV)).to(torch.bfloat16).requires_grad_(True)
    _, labels, al, ll = make_case(B, T, U, V, seed=1, ragged=True)
    lab, ald, lld = (torch.from_numpy(x).to(DEV) for x in (labels, al, ll))
    crit = rnnt_amd.FusedJointLoss(min_nodes=0)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = crit(hidden, proj, lab, ald, lld)
    loss.backward()
    assert proj.weight.grad.dtype is torch.float32 and proj.bias.grad.dtype is torch.float32
    w16 = proj.weight.detach().to(torch.bfloat16).requires_grad_(True)
    h2 = hidden.detach().clone().requires_grad_(True)
    loss2 = rnnt_amd.fused_joint_loss(h2, w16, proj.bias.detach(), lab, ald, lld, min_nodes=0)
    loss2.backward()
    assert torch.equal(loss.detach(), loss2.detach()) and torch.equal(hidden.grad, h2.grad)
    assert torch.equal(proj.weight.grad, w16.grad.float())
