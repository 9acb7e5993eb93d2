"""GPU parity for the joint-network widening: rnnt_amd.joint_hidden / JointNet
through the C ABI against the CPU restatement of the reference's JointNet (oracle/joint_oracle.py).

Tolerances, fp32: hidden <= 1e-6 absolute (|tanh| <= 1; the GPU uses the device libm tanhf, the CPU
torch's), logits <= 2e-5 absolute at |logit| ~ 10, gradients <= 1e-5 relative to the largest
reference entry of the same tensor (sums of up to T*(U+1)*B fp32 terms in a different order).
bf16/fp16: the result must equal the float64 answer rounded to the dtype within 1 ulp-ish
(2^-8 / 2^-11 relative).  Pinned to the reference's own JointNet by tests/golden/ref_joint_*.npz
(test_jointnet_module_against_reference_fixture; tests/golden/make_ref_fixtures.py wrote them).
"""
import numpy as np
import pytest
import torch

import oracle
from oracle.joint_oracle import joint_logits

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def rel_to_max(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("B,T,U1,H", [(2, 5, 4, 512), (1, 33, 3, 320), (3, 16, 1, 7), (2, 17, 5, 1030),
                                       (1, 1, 1, 4), (2, 40, 6, 2049)])
def test_joint_hidden_forward_backward_fp32(B, T, U1, H):
    import rnnt_amd

    g = torch.Generator().manual_seed(B * 1000 + T * 10 + H)
    ep = torch.randn(B, T, H, generator=g) * 1.5
    dp = torch.randn(B, U1, H, generator=g) * 1.5
    gh = torch.randn(B, T, U1, H, generator=g)
    ep64, dp64 = ep.double().requires_grad_(True), dp.double().requires_grad_(True)
    ref = torch.tanh(ep64[:, :, None, :] + dp64[:, None, :, :])
    ref.backward(gh.double())

    e, d = ep.to(DEV).requires_grad_(True), dp.to(DEV).requires_grad_(True)
    h = rnnt_amd.joint_hidden(e, d)
    assert h.shape == (B, T, U1, H) and h.dtype is torch.float32 and h.is_contiguous()
    h.backward(gh.to(DEV))
    assert (h.detach().cpu().double() - ref.detach()).abs().max() <= 1e-6
    assert rel_to_max(e.grad.cpu().double(), ep64.grad) <= 1e-5
    assert rel_to_max(d.grad.cpu().double(), dp64.grad) <= 1e-5


def test_joint_hidden_backward_is_bitwise_reproducible_and_ordered():
    import rnnt_amd

    g = torch.Generator().manual_seed(5)
    e = (torch.randn(2, 70, 512, generator=g)).to(DEV).requires_grad_(True)
    d = (torch.randn(2, 9, 512, generator=g)).to(DEV).requires_grad_(True)
    gh = torch.randn(2, 70, 9, 512, generator=g).to(DEV)
    outs = []
    for _ in range(2):
        e.grad = d.grad = None
        rnnt_amd.joint_hidden(e, d).backward(gh)
        outs.append((e.grad.clone(), d.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("dtype,eps", [(torch.bfloat16, 2.0 ** -8), (torch.float16, 2.0 ** -11)])
def test_joint_hidden_half_precision(dtype, eps):
    import rnnt_amd

    g = torch.Generator().manual_seed(11)
    ep = (torch.randn(2, 19, 512, generator=g)).to(dtype)
    dp = (torch.randn(2, 4, 512, generator=g)).to(dtype)
    gh = torch.randn(2, 19, 4, 512, generator=g).to(dtype)
    ep64, dp64 = ep.double().requires_grad_(True), dp.double().requires_grad_(True)
    ref = torch.tanh(ep64[:, :, None, :] + dp64[:, None, :, :])
    e, d = ep.to(DEV).requires_grad_(True), dp.to(DEV).requires_grad_(True)
    h = rnnt_amd.joint_hidden(e, d)
    assert h.dtype is dtype
    h.backward(gh.to(DEV))
    assert (h.detach().cpu().double() - ref.detach()).abs().max() <= eps
    # the kernel multiplies by 1 - h^2 with h as STORED (rounded to the dtype), as autograd through a
    # half-precision Tanh does; the stored value may sit one rounding step from torch's in near-ties
    hq = h.detach().cpu().double()
    gpre = gh.double() * (1 - hq * hq)
    ge, gd = gpre.sum(2), gpre.sum(1)
    assert ((e.grad.cpu().double() - ge).abs() <= 2 * eps * ge.abs() + 1e-6).all()
    assert ((d.grad.cpu().double() - gd).abs() <= 2 * eps * gd.abs() + 1e-6).all()


@pytest.mark.parametrize("joint", ["concat", "sum"])
def test_jointnet_module_against_reference_restatement(joint):
    """rnnt_amd.JointNet == the reference's JointNet on the same weights: logits and every gradient."""
    import rnnt_amd

    torch.manual_seed(21)
    B, T, U1, He, Hd, H, V = 2, 11, 5, 40, (40 if joint == "sum" else 24), 64, 37
    nin = He if joint == "sum" else He + Hd
    net = rnnt_amd.JointNet(nin, H, V, joint=joint)
    enc, dec = torch.randn(B, T, He), torch.randn(B, U1, Hd)
    gl = torch.randn(B, T, U1, V)

    p64 = [p.detach().double().requires_grad_(True) for p in net.parameters()]
    e64, d64 = enc.double().requires_grad_(True), dec.double().requires_grad_(True)
    ref = joint_logits(e64, d64, *p64, joint=joint)
    ref.backward(gl.double())

    net = net.to(DEV)
    e, d = enc.to(DEV).requires_grad_(True), dec.to(DEV).requires_grad_(True)
    out = net(e, d)
    out.backward(gl.to(DEV))
    assert (out.detach().cpu().double() - ref.detach()).abs().max() <= 2e-5
    assert rel_to_max(e.grad.cpu().double(), e64.grad) <= 1e-5
    assert rel_to_max(d.grad.cpu().double(), d64.grad) <= 1e-5
    for p, q in zip(net.parameters(), p64):
        assert rel_to_max(p.grad.cpu().double(), q.grad) <= 1e-5
    # pre-unsqueezed 4-D states (model.py:46-50) take the same path
    out4 = net(e.detach().unsqueeze(2), d.detach().unsqueeze(1))
    assert torch.equal(out4, out.detach())


@pytest.mark.parametrize("kind", ["concat", "sum"])
def test_jointnet_module_against_reference_fixture(kind):
    """rnnt_amd.JointNet with the fixture's parameters against what the REFERENCE'S OWN JointNet
    (model.py:20-63, float64) produced for the same states: logits and every gradient."""
    import rnnt_amd
    from ref_fixtures import joint_case

    c = joint_case(kind)
    nin, inner, V = c["params"][0].shape[1], c["params"][0].shape[0], c["params"][2].shape[0]
    net = rnnt_amd.JointNet(nin, inner, V, joint=kind)
    net.load_state_dict({n: p.float() for n, p in zip(c["names"], c["params"])})
    net = net.to(DEV)
    e = c["enc"].float().to(DEV).requires_grad_(True)
    d = c["dec"].float().to(DEV).requires_grad_(True)
    out = net(e, d)
    (out * c["go"].float().to(DEV)).sum().backward()
    assert (out.detach().cpu().double() - c["logits"]).abs().max() <= 2e-5
    assert rel_to_max(e.grad.cpu().double(), c["grad_enc"]) <= 1e-5
    assert rel_to_max(d.grad.cpu().double(), c["grad_dec"]) <= 1e-5
    got = dict(net.named_parameters())
    for n, g in zip(c["names"], c["grads"]):
        assert rel_to_max(got[n].grad.cpu().double(), g) <= 1e-5, n


def test_joint_then_loss_end_to_end_against_oracles():
    """The reference's two lines model.py:115-116 — joint, then RNNTLoss — on the MI355X path against
    the CPU restatements of both, down to the gradient of the encoder and decoder states."""
    import rnnt_amd

    torch.manual_seed(4)
    B, T, U, He, Hd, H, V = 3, 12, 4, 16, 8, 32, 29
    net = rnnt_amd.JointNet(He + Hd, H, V)
    enc, dec = torch.randn(B, T, He), torch.randn(B, U + 1, Hd)
    rng = np.random.default_rng(9)
    labels = rng.integers(1, V, size=(B, U)).astype(np.int32)
    al = np.array([T, 7, 10], np.int32)
    ll = np.array([U, 2, 4], np.int32)

    params = [p.detach().clone().requires_grad_(True) for p in net.parameters()]
    e32, d32 = enc.clone().requires_grad_(True), dec.clone().requires_grad_(True)
    logits = joint_logits(e32, d32, *params)
    c, g, _ = oracle.rnnt_loss_c(logits.detach().numpy(), labels, al, ll)
    ref_loss, ref_g = oracle.reduce_like_upstream(c, g, "mean", 1.0)
    logits.backward(torch.from_numpy(ref_g))

    net = net.to(DEV)
    e, d = enc.to(DEV).requires_grad_(True), dec.to(DEV).requires_grad_(True)
    loss = rnnt_amd.RNNTLoss()(net(e, d), torch.from_numpy(labels).to(DEV), torch.from_numpy(al).to(DEV),
                               torch.from_numpy(ll).to(DEV))
    loss.backward()
    assert abs(loss.item() - ref_loss[0]) <= 1e-5 * abs(ref_loss[0])
    assert rel_to_max(e.grad.cpu(), e32.grad) <= 1e-4
    assert rel_to_max(d.grad.cpu(), d32.grad) <= 1e-4
    for p, q in zip(net.parameters(), params):
        assert rel_to_max(p.grad.cpu(), q.grad) <= 1e-4


def test_jointnet_transducer_loss_is_the_two_reference_lines_in_one_call():
    """JointNet.transducer_loss(enc, dec, tokens, out_len, tok_len) = model.py:115-116 in one call.  fp32: the same
    kernels as the two calls, so the same bits.  bf16 at the reference's sizes (He = Hd = 320, inner 512, V = 4232): the
    fused projection + loss forward and backward; the loss against the C loss oracle run on the float64 joint of the
    same rounded parameters (measured 5e-6 relative: the bf16 roundings of hidden and logits average out over the path; bar
    1e-4), every gradient against float64 autograd through that joint and the oracle's gradient, within 2 % of the
    tensor's largest entry (measured 0.03-1.0 %: hidden and logits are rounded to bf16 on the way)."""
    import rnnt_amd

    torch.manual_seed(11)
    rng = np.random.default_rng(11)
    # fp32, small: identical to RNNTLoss()(net(enc, dec), ...)
    B, T, U, He, Hd, H, V = 2, 9, 3, 16, 8, 32, 70
    net = rnnt_amd.JointNet(He + Hd, H, V).to(DEV)
    enc, dec = torch.randn(B, T, He, device=DEV), torch.randn(B, U + 1, Hd, device=DEV)
    labels = torch.from_numpy(rng.integers(1, V, size=(B, U)).astype(np.int32)).to(DEV)
    al = torch.tensor([T, 6], dtype=torch.int32, device=DEV)
    ll = torch.tensor([U, 1], dtype=torch.int32, device=DEV)
    e1, d1 = enc.clone().requires_grad_(True), dec.clone().requires_grad_(True)
    l1 = net.transducer_loss(e1, d1, labels, al, ll)
    l1.backward()
    g1 = [p.grad.clone() for p in net.parameters()]
    net.zero_grad()
    e2, d2 = enc.clone().requires_grad_(True), dec.clone().requires_grad_(True)
    l2 = rnnt_amd.RNNTLoss()(net(e2, d2), labels, al, ll)
    l2.backward()
    assert torch.equal(l1, l2) and torch.equal(e1.grad, e2.grad) and torch.equal(d1.grad, d2.grad)
    assert all(torch.equal(a, p.grad) for a, p in zip(g1, net.parameters()))

    # bf16, the reference's layer sizes: the fused path
    B, T, U, He, Hd, H, V = 3, 150, 12, 320, 320, 512, 4232
    net = rnnt_amd.JointNet(He + Hd, H, V).to(DEV).to(torch.bfloat16)
    enc = torch.randn(B, T, He, device=DEV).to(torch.bfloat16)
    dec = torch.randn(B, U + 1, Hd, device=DEV).to(torch.bfloat16)
    lab = rng.integers(1, V, size=(B, U)).astype(np.int32)
    aln, lln = np.array([T, 97, 140], np.int32), np.array([U, 5, 12], np.int32)
    e, d = enc.clone().requires_grad_(True), dec.clone().requires_grad_(True)
    loss = net.transducer_loss(e, d, torch.from_numpy(lab).to(DEV), torch.from_numpy(aln).to(DEV),
                               torch.from_numpy(lln).to(DEV))
    loss.backward()
    # float64 reference of the same rounded parameters and inputs
    params = [p.detach().double().cpu().requires_grad_(True) for p in net.parameters()]
    e64, d64 = enc.double().cpu().requires_grad_(True), dec.double().cpu().requires_grad_(True)
    logits = joint_logits(e64, d64, *params)
    c, g, _ = oracle.rnnt_loss_c(logits.detach().numpy(), lab, aln, lln, dtype=np.float64)
    logits.backward(torch.from_numpy(g / B))
    assert abs(loss.item() - c.sum() / B) <= 1e-4 * abs(c.sum() / B)
    assert rel_to_max(e.grad.double().cpu(), e64.grad) <= 0.02
    assert rel_to_max(d.grad.double().cpu(), d64.grad) <= 0.02
    for p, q in zip(net.parameters(), params):
        assert rel_to_max(p.grad.double().cpu(), q.grad) <= 0.02
    with pytest.raises(ValueError):
        net.transducer_loss(enc[0], dec[0], torch.from_numpy(lab).to(DEV), torch.from_numpy(aln).to(DEV),
                            torch.from_numpy(lln).to(DEV))


def test_jointnet_transducer_loss_never_waits_on_the_host():
    """The one-call joint + loss (fused path) and its backward enqueue work only: no host synchronisation."""
    import rnnt_amd

    torch.manual_seed(5)
    B, T, U, He, Hd, H, V = 2, 120, 9, 64, 32, 512, 1000
    net = rnnt_amd.JointNet(He + Hd, H, V).to(DEV).to(torch.bfloat16)
    enc = torch.randn(B, T, He, device=DEV).to(torch.bfloat16).requires_grad_(True)
    dec = torch.randn(B, U + 1, Hd, device=DEV).to(torch.bfloat16).requires_grad_(True)
    labels = torch.randint(1, V, (B, U), device=DEV, dtype=torch.int32)
    al = torch.tensor([T, 77], dtype=torch.int32, device=DEV)
    ll = torch.tensor([U, 4], dtype=torch.int32, device=DEV)
    net.transducer_loss(enc, dec, labels, al, ll).backward()      # (first call: library load, allocator warm-up)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        loss = net.transducer_loss(enc, dec, labels, al, ll)
        loss.backward()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert bool(torch.isfinite(loss).all()) and bool(torch.isfinite(enc.grad.float()).all())
    # lengths on the host, as the reference has them (model.py:104-111): counted there, sent over by one asynchronous copy
    # from pinned memory — still no wait; the compacted routes, fused (bf16) and library GEMM + compacted loss (fp32)
    al_host, ll_host = al.cpu(), ll.cpu()
    for dtype in (torch.bfloat16, torch.float32):
        net2 = rnnt_amd.JointNet(He + Hd, H, V).to(DEV).to(dtype)
        e2, d2 = enc.detach().to(dtype).requires_grad_(True), dec.detach().to(dtype).requires_grad_(True)
        net2.transducer_loss(e2, d2, labels, al_host, ll_host).backward()
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            loss = net2.transducer_loss(e2, d2, labels, al_host, ll_host)
            loss.backward()
        finally:
            torch.cuda.set_sync_debug_mode("default")
        assert bool(torch.isfinite(loss).all())


def test_joint_hidden_config2_size():
    """B32 T400 U40 H512 (the reference's joint at BASELINE config 2): 268 M outputs, checked against
    the same expression evaluated by torch on the device, and the reductions against torch sums."""
    import rnnt_amd

    g = torch.Generator(device=DEV).manual_seed(2)
    e = torch.randn(32, 400, 512, device=DEV, generator=g).requires_grad_(True)
    d = torch.randn(32, 41, 512, device=DEV, generator=g).requires_grad_(True)
    h = rnnt_amd.joint_hidden(e, d)
    ref = torch.tanh(e.detach()[:, :, None, :] + d.detach()[:, None, :, :])
    assert (h.detach() - ref).abs().max().item() <= 1e-6
    gh = torch.randn(h.shape, device=DEV, generator=g)
    h.backward(gh)
    gpre = gh * (1 - ref * ref)
    assert rel_to_max(e.grad, gpre.sum(2)) <= 1e-5
    assert rel_to_max(d.grad.double(), gpre.double().sum(1)) <= 1e-5


def test_joint_hidden_argument_checks():
    import rnnt_amd

    e = torch.zeros(2, 3, 8, device=DEV)
    with pytest.raises(ValueError):
        rnnt_amd.joint_hidden(e, torch.zeros(3, 4, 8, device=DEV))
    with pytest.raises(TypeError):
        rnnt_amd.joint_hidden(e, torch.zeros(2, 4, 8, device=DEV, dtype=torch.bfloat16))
    with pytest.raises(RuntimeError, match="no CPU"):
        rnnt_amd.joint_hidden(e.cpu(), torch.zeros(2, 4, 8))


def test_projection_with_split_weight_gradient():
    """rnnt_amd.joint._Projection (the output projection of the fp32 routes): the weight gradient as 32 partial products
    over row ranges added in range order, against torch's own Linear backward in float64."""
    from rnnt_amd.joint import _Projection

    g = torch.Generator(device=DEV).manual_seed(4)
    for n in (100, 8192, 20011):
        x = torch.randn(n, 512, device=DEV, generator=g).requires_grad_(True)
        w = (torch.randn(700, 512, device=DEV, generator=g) * 0.05).requires_grad_(True)
        b = torch.randn(700, device=DEV, generator=g).requires_grad_(True)
        go = torch.randn(n, 700, device=DEV, generator=g)
        y = _Projection.apply(x, w, b)
        y.backward(go)
        x64, w64, b64 = (t.detach().double().requires_grad_(True) for t in (x, w, b))
        y64 = torch.nn.functional.linear(x64, w64, b64)
        y64.backward(go.double())
        assert rel_to_max(y.detach().double(), y64.detach()) <= 5e-6
        for got, ref in ((x.grad, x64.grad), (w.grad, w64.grad), (b.grad, b64.grad)):
            assert rel_to_max(got.double(), ref) <= 5e-6
        x3, w3 = x.detach().requires_grad_(True), w.detach().requires_grad_(True)
        _Projection.apply(x3, w3, b.detach()).backward(go)        # a second run: the same bits (fixed summation order)
        assert torch.equal(w3.grad, w.grad)


@pytest.mark.parametrize("amp", [torch.bfloat16, torch.float16])
def test_transducer_loss_under_autocast_takes_the_fused_route(amp):
    """fp32 parameters and fp32 states under torch.autocast (mixed-precision training of the reference's fp32 model):
    `JointNet.transducer_loss` casts the states and the projection's weight and runs the fused 16-bit kernels; loss
    and fp32 gradients agree with the fp32 route to the 16-bit rounding of states, hidden layer and logits."""
    import rnnt_amd
    from cases import make_case

    B, T, U, V = 4, 96, 47, 600                                    # 18 432 rows: above the fused kernels' threshold
    torch.manual_seed(2)
    joint = rnnt_amd.JointNet(640, 512, V).to(DEV)
    _, labels, al, ll = make_case(B, T, U, V, seed=9, ragged=True)
    lab, al_host, ll_host = torch.from_numpy(labels).to(DEV), torch.from_numpy(al), torch.from_numpy(ll)
    enc0 = 0.5 * torch.randn(B, T, 320, device=DEV)
    dec0 = 0.5 * torch.randn(B, U + 1, 320, device=DEV)

    def run(autocast):
        joint.zero_grad(set_to_none=True)
        enc, dec = enc0.clone().requires_grad_(True), dec0.clone().requires_grad_(True)
        if autocast:
            with torch.autocast("cuda", dtype=amp):
                loss = joint.transducer_loss(enc, dec, lab, al_host, ll_host)
        else:
            loss = joint.transducer_loss(enc, dec, lab, al_host, ll_host)
        loss.backward()
        return [loss.detach(), enc.grad, dec.grad] + [p.grad.clone() for p in joint.parameters()]

    import rnnt_amd.fused as fused_mod
    calls = []
    orig = fused_mod._FusedJointLossFunction.apply
    fused_mod._FusedJointLossFunction.apply = lambda *a: (calls.append(a[0].dtype), orig(*a))[1]
    try:
        mixed = run(True)
    finally:
        fused_mod._FusedJointLossFunction.apply = orig
    assert calls == [amp]                                          # the fused kernels ran, on 16-bit operands
    full = run(False)
    assert all(t.dtype is torch.float32 for t in mixed[1:])        # gradients of fp32 leaves stay fp32
    tol = 3e-2 if amp is torch.bfloat16 else 6e-3
    torch.testing.assert_close(mixed[0], full[0], rtol=tol, atol=0)
    for a, b in zip(mixed[1:], full[1:]):
        assert float((a - b).abs().max()) <= tol * float(b.abs().max())
