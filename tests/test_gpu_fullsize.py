"""GPU tests at BASELINE.json's full sizes, through size-independent properties (the CPU oracle
would take minutes there): rows of the gradient sum to zero, per-node logit shifts change nothing,
padding is exactly zero and its content (NaN) is ignored, batch permutation commutes, repeated
runs are bitwise identical (no atomics anywhere), alpha/beta agree on the likelihood, and a
sampled utterance matches the oracle."""
import os

import numpy as np
import pytest
import torch

import oracle
from cases import assert_grad_parity

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def device_case(B, T, U, V, ragged, seed=1234, dtype=torch.float32, nan_padding=False):
    g = torch.Generator(device=DEV).manual_seed(seed)
    acts = torch.empty((B, T, U + 1, V), dtype=dtype, device=DEV)
    for b in range(B):
        if dtype is torch.float32:
            acts[b].normal_(0, 1, generator=g)
        else:
            for t0 in range(0, T, 100):
                t1 = min(T, t0 + 100)
                acts[b, t0:t1] = torch.randn((t1 - t0, U + 1, V), device=DEV, generator=g).to(dtype)
    rng = np.random.default_rng(seed)
    labels = rng.integers(1, V, size=(B, U)).astype(np.int32)
    if ragged:
        al = rng.integers(max(1, T // 8), T + 1, size=B).astype(np.int32)
        ll = rng.integers(max(0, U // 6), U + 1, size=B).astype(np.int32)
        al[0], ll[0] = T, U
        for b in range(B):
            labels[b, ll[b]:] = 0
            if nan_padding:
                acts[b, al[b]:] = float("nan")
                acts[b, :, ll[b] + 1:] = float("nan")
    else:
        al = np.full(B, T, np.int32)
        ll = np.full(B, U, np.int32)
    return acts, labels, al, ll


def run(acts, labels, al, ll, reduction="mean", go=0.5):
    import rnnt_amd

    a = acts.detach().requires_grad_(True)
    loss = rnnt_amd.RNNTLoss(reduction=reduction)(a, torch.from_numpy(labels).to(DEV), torch.from_numpy(al).to(DEV),
                                                  torch.from_numpy(ll).to(DEV))
    loss.backward(torch.full_like(loss, go))
    return loss.detach(), a.grad


def check_row_sums_and_padding(grads, al, ll, tol):
    B = grads.shape[0]
    rs = grads.float().sum(-1)  # [B,T,U+1]
    for b in range(B):
        T, U = int(al[b]), int(ll[b])
        assert float(rs[b, :T, :U + 1].abs().max()) <= tol
        assert not bool(grads[b, T:].any()) and not bool(grads[b, :, U + 1:].any())  # exact zeros
    assert bool(torch.isfinite(grads).all())


def test_config2_full_size_properties():
    """BASELINE.json configs[1]: B=32 T=400 U=40 V=4232 fp32, uniform lengths."""
    acts, labels, al, ll = device_case(32, 400, 40, 4232, ragged=False)
    loss, g = run(acts, labels, al, ll)
    assert loss.shape == (1,) and bool(torch.isfinite(loss).all())
    check_row_sums_and_padding(g, al, ll, tol=2e-6)
    # the gradient of the mean w.r.t. blank logits is <= 0 off the final node: mass leaves through blank
    # determinism: second run bitwise identical
    loss2, g2 = run(acts, labels, al, ll)
    assert torch.equal(loss, loss2) and torch.equal(g, g2)
    del g2
    # per-node shift invariance (softmax): add a different constant to every node's row
    shift = torch.randn(acts.shape[:3] + (1,), device=DEV) * 3
    loss3, g3 = run(acts + shift, labels, al, ll)
    assert abs(float(loss3) - float(loss)) <= 2e-5 * abs(float(loss))
    assert float((g3 - g).abs().max()) <= 1e-4
    del g3, shift
    # one utterance against the oracle (unpadded == batched)
    b = 17
    c, go_, _ = oracle.rnnt_loss_c(acts[b:b + 1].cpu().numpy(), labels[b:b + 1], al[b:b + 1], ll[b:b + 1])
    assert float(np.abs(g[b].cpu().numpy() - go_[0] * (0.5 / 32)).max()) <= 1e-4
    costs, _ = run(acts, labels, al, ll, reduction="none", go=1.0)
    assert abs(float(costs[b]) - c[0]) <= 1e-5 * abs(c[0])
    # reduction 'mean' is the index-order fp32 sum of 'none' divided by B
    s = np.float32(0)
    for v in costs.cpu().numpy():
        s = np.float32(s + v)
    assert float(loss) == float(np.float32(s / np.float32(32)))


def test_config2_batch_permutation():
    acts, labels, al, ll = device_case(8, 400, 40, 4232, ragged=True, seed=99)
    costs, g = run(acts, labels, al, ll, reduction="none", go=1.0)
    perm = np.array([3, 7, 0, 5, 1, 6, 2, 4])
    pt = torch.from_numpy(perm).to(DEV)
    costs_p, g_p = run(acts[pt].contiguous(), labels[perm], al[perm], ll[perm], reduction="none", go=1.0)
    assert torch.equal(costs_p, costs[pt]) and torch.equal(g_p, g[pt])  # utterances are independent: bitwise


def test_config3_ragged_nan_padding():
    """BASELINE.json configs[2]: ragged B=64 (T 100-800, U 10-60); padded logits are NaN to prove masking."""
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    B = 64
    if free < 112e9:  # 52.9 GB logits + 52.9 GB gradients + temporaries
        pytest.skip(f"CONFIG 3 NOT RUN AT FULL SIZE: needs 112 GB of device memory, {free / 1e9:.1f} GB free")
    acts, labels, al, ll = device_case(B, 800, 60, 4232, ragged=True, seed=7, nan_padding=True)
    loss, g = run(acts, labels, al, ll)
    assert bool(torch.isfinite(loss).all())
    check_row_sums_and_padding(g, al, ll, tol=2e-6)
    # ragged == per-utterance unpadded, on the shortest utterance, against the oracle too
    b = int(np.argmin(al.astype(np.int64) * (ll + 1)))
    T, U = int(al[b]), int(ll[b])
    sub = acts[b:b + 1, :T, :U + 1].contiguous()
    costs, gs = run(sub, labels[b:b + 1, :U].copy(), al[b:b + 1], ll[b:b + 1], reduction="none", go=1.0)
    costs_all, _ = run(acts, labels, al, ll, reduction="none", go=1.0)
    assert float(costs[0]) == float(costs_all[b])
    c, go_, _ = oracle.rnnt_loss_c(sub.cpu().numpy(), labels[b:b + 1, :U], [T], [U])
    assert abs(float(costs[0]) - c[0]) <= 1e-5 * abs(c[0])
    # UNSCALED per-utterance gradients (reduction 'none', grad_output 1) reach O(1) while |ll| ~ 1e3,
    # where one fp32 ulp of alpha+beta-ll is ~1e-4: two fp32 implementations (different summation
    # order in the softmax denominators) differ by that much relative.  So: 1e-4 absolute on the
    # gradient the reference's call actually produces ('mean' over B, checked below), and
    # 1e-4 + 1e-3 |g| on the unscaled one; and the GPU must be no further from the float64 oracle
    # than the fp32 CPU oracle is (x2 slack).
    gnp = gs.cpu().numpy()
    _, g64, _ = oracle.rnnt_loss_c(sub.cpu().numpy(), labels[b:b + 1, :U], [T], [U], dtype=np.float64)
    assert_grad_parity(gnp, go_, g64, reduction="none", label=f"config3 shortest utterance T={T} U={U} unscaled")
    # the gradient the reference's call produces ('mean' over B, grad_output 0.5): the strict bar
    assert_grad_parity(gnp * np.float32(0.5 / B), go_ * np.float32(0.5 / B), reduction="mean", label="config3 mean")
    assert float((g[b, :T, :U + 1] - gs[0] * (0.5 / B)).abs().max()) <= 1e-7


def test_config5_shape_bf16_slice():
    """BASELINE.json configs[4] numerics at reduced B/T (same U=150, V=16384, bf16 IO, fp32 lattice):
    three waves per direction in K2, two waves per node in K1, two segments per node in K3."""
    acts, labels, al, ll = device_case(2, 300, 150, 16384, ragged=True, seed=5, dtype=torch.bfloat16)
    loss, g = run(acts, labels, al, ll)
    assert bool(torch.isfinite(loss).all())
    check_row_sums_and_padding(g, al, ll, tol=3e-3)  # bf16-rounded gradients
    c, go_, _ = oracle.rnnt_loss_c(acts.float().cpu().numpy(), labels, al, ll)
    ref = np.float32(c.sum(dtype=np.float32) / 2)
    assert abs(float(loss) - ref) <= 1e-5 * abs(ref)
    ref_g = go_ * (0.5 / 2)
    assert np.all(np.abs(g.float().cpu().numpy() - ref_g) <= 1e-4 + 2.0 ** -8 * np.abs(ref_g))


def test_invalid_lengths_give_nan_not_fault():
    """check_lengths=False: the kernels guard themselves (NaN cost, zero gradient, no OOB access)."""
    import rnnt_amd

    acts, labels, al, ll = device_case(3, 6, 3, 64, ragged=False, seed=3)
    al = al.copy()
    ll = ll.copy()
    al[1] = 7   # > T
    ll[2] = -1  # < 0
    a = acts.requires_grad_(True)
    costs = rnnt_amd.RNNTLoss(reduction="none", check_lengths=False)(
        a, torch.from_numpy(labels).to(DEV), torch.from_numpy(al).to(DEV), torch.from_numpy(ll).to(DEV))
    costs.backward(torch.ones_like(costs))
    c = costs.detach().cpu().numpy()
    assert np.isfinite(c[0]) and np.isnan(c[1]) and np.isnan(c[2])
    assert not bool(a.grad[1].any()) and not bool(a.grad[2].any()) and bool(torch.isfinite(a.grad[0]).all())
    with pytest.raises(ValueError, match="Input length mismatch"):
        rnnt_amd.RNNTLoss()(a, torch.from_numpy(labels).to(DEV), torch.from_numpy(al).to(DEV),
                            torch.from_numpy(np.maximum(ll, 0)).to(DEV))


def test_compute_rnnt_loss_upstream_shaped_call():
    """The upstream-shaped single C entry point: costs + unscaled gradients."""
    import ctypes

    from rnnt_amd import _lib
    from cases import make_case

    acts, labels, al, ll = make_case(3, 10, 4, 40, seed=42, ragged=True)
    lib = _lib.load()
    d = lambda x: torch.from_numpy(x).to(DEV)
    a, lab, ald, lld = d(acts), d(labels), d(al), d(ll)
    grads = torch.empty_like(a)
    costs = torch.empty(3, dtype=torch.float32, device=DEV)
    ws = torch.empty(_lib.workspace_bytes(10, 5, 3), dtype=torch.uint8, device=DEV)
    opts = _lib.RnntOptions(torch.cuda.current_stream().cuda_stream, 0, 10, 5, _lib.DTYPE_F32, None)
    _lib.check(lib.compute_rnnt_loss(a.data_ptr(), grads.data_ptr(), lab.data_ptr(), lld.data_ptr(), ald.data_ptr(),
                                     40, 3, costs.data_ptr(), ws.data_ptr(), opts), "compute_rnnt_loss")
    torch.cuda.synchronize()
    c, g, _ = oracle.rnnt_loss_c(acts, labels, al, ll)
    np.testing.assert_allclose(costs.cpu().numpy(), c, rtol=1e-5)
    assert np.abs(grads.cpu().numpy() - g).max() <= 1e-4
    # misaligned workspace is rejected with a status, not a fault
    bad = lib.compute_rnnt_loss(a.data_ptr(), grads.data_ptr(), lab.data_ptr(), lld.data_ptr(), ald.data_ptr(),
                                40, 3, costs.data_ptr(), ws.data_ptr() + 4, opts)
    assert bad == 2


def test_config4_sharded_equals_unsharded_bitwise_at_full_size():
    """BASELINE config 4 (B=256, T=400, U=40, V=4232 fp32; 71 GB of logits) on ONE GPU: the whole batch in a
    single call against its eight contiguous shards of 32 utterances run one after the other exactly as
    the eight ranks of `bench.py --gpus 8` run them (per-utterance costs with reduction='none', local
    gradient scaled by grad_out / B_global).  Costs and gradients must agree bit for bit; the all-reduce
    of the zero-padded cost vector is exact, so this is the whole of the sharded path but the wire."""
    import rnnt_amd

    B, T, U, V, W = 256, 400, 40, 4232, 8
    acts, labels, al, ll = device_case(B, T, U, V, ragged=False, seed=99)
    lab_d, al_d, ll_d = (torch.from_numpy(x).to(DEV) for x in (labels, al, ll))
    go = 0.5
    a = acts.requires_grad_(True)
    costs = rnnt_amd.rnnt_loss(a, lab_d, al_d, ll_d, reduction="none")
    mean = rnnt_amd.loss.reduce_costs(costs.detach(), "mean")
    costs.backward(torch.full((B,), go / B, device=DEV))     # what 'mean' with grad_output = go does
    full_grad = a.grad
    a.grad = None
    loss_mean, grad_mean = run(acts.detach(), labels, al, ll, "mean", go)
    assert torch.equal(loss_mean, mean) and torch.equal(grad_mean, full_grad)
    del grad_mean
    gathered = torch.zeros(B, device=DEV)
    for r in range(W):
        lo, hi = rnnt_amd.shard_bounds(B, W, r)
        s = acts.detach()[lo:hi].requires_grad_(True)
        c = rnnt_amd.rnnt_loss(s, lab_d[lo:hi], al_d[lo:hi], ll_d[lo:hi], reduction="none")
        c.backward(torch.full((hi - lo,), go * (1.0 / B), device=DEV))
        assert torch.equal(s.grad, full_grad[lo:hi]), f"shard {r}: gradient differs from the unsharded one"
        gathered[lo:hi] = c.detach()
        del s, c
    assert torch.equal(gathered, costs.detach())
    assert torch.equal(rnnt_amd.loss.reduce_costs(gathered, "mean"), mean)
    assert bool(torch.isfinite(mean).all())


def lattice_planes(acts, labels, al, ll):
    """rnnt_forward through the C ABI with a workspace of our own: returns (costs, ll from alpha, beta(0,0))
    per utterance.  Layout as DESIGN.md §2: six fp32 planes of B*(ND+1)*US (+2*US) floats, node (b,t,u) at
    (b*(ND+1) + t+u)*US + u; beta is plane 5, ll[B] follows plane 6."""
    from rnnt_amd import _lib
    from rnnt_amd.loss import _DTYPES

    B, T, U1, V = acts.shape
    ws = torch.empty(_lib.workspace_bytes(T, U1, B), dtype=torch.uint8, device=DEV)
    costs = torch.empty(B, dtype=torch.float32, device=DEV)
    opts = _lib.RnntOptions(torch.cuda.current_stream().cuda_stream, 0, T, U1, _DTYPES[acts.dtype], None)
    d = lambda x: torch.from_numpy(x).to(DEV)
    lab, ald, lld = d(labels), d(al), d(ll)
    _lib.check(_lib.load().rnnt_forward(acts.data_ptr(), lab.data_ptr(), ald.data_ptr(), lld.data_ptr(), V, B,
                                        costs.data_ptr(), None, 0, ws.data_ptr(), opts), "rnnt_forward")
    torch.cuda.synchronize()
    f = ws.view(torch.float32)
    ND, US = T + U1 - 1, (U1 + 63) // 64 * 64
    plane = B * (ND + 1) * US + 2 * US
    beta00 = f[5 * plane: 6 * plane][torch.arange(B, device=DEV) * (ND + 1) * US]
    return costs.cpu().numpy(), f[6 * plane: 6 * plane + B].cpu().numpy(), beta00.cpu().numpy()


@pytest.mark.timeout(600)
def test_config5_full_size():
    """BASELINE.json configs[4] AT FULL SIZE: B=16 T=1500 U=150 V=16384, bf16 logits and gradients (118.75 GB
    each), fp32 lattice: 5.9e10-element indexing in one call, 1650 K2 steps on three waves per direction.
    Uniform lengths as the configuration says, except the last utterance (T=1100, U=100) so that padding
    exists.  The gradient cannot be held twice, so repeatability is checked on per-utterance checksums."""
    B, T, U, V = 16, 1500, 150, 16384
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    need = 2 * B * T * (U + 1) * V * 2 + 20e9
    if free < need:
        pytest.skip(f"CONFIG 5 NOT RUN AT FULL SIZE: needs {need / 1e9:.0f} GB of device memory, "
                    f"{free / 1e9:.1f} of {total / 1e9:.1f} GB free")
    acts, labels, al, ll = device_case(B, T, U, V, ragged=False, seed=55, dtype=torch.bfloat16)
    al[B - 1], ll[B - 1] = 1100, 100
    labels[B - 1, 100:] = 0
    acts[B - 1, 1100:] = float("nan")          # padding content must not matter
    acts[B - 1, :, 101:] = float("nan")

    def checksums(g):
        # per utterance: sum of the raw 16-bit patterns (exact integer arithmetic), and the largest |row sum|
        sums, worst = [], 0.0
        for b in range(B):
            sums.append(int(g[b].view(torch.int16).to(torch.int32).sum(dtype=torch.int64)))
            for t0 in range(0, int(al[b]), 250):
                rs = g[b, t0:min(t0 + 250, int(al[b])), :int(ll[b]) + 1].float().sum(-1)
                worst = max(worst, float(rs.abs().max()))
        return sums, worst

    loss, g = run(acts, labels, al, ll)
    assert loss.shape == (1,) and bool(torch.isfinite(loss).all())
    assert not bool(g[B - 1, 1100:].any()) and not bool(g[B - 1, :, 101:].any())   # exact zeros in the padding
    for b in range(0, B, 5):
        assert bool(torch.isfinite(g[b]).all())
    sums1, worst = checksums(g)
    # a valid node's gradients sum to zero over V up to the bf16 rounding of 16384 stored values
    assert worst <= 3e-3, worst
    g_last = g[B - 1, :1100, :101].clone()
    del g
    loss2, g = run(acts, labels, al, ll)
    sums2, _ = checksums(g)
    assert torch.equal(loss, loss2) and sums1 == sums2                    # bitwise repeatable (no atomics)
    assert torch.equal(g_last, g[B - 1, :1100, :101])
    del g, g_last
    torch.cuda.empty_cache()

    costs, ll_alpha, beta00 = lattice_planes(acts, labels, al, ll)
    assert np.all(np.isfinite(costs)) and np.array_equal(costs, -ll_alpha)
    np.testing.assert_allclose(beta00, ll_alpha, rtol=1e-5)              # alpha and beta agree on the likelihood
    s = np.float32(0)
    for v in costs:
        s = np.float32(s + v)
    assert float(loss) == float(np.float32(s / np.float32(B)))           # 'mean' = index-order fp32 sum / B

    # one full-length utterance and the ragged one against the CPU oracle (forward only: the oracle's
    # gradient buffer would be another 15 GB per utterance)
    for b in (3, B - 1):
        Tb, Ub = int(al[b]), int(ll[b])
        sub = acts[b, :Tb, :Ub + 1].float().cpu().numpy()[None]
        c = oracle.rnnt_loss_c(sub, labels[b:b + 1, :Ub], [Tb], [Ub], want_grads=False)[0]
        assert abs(float(costs[b]) - c[0]) <= 1e-5 * abs(c[0]), (b, costs[b], c[0])
        del sub


@pytest.mark.parametrize("B,V", [(16, 64), (64, 16)])
def test_config5_lattice_mean_reduction_gradient_bar(B, V):
    """reduction='mean' — the reference's only mode (model.py:87) — at config 5's lattice (T=1500, U=150; a small
    vocabulary so that the CPU oracle can hold the gradients), against the fp32 CPU oracle AND float64.

    The bar as DESIGN.md §4 states it:  |g - g_cpu32| <= max(1e-4, 8 ulp32(max|ll|) |grad_output| / B).
    |ll| ~ 6e3 here, so one fp32 ulp of alpha + beta - ll is 4.9e-4; the recurrences of both fp32 implementations
    accumulate a few of those over the 1650 anti-diagonals (the fp32 oracle itself sits that far from float64), and
    the 1/B of 'mean' scales the result.  B = 64: the bound is 6e-5 and the north-star's 1e-4 is asserted strictly.
    B = 16 (config 5's batch): the bound is 2.4e-4; measured 1.2e-4 against the fp32 oracle, i.e. the strict 1e-4
    is NOT met there — and cannot be by any fp32 implementation that is not the oracle's own rounding sequence, since
    the oracle is further than that from float64.  What is asserted instead: the stated bound, and that the GPU is no
    further from float64 than twice the fp32 oracle's own distance.  Every measured figure goes to
    gpurun_out/relaxed_grad_parity.jsonl (committed per round under profiles/)."""
    import json

    from cases import RELAXED_LOG, make_case

    T, U = 1500, 150
    acts, labels, al, ll = make_case(B, T, U, V, seed=1516, ragged=True)
    al[0], ll[0] = T, U
    c, g, _ = oracle.rnnt_loss_c(acts, labels, al, ll)
    ref_loss, ref_g = oracle.reduce_like_upstream(c, g, "mean", 1.0)
    del g
    _, g64, _ = oracle.rnnt_loss_c(acts, labels, al, ll, dtype=np.float64)
    g64 /= B
    loss, gm = run(torch.from_numpy(acts).to(DEV), labels, al, ll, reduction="mean", go=1.0)
    gm = gm.cpu().numpy()
    ref_loss = float(np.asarray(ref_loss).ravel()[0])
    assert abs(float(loss.reshape(-1)[0]) - ref_loss) <= 1e-5 * abs(ref_loss)
    max_ll = float(np.abs(c).max())
    assert max_ll > 4096                                                   # the regime the statement is about
    ulp = float(np.spacing(np.float32(max_ll)))
    bound = max(1e-4, 8.0 * ulp * 1.0 / B)
    err = float(np.abs(gm - ref_g).max())
    e_gpu, e_cpu = float(np.abs(gm - g64).max()), float(np.abs(ref_g - g64).max())
    try:
        os.makedirs(os.path.dirname(RELAXED_LOG), exist_ok=True)
        with open(RELAXED_LOG, "a") as f:
            f.write(json.dumps({"case": f"T1500 U150 V{V} B{B} mean", "reduction": "mean", "shape": list(gm.shape),
                                "max_abs_ll": max_ll, "ulp32_of_ll": ulp, "stated_bound": bound,
                                "max_abs_dg_vs_f32_oracle": err, "gpu_vs_f64": e_gpu, "f32_oracle_vs_f64": e_cpu,
                                "strict_1e-4_met": err <= 1e-4}) + "\n")
    except OSError:
        pass
    assert err <= bound, (err, bound)
    assert e_gpu <= max(2 * e_cpu, 1e-4), (e_gpu, e_cpu)
    if B == 64:
        assert bound == 1e-4 and err <= 1e-4


def test_config5_lattice_shape_small_vocab_with_gradients():
    """T=1500, U=150 (config 5's lattice: 1650 anti-diagonals, three K2 waves per direction with the LDS
    edge hand-off) at a vocabulary small enough for the oracle to produce gradients: costs and gradients
    against the fp32 CPU oracle; one utterance full length, one ragged."""
    from cases import make_case

    B, T, U, V = 2, 1500, 150, 256
    acts, labels, al, ll = make_case(B, T, U, V, seed=1500, ragged=False)
    al[1], ll[1] = 977, 131
    labels[1, 131:] = 0
    a = torch.from_numpy(acts).to(DEV)
    c, g, _ = oracle.rnnt_loss_c(acts, labels, al, ll)
    _, g64, _ = oracle.rnnt_loss_c(acts, labels, al, ll, dtype=np.float64)
    costs, gn = run(a, labels, al, ll, reduction="none", go=1.0)
    np.testing.assert_allclose(costs.cpu().numpy(), c, rtol=1e-5)
    assert_grad_parity(gn.cpu().numpy(), g, g64, reduction="none", label="T1500 U150 V256 unscaled")
    # |ll| is ~9e3 here, one fp32 ulp of alpha+beta-ll is 1e-3: the fp32 CPU oracle itself sits 8e-3 from
    # float64 on the few nodes every path crosses (profiles/r03_relaxed_grad_parity.jsonl), so 1e-4 absolute
    # is below what fp32 resolves on this lattice FOR UNSCALED GRADIENTS, for the reference's CPU path as for this
    # one; the check above is the float64-anchored one.  The bar itself: 1e-4 absolute holds while
    # 8 ulp(|ll|) * |grad_output| / B <= 1e-4 — see the 'mean' test above, which measures both sides of that line.
    ca, ll_alpha, beta00 = lattice_planes(a, labels, al, ll)
    np.testing.assert_allclose(beta00, ll_alpha, rtol=1e-5)
