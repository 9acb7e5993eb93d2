"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Tolerances (BASELINE.json north_star): fp32 loss <= 1e-5 relative, gradients <= 1e-4 absolute,
both against the fp32 CPU restatement (oracle/rnnt_oracle.c).  PARITY UNPINNED w.r.t. the real
warp-transducer (see oracle/__init__.py).
"""
import numpy as np
import pytest
import torch

import oracle
import glob
import os

from cases import KAT1, KAT2, make_case

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-5
GRAD_ATOL = 1e-4


def run_gpu(acts, labels, act_lens, label_lens, reduction="mean", grad_output=None, blank=0,
            dtype=torch.float32):
    import rnnt_amd

    dev = torch.device("cuda:0")
    a = torch.from_numpy(acts).to(dev).to(dtype).requires_grad_(True)
    loss = rnnt_amd.RNNTLoss(blank=blank, reduction=reduction)(
        a, torch.from_numpy(labels).to(dev), torch.from_numpy(act_lens).to(dev),
        torch.from_numpy(label_lens).to(dev))
    if grad_output is None:
        go = torch.ones_like(loss)
    else:
        go = torch.as_tensor(grad_output, dtype=torch.float32, device=dev).reshape(loss.shape)
    loss.backward(go)
    return loss.detach().cpu().numpy(), a.grad.detach().float().cpu().numpy()


def check_against_oracle(acts, labels, act_lens, label_lens, reduction="mean", grad_output=1.0, blank=0):
    loss, grads = run_gpu(acts, labels, act_lens, label_lens, reduction, grad_output, blank)
    c, g, _ = oracle.rnnt_loss_c(acts, labels, act_lens, label_lens, blank=blank)
    ref_loss, ref_g = oracle.reduce_like_upstream(c, g, reduction, grad_output)
    np.testing.assert_allclose(loss, ref_loss, rtol=LOSS_RTOL, atol=0)
    assert np.abs(grads - ref_g).max() <= GRAD_ATOL
    return loss, grads


def test_kat1_upstream_vector():
    loss, grads = run_gpu(KAT1["acts"], KAT1["labels"], KAT1["act_lens"], KAT1["label_lens"])
    assert loss.shape == (1,)
    np.testing.assert_allclose(loss[0], KAT1["cost"], rtol=LOSS_RTOL)
    assert np.abs(grads - KAT1["grads"]).max() <= 1e-6


def test_kat2_upstream_two_utterance_vector():
    """reduction='none' shows both of upstream's expected costs; 'mean' is RNNTLoss()'s default."""
    args = (KAT2["acts"], KAT2["labels"], KAT2["act_lens"], KAT2["label_lens"])
    costs, grads = run_gpu(*args, reduction="none")
    np.testing.assert_allclose(costs, KAT2["costs"], rtol=LOSS_RTOL)
    assert np.abs(grads[0, 0] - KAT2["grads_b0_t0"]).max() <= 1e-6
    loss, grads_mean = run_gpu(*args)
    np.testing.assert_allclose(loss[0], KAT2["costs"].mean(), rtol=LOSS_RTOL)
    assert np.abs(grads_mean - grads / 2).max() <= 1e-7


GOLDEN = sorted(f for f in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
                if not os.path.basename(f).startswith("ref_"))   # ref_*: the joint / decoding fixtures (ref_fixtures.py)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_committed_golden_fixtures(path):
    """The fixtures hold float64 answers (tests/golden/make_golden.py); the GPU gets the fp32 logits."""
    z = np.load(path)  # allow_pickle=False (default)
    costs, grads = run_gpu(z["acts"].astype(np.float32), z["labels"], z["act_lens"], z["label_lens"],
                           reduction="none", blank=int(z["blank"]))
    np.testing.assert_allclose(costs, z["costs"], rtol=LOSS_RTOL)
    assert np.abs(grads - z["grads"]).max() <= GRAD_ATOL


@pytest.mark.parametrize("B,T,U,V,seed", [
    (1, 1, 0, 4, 1), (2, 1, 3, 8, 2), (3, 5, 0, 12, 3), (4, 7, 5, 16, 4),
    (5, 33, 17, 64, 5), (2, 70, 63, 260, 6), (2, 50, 64, 1024, 7), (3, 40, 100, 128, 8),
])
@pytest.mark.parametrize("reduction", ["mean", "sum", "none"])
def test_small_ragged_vs_oracle(B, T, U, V, seed, reduction):
    acts, labels, al, ll = make_case(B, T, U, V, seed=seed, ragged=True, sigma=2.0)
    go = 0.5 if reduction != "none" else np.linspace(0.25, 1.5, B).astype(np.float32)
    check_against_oracle(acts, labels, al, ll, reduction, go)


@pytest.mark.parametrize("V", [29, 520, 4232, 8200, 17412, 70001])   # small-row / one wave / two waves / streaming / generic K1
@pytest.mark.parametrize("B", [63, 64, 65, 127, 130, 200])
def test_many_ragged_utterances_compacted_node_order(B, V):
    """K1 walks the nodes inside the lattices through per-utterance prefix offsets, 64 utterances per search step
    (rnnt_node_spans / find_node): batches around and beyond one step, with utterances that have no nodes at all
    (lengths that do not fit the tensor: NaN cost, zero gradient) at the start, in the middle and at the end."""
    T, U = (6, 3) if V > 5000 else (11, 4)
    acts, labels, al, ll = make_case(B, T, U, V, seed=B * 7 + V, ragged=True)
    bad = [0, B // 2, 62 if B > 63 else 5, B - 1]
    al_bad, ll_bad = al.copy(), ll.copy()
    al_bad[bad[0]] = T + 1
    ll_bad[bad[1]] = U + 1
    al_bad[bad[2]] = 0
    al_bad[bad[3]] = -3
    import rnnt_amd

    dev = torch.device("cuda:0")
    a = torch.from_numpy(acts).to(dev).requires_grad_(True)
    costs = rnnt_amd.RNNTLoss(reduction="none", check_lengths=False)(
        a, torch.from_numpy(labels).to(dev), torch.from_numpy(al_bad).to(dev), torch.from_numpy(ll_bad).to(dev))
    good = np.ones(B, bool)
    good[bad] = False
    go = torch.from_numpy(np.where(good, 1.0, 0.0).astype(np.float32)).to(dev)
    costs.backward(go)
    c, g, _ = oracle.rnnt_loss_c(acts, labels, al, ll)
    got = costs.detach().cpu().numpy()
    assert np.isnan(got[~good]).all()
    np.testing.assert_allclose(got[good], c[good], rtol=LOSS_RTOL)
    grads = a.grad.cpu().numpy()
    assert np.abs(grads[good] - g[good]).max() <= GRAD_ATOL
    assert not grads[~good].any()


@pytest.mark.parametrize("V", [5, 6, 7, 29, 32, 33, 64, 100, 128, 250, 256, 257, 500, 512, 513, 1000, 1024, 4234,
                               4232, 4100, 4352, 8200, 8196, 16384, 17408, 17412, 20000, 32768, 65536, 70001])
def test_vocab_sizes_and_alignment(V):
    # up to 128 chunks per row (V <= 512 fp32): the small-row kernels, eight lanes per node with 1, 2, 4, 8 or
    # 16 chunks per lane (K3 up to 8), several nodes of several utterances per wave instruction;
    # 4352 = exactly 17 chunks/lane; 8200 -> two waves per
    # node in K1 and two segments in K3; 16384 -> four; from 17408 on K1 streams the row with one
    # workgroup per node (running max/sum); 20000 -> five K3 segments
    acts, labels, al, ll = make_case(2, 9, 4, V, seed=V, ragged=True)
    check_against_oracle(acts, labels, al, ll, "mean", 0.5)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("V", [64, 128, 250, 1000, 1024, 4232, 4234, 16384, 40000])
def test_half_precision_logits(dtype, V):
    """bf16 / fp16 logits and grads with the fp32 lattice (BASELINE.json configs[4] numerics).

    Tolerance, stated separately from fp32 as north_star asks: the oracle runs in fp32 on the
    SAME rounded logits; loss within 1e-5 relative (the lattice is fp32 either way); each gradient
    within one rounding step of its dtype: |g - g_ref| <= 1e-4 + 2^-8 |g_ref| (bf16), 2^-11 (fp16).
    """
    acts, labels, al, ll = make_case(3, 12, 5, V, seed=V + 1, ragged=True, sigma=2.0)
    acts = torch.from_numpy(acts).to(dtype).float().numpy()  # what the kernel will actually see
    loss, grads = run_gpu(acts, labels, al, ll, "mean", 0.5, dtype=dtype)
    c, g, _ = oracle.rnnt_loss_c(acts, labels, al, ll)
    ref_loss, ref_g = oracle.reduce_like_upstream(c, g, "mean", 0.5)
    np.testing.assert_allclose(loss, ref_loss, rtol=LOSS_RTOL, atol=0)
    rel = 2.0 ** -8 if dtype is torch.bfloat16 else 2.0 ** -11
    assert np.all(np.abs(grads - ref_g) <= 1e-4 + rel * np.abs(ref_g))


def test_peaked_trained_like():
    acts, labels, al, ll = make_case(4, 60, 12, 512, seed=11, ragged=True, peaked=True)
    check_against_oracle(acts, labels, al, ll, "mean", 1.0)


def test_config1_shape_vs_oracle():
    # BASELINE.json configs[0]: B=8 T=200 U=20 V=4232 fp32 (the reference's CPU-runnable case)
    acts, labels, al, ll = make_case(8, 200, 20, 4232, seed=21, ragged=False)
    check_against_oracle(acts, labels, al, ll, "mean", 0.5)


def test_nonzero_blank():
    acts, labels, al, ll = make_case(3, 11, 6, 40, seed=31, ragged=True, blank=7)
    check_against_oracle(acts, labels, al, ll, "mean", 1.0, blank=7)


def test_labels_equal_to_blank_are_allowed():
    # upstream does not forbid y == blank; both sparse gradient terms then land on the same logit
    acts, labels, al, ll = make_case(3, 9, 5, 12, seed=41, ragged=True)
    labels[0, 1] = 0
    labels[1, 0] = 0
    labels[2, :] = 0
    check_against_oracle(acts, labels, al, ll, "mean", 1.0)


def test_masked_vocabulary_entries_minus_inf():
    acts, labels, al, ll = make_case(2, 8, 4, 20, seed=43, ragged=True)
    acts[:, :, :, 17:] = -np.inf          # never a label (labels < 17 below), never blank
    labels = np.minimum(labels, 16).astype(np.int32)
    loss, grads = check_against_oracle(acts, labels, al, ll, "mean", 1.0)
    assert np.isfinite(loss).all() and np.all(grads[..., 17:] == 0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("V", [1, 2, 3, 9, 4233, 4234, 5455, 8201])
def test_rows_not_16_byte_aligned(dtype, V):
    """The reference's own vocabularies include 4234 and 5455 (egs/aishell/exp train.log, config/magic_*):
    V*elem is then no multiple of 16 and rows start at every element offset of a 16-byte chunk.  The
    kernels move whole aligned chunks and treat the two boundary chunks by element; neighbouring rows
    (written by other waves) must come out intact, padding rows exactly zero."""
    B, T, U = 3, 5, 3
    acts, labels, al, ll = make_case(B, T, U, max(V, 2), seed=900 + V, ragged=True)
    if V == 1:
        acts, labels = acts[..., :1].copy(), np.zeros_like(labels)     # only blank exists: labels unused below
        ll[:] = 0
    q = torch.from_numpy(acts).to(dtype).float().numpy()
    c, g, _ = oracle.rnnt_loss_c(q, labels, al, ll)
    ref_loss, ref_g = oracle.reduce_like_upstream(c, g, "sum", 1.0)
    loss, grads = run_gpu(q, labels, al, ll, "sum", 1.0, dtype=dtype)
    np.testing.assert_allclose(loss, ref_loss, rtol=LOSS_RTOL)
    eps = {torch.float32: 0.0, torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
    assert np.all(np.abs(grads - ref_g) <= GRAD_ATOL + eps * np.abs(ref_g))
    assert np.all((grads == 0) | (ref_g != 0) | (np.abs(grads) <= GRAD_ATOL))
    for b in range(B):                                                 # padding stays exactly zero
        assert np.all(grads[b, al[b]:] == 0) and np.all(grads[b, :, ll[b] + 1:] == 0)


def test_sliced_batch_with_misaligned_base_pointer():
    """acts[1:] of a V=4233 tensor starts 12 bytes off a 16-byte boundary while autograd's fresh gradient
    buffer is aligned: logits and gradient rows then differ in misalignment, which the chunked kernels do
    not take (K1 does, K3 falls back to the element-granular kernel)."""
    import rnnt_amd

    acts, labels, al, ll = make_case(3, 5, 2, 4234 + 1, seed=77, ragged=True)
    acts = acts[..., :4233].copy()                                     # 4233*4 bytes per row: odd dword count
    labels = np.minimum(labels, 4232).astype(np.int32)
    dev = torch.device("cuda:0")
    full = torch.from_numpy(acts).to(dev)
    a = full[1:].detach().requires_grad_(True)                          # contiguous view, base % 16 != 0
    assert a.data_ptr() % 16 != 0 and a.is_contiguous()
    loss = rnnt_amd.RNNTLoss(reduction="sum")(a, torch.from_numpy(labels[1:]).to(dev),
                                              torch.from_numpy(al[1:]).to(dev), torch.from_numpy(ll[1:]).to(dev))
    loss.backward()
    c, g, _ = oracle.rnnt_loss_c(acts[1:], labels[1:], al[1:], ll[1:])
    np.testing.assert_allclose(loss.item(), c.sum(), rtol=LOSS_RTOL)
    assert np.abs(a.grad.cpu().numpy() - g).max() <= GRAD_ATOL


def test_wide_row_running_max_edge_cases():
    """V = 32768 takes K1's single-pass kernel: the running {max, sum} must survive a max that rises by
    hundreds late in the row, -inf stretches, a large common offset, and a NaN."""
    acts, labels, al, ll = make_case(2, 3, 2, 32768, seed=46, ragged=False)
    acts[0, :, :, :30000] -= 400.0        # first batches far below the final max
    acts[0, 1, 1, :32000] = -np.inf       # whole batches of -inf before finite values
    acts[1] += 1e4
    labels[0, :] = [32001, 32002]
    check_against_oracle(acts, labels, al, ll, "none", np.ones(2, np.float32))
    acts[1, 2, 0, 31999] = np.nan
    loss, _ = run_gpu(acts, labels, al, ll, "none", np.ones(2, np.float32))
    assert np.isfinite(loss[0]) and np.isnan(loss[1])


def test_extreme_logit_scale():
    acts, labels, al, ll = make_case(2, 10, 4, 33, seed=44, ragged=True, sigma=30.0)
    acts += 1e4                            # large common offset: needs the max subtraction
    check_against_oracle(acts, labels, al, ll, "mean", 1.0)


def test_nan_logit_gives_nan_loss_not_an_exception():
    # bin/train.py:36-41 expects to SEE a NaN loss and skip the update
    acts, labels, al, ll = make_case(2, 6, 3, 10, seed=45, ragged=False)
    acts[1, 2, 1, 4] = np.nan
    loss, grads = run_gpu(acts, labels, al, ll, "none", np.ones(2, np.float32))
    assert np.isfinite(loss[0]) and np.isnan(loss[1])
    assert np.isfinite(grads[0]).all()
    loss_mean, _ = run_gpu(acts, labels, al, ll, "mean")
    assert np.isnan(loss_mean[0])


@pytest.mark.parametrize("V,dtype", [(10, torch.float32), (300, torch.float32), (4232, torch.float32),
                                     (4233, torch.float32), (8200, torch.float32), (32768, torch.float32),
                                     (4232, torch.float16), (40000, torch.bfloat16)])
def test_plus_inf_logit_gives_nan_loss(V, dtype):
    """A +inf logit (fp16 logits overflow to it at 65504) has no log-softmax: torch.log_softmax, hence the
    upstream CPU path, returns NaN for the whole row and the cost is NaN — also when the +inf sits on a
    logit that is neither the blank nor the label, where a finite cost with a NaN gradient would slip
    past the reference's `torch.isnan(loss)` guard (bin/train.py:36-41) into the parameters.  Every K1
    shape: small rows, one wave per node (aligned and not), two waves per node, the streaming kernel."""
    acts, labels, al, ll = make_case(2, 6, 3, V, seed=47 + V, ragged=False)
    labels = np.minimum(labels, V - 2).astype(np.int32)
    acts[1, 2, 1, V - 1] = np.inf                     # never blank (0), never a label (< V-1)
    ref = oracle.rnnt_loss_c(acts, labels, al, ll)[0]
    assert np.isfinite(ref[0]) and np.isnan(ref[1])   # the CPU restatement agrees
    loss, grads = run_gpu(acts, labels, al, ll, "none", np.ones(2, np.float32), dtype=dtype)
    assert np.isfinite(loss[0]) and np.isnan(loss[1])
    assert np.isfinite(grads[0]).all()
    loss_mean, _ = run_gpu(acts, labels, al, ll, "mean", dtype=dtype)
    assert np.isnan(loss_mean[0])


def test_plus_inf_logit_generic_kernel():
    """The element-granular K1 (logits whose base pointer is off a 16-byte boundary with an odd row size)."""
    import rnnt_amd

    acts, labels, al, ll = make_case(3, 5, 2, 4233, seed=78, ragged=False)
    labels = np.minimum(labels, 4000).astype(np.int32)
    acts[2, 1, 1, 4232] = np.inf
    dev = torch.device("cuda:0")
    a = torch.from_numpy(acts).to(dev)[1:]
    assert a.data_ptr() % 16 != 0
    loss = rnnt_amd.RNNTLoss(reduction="none")(a, torch.from_numpy(labels[1:]).to(dev),
                                               torch.from_numpy(al[1:]).to(dev), torch.from_numpy(ll[1:]).to(dev))
    loss = loss.cpu().numpy()
    assert np.isfinite(loss[0]) and np.isnan(loss[1])


@pytest.mark.parametrize("B,T,U,V,dtype", [(64, 12, 5, 4232, torch.float32), (130, 9, 4, 520, torch.float32),
                                           (2, 6, 600, 8, torch.float32), (33, 17, 6, 4234, torch.bfloat16),
                                           (5, 30, 7, 29, torch.float16)])
def test_repeated_launches_give_the_same_bits(B, T, U, V, dtype):
    """No atomics, fixed summation orders, counted waits: 150 forward+backward launches of one ragged batch (compacted
    node order, wide lattice, rows that are not 16-byte aligned, small rows) return identical costs and gradients."""
    import rnnt_amd

    acts, labels, al, ll = make_case(B, T, U, V, seed=B + V, ragged=True, min_U=U // 2)
    dev = torch.device("cuda:0")
    a = torch.from_numpy(acts).to(dev).to(dtype).requires_grad_(True)
    lab, ald, lld = (torch.from_numpy(x).to(dev) for x in (labels, al, ll))
    crit = rnnt_amd.RNNTLoss(reduction="none", check_lengths=False)
    first = None
    for i in range(150):
        a.grad = None
        c = crit(a, lab, ald, lld)
        c.sum().backward()
        if first is None:
            first = (c.detach().clone(), a.grad.clone())
        else:
            assert torch.equal(c.detach(), first[0]) and torch.equal(a.grad, first[1]), f"launch {i} differs"
