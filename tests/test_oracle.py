"""CPU tests of the oracle itself (no GPU): the three independent statements of the loss agree,
the upstream known-answer vector reproduces, the golden fixtures are stable, and the domain
properties SURVEY.md §4 lists hold.  PARITY UNPINNED w.r.t. the real warp-transducer (no reference
fixtures exist); KAT-1 and KAT-2 (upstream's public unit-test vectors, recalled) are the anchors."""
import glob
import os

import numpy as np
import pytest

import oracle
from cases import KAT1, KAT2, make_case

GOLDEN = sorted(f for f in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
                if not os.path.basename(f).startswith("ref_"))   # ref_*: the joint / decoding fixtures (ref_fixtures.py)


def test_kat1_all_statements():
    a, lab, al, ll = KAT1["acts"], KAT1["labels"], KAT1["act_lens"], KAT1["label_lens"]
    c32, g32, llb = oracle.rnnt_loss_c(a, lab, al, ll)
    a64 = KAT1["acts64"]
    c64, g64, _ = oracle.rnnt_loss_c(a64, lab, al, ll, dtype=np.float64)
    cdp, gdp = oracle.rnnt_loss_dp_f64(a64, lab, al, ll)
    cbf = oracle.rnnt_loss_bruteforce(a64, lab, al, ll)
    assert abs(c64[0] - KAT1["cost"]) < 1e-12
    assert abs(cdp[0] - KAT1["cost"]) < 1e-12
    assert abs(cbf[0] - KAT1["cost"]) < 1e-12
    assert abs(c32[0] - 4.4956668) < 1e-6          # the value upstream's unit test expects
    assert abs(-llb[0] - c32[0]) < 1e-5            # beta(0,0) == ll
    assert np.abs(g64 - KAT1["grads"]).max() < 1e-8
    assert np.abs(gdp - KAT1["grads"]).max() < 1e-8
    assert np.abs(g32 - KAT1["grads"]).max() < 1e-6


def test_kat2_two_utterances_all_statements():
    lab, al, ll = KAT2["labels"], KAT2["act_lens"], KAT2["label_lens"]
    a64 = KAT2["acts64"]
    c64, g64, _ = oracle.rnnt_loss_c(a64, lab, al, ll, dtype=np.float64)
    cdp, gdp = oracle.rnnt_loss_dp_f64(a64, lab, al, ll)
    cbf = oracle.rnnt_loss_bruteforce(a64, lab, al, ll)
    for c in (c64, cdp, cbf):
        assert np.abs(np.asarray(c) - KAT2["costs"]).max() < 1e-12   # upstream's 17-digit expected costs
    assert np.abs(g64 - gdp).max() < 1e-12
    # upstream's expected gradient (recalled part) was produced in float32: it sits 3e-9..7e-8 from float64
    assert np.abs(g64[0, 0] - KAT2["grads_b0_t0"]).max() < 2e-7
    c32, g32, _ = oracle.rnnt_loss_c(KAT2["acts"], lab, al, ll)
    np.testing.assert_allclose(c32, KAT2["costs"], rtol=1e-6)
    assert np.abs(g32 - g64).max() < 1e-6
    # 'mean' as the reference's RNNTLoss() default: (c0 + c1) / 2
    loss, _ = oracle.reduce_like_upstream(c32, g32, "mean", 1.0)
    assert abs(loss[0] - KAT2["costs"].mean()) < 1e-6


@pytest.mark.parametrize("seed", range(12))
def test_c_oracle_vs_autograd_dp_and_bruteforce(seed):
    rng = np.random.default_rng(seed)
    B, T, U, V = int(rng.integers(1, 4)), int(rng.integers(1, 6)), int(rng.integers(0, 4)), int(rng.integers(2, 9))
    a, lab, al, ll = make_case(B, T, U, V, seed=seed, ragged=True, sigma=float(rng.choice([1.0, 3.0])))
    c64, g64, llb = oracle.rnnt_loss_c(a, lab, al, ll, dtype=np.float64)
    cdp, gdp = oracle.rnnt_loss_dp_f64(a, lab, al, ll)
    cbf = oracle.rnnt_loss_bruteforce(a, lab, al, ll)
    np.testing.assert_allclose(c64, cdp, rtol=1e-12)
    np.testing.assert_allclose(c64, cbf, rtol=1e-12)
    np.testing.assert_allclose(-llb, c64, rtol=1e-12)
    assert np.abs(g64 - gdp).max() < 1e-12
    # fp32 instantiation against fp64: the tolerance the GPU is later held to against fp32
    c32, g32, _ = oracle.rnnt_loss_c(a, lab, al, ll)
    np.testing.assert_allclose(c32, c64, rtol=1e-5)
    assert np.abs(g32 - g64).max() < 1e-4


@pytest.mark.parametrize("seed", range(50, 56))
def test_c_oracle_medium_ragged_f32_vs_f64(seed):
    a, lab, al, ll = make_case(4, 40, 12, 64, seed=seed, ragged=True, sigma=2.0)
    c64, g64, _ = oracle.rnnt_loss_c(a, lab, al, ll, dtype=np.float64)
    c32, g32, _ = oracle.rnnt_loss_c(a, lab, al, ll)
    np.testing.assert_allclose(c32, c64, rtol=1e-5)
    assert np.abs(g32 - g64).max() < 1e-4


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_golden_fixtures(path):
    z = np.load(path)  # allow_pickle=False (default)
    blank = int(z["blank"])
    c64, g64, _ = oracle.rnnt_loss_c(z["acts"], z["labels"], z["act_lens"], z["label_lens"], blank=blank,
                                     dtype=np.float64)
    np.testing.assert_allclose(c64, z["costs"], rtol=1e-9)
    assert np.abs(g64 - z["grads"]).max() < 1e-8
    c32, g32, _ = oracle.rnnt_loss_c(z["acts"], z["labels"], z["act_lens"], z["label_lens"], blank=blank)
    np.testing.assert_allclose(c32, z["costs"], rtol=1e-5)
    assert np.abs(g32 - z["grads"]).max() < 1e-4


def test_golden_set_is_complete():
    names = {os.path.basename(p) for p in GOLDEN}
    assert {"kat1.npz", "kat2.npz", "tiny_T1_U0.npz", "ragged_small.npz", "wide_lattice_U70.npz"} <= names


# ---- properties -------------------------------------------------------------------
def test_grads_sum_to_zero_and_padding_is_zero():
    a, lab, al, ll = make_case(3, 9, 5, 13, seed=7, ragged=True)
    _, g, _ = oracle.rnnt_loss_c(a, lab, al, ll, dtype=np.float64)
    for b in range(3):
        T, U = al[b], ll[b]
        assert np.abs(g[b, :T, :U + 1].sum(-1)).max() < 1e-12     # softmax gradient: rows sum to 0
        assert np.all(g[b, T:] == 0) and np.all(g[b, :, U + 1:] == 0)


def test_shift_invariance_per_node():
    a, lab, al, ll = make_case(2, 6, 3, 9, seed=8, ragged=True)
    shift = np.random.default_rng(1).standard_normal(a.shape[:3] + (1,)).astype(np.float32) * 3
    c0, g0, _ = oracle.rnnt_loss_c(a, lab, al, ll, dtype=np.float64)
    c1, g1, _ = oracle.rnnt_loss_c(a.astype(np.float64) + shift, lab, al, ll, dtype=np.float64)
    np.testing.assert_allclose(c0, c1, rtol=1e-10)
    assert np.abs(g0 - g1).max() < 1e-10


def test_batch_permutation_and_unpadded_equivalence():
    a, lab, al, ll = make_case(4, 8, 4, 10, seed=9, ragged=True)
    c, g, _ = oracle.rnnt_loss_c(a, lab, al, ll, dtype=np.float64)
    perm = np.array([2, 0, 3, 1])
    cp, gp, _ = oracle.rnnt_loss_c(a[perm], lab[perm], al[perm], ll[perm], dtype=np.float64)
    np.testing.assert_array_equal(cp, c[perm])
    np.testing.assert_array_equal(gp, g[perm])
    for b in range(4):  # ragged == per-utterance unpadded
        T, U = al[b], ll[b]
        cb, gb, _ = oracle.rnnt_loss_c(a[b:b + 1, :T, :U + 1], lab[b:b + 1, :U], [T], [U], dtype=np.float64)
        assert cb[0] == c[b]
        np.testing.assert_array_equal(gb[0], g[b, :T, :U + 1])


def test_padding_content_is_ignored():
    a, lab, al, ll = make_case(3, 7, 4, 8, seed=10, ragged=True)
    c, g, _ = oracle.rnnt_loss_c(a, lab, al, ll)
    poisoned = a.copy()
    for b in range(3):
        poisoned[b, al[b]:] = np.nan
        poisoned[b, :, ll[b] + 1:] = np.nan
    cp, gp, _ = oracle.rnnt_loss_c(poisoned, lab, al, ll)
    np.testing.assert_array_equal(c, cp)
    np.testing.assert_array_equal(g, gp)


def test_finite_differences():
    a, lab, al, ll = make_case(1, 3, 2, 4, seed=11, ragged=False)
    a = a.astype(np.float64)
    _, g, _ = oracle.rnnt_loss_c(a, lab, al, ll, dtype=np.float64)
    eps = 1e-6
    rng = np.random.default_rng(0)
    for _ in range(10):
        idx = tuple(int(rng.integers(0, s)) for s in a.shape)
        ap, am = a.copy(), a.copy()
        ap[idx] += eps
        am[idx] -= eps
        cp, _, _ = oracle.rnnt_loss_c(ap, lab, al, ll, dtype=np.float64, want_grads=False)
        cm, _, _ = oracle.rnnt_loss_c(am, lab, al, ll, dtype=np.float64, want_grads=False)
        assert abs((cp[0] - cm[0]) / (2 * eps) - g[idx]) < 1e-6


def test_reductions_like_upstream():
    a, lab, al, ll = make_case(3, 5, 2, 6, seed=12, ragged=True)
    c, g, _ = oracle.rnnt_loss_c(a, lab, al, ll)
    m, gm = oracle.reduce_like_upstream(c, g, "mean", 0.5)
    s, gs = oracle.reduce_like_upstream(c, g, "sum", 2.0)
    n, gn = oracle.reduce_like_upstream(c, g, "none", np.array([1.0, 2.0, 3.0], np.float32))
    assert m.shape == (1,) and s.shape == (1,) and n.shape == (3,)
    np.testing.assert_allclose(m[0], c.sum() / 3, rtol=1e-6)
    np.testing.assert_allclose(gm, g * 0.5 / 3, rtol=1e-6)
    np.testing.assert_allclose(gs, g * 2.0, rtol=1e-6)
    np.testing.assert_allclose(gn[2], g[2] * 3.0, rtol=1e-6)


def test_invalid_arguments_are_rejected():
    a, lab, al, ll = make_case(2, 4, 2, 5, seed=13, ragged=False)
    with pytest.raises(ValueError):
        oracle.rnnt_loss_c(a, lab, [5, 4], ll)          # act_len > T
    with pytest.raises(ValueError):
        oracle.rnnt_loss_c(a, lab, al, [3, 2])          # label_len + 1 > U+1
    bad = lab.copy()
    bad[0, 0] = 5
    with pytest.raises(ValueError):
        oracle.rnnt_loss_c(a, bad, al, ll)              # label >= V


def test_oracle_under_sanitizers():
    """the oracle's C source, instrumented with AddressSanitizer and UBSan (host code only),
    runs both known-answer vectors and a ragged batch whose padding holds NaN (oracle/selftest.c)."""
    import subprocess

    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    subprocess.check_call(["make", "-C", here, "selftest_asan"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0:detect_leaks=0", OMP_NUM_THREADS="3")
    out = subprocess.run([os.path.join(here, "selftest_asan")], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "oracle selftest ok" in out.stdout


@pytest.mark.parametrize("reduction", ["mean", "sum", "none"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_upstream_shaped_path_equals_the_all_c_one(reduction, dtype):
    """torch.log_softmax -> native alpha/beta + sparse gradient -> autograd (how upstream's CPU path is shaped,
    bench.py's second CPU baseline) against the all-C statement: same costs and gradients up to rounding, on the
    two known answers and a ragged batch with padding."""
    k1 = oracle.rnnt_loss_upstream_shaped(KAT1["acts"].astype(dtype), KAT1["labels"], KAT1["act_lens"],
                                          KAT1["label_lens"], reduction="sum")
    np.testing.assert_allclose(k1[0][0], KAT1["cost"], rtol=1e-6)
    assert np.abs(k1[1] - KAT1["grads"]).max() <= 1e-6
    acts, labels, al, ll = make_case(5, 13, 6, 31, seed=11, ragged=True, sigma=2.0)
    acts = acts.astype(dtype)
    go = 0.5 if reduction != "none" else np.linspace(0.25, 1.5, 5).astype(dtype)
    loss, grads = oracle.rnnt_loss_upstream_shaped(acts, labels, al, ll, reduction=reduction, grad_output=go)
    c, g, _ = oracle.rnnt_loss_c(acts, labels, al, ll, dtype=dtype)
    ref_loss, ref_g = oracle.reduce_like_upstream(c, g, reduction, go)
    tol = 1e-5 if dtype == np.float32 else 1e-12
    np.testing.assert_allclose(loss, ref_loss, rtol=tol)
    # fp32: torch's log_softmax and the C one round differently, and exp(alpha + beta - ll) carries that
    assert np.abs(grads - ref_g).max() <= (1e-5 if dtype == np.float32 else 1e-13)
    pad = np.ones(acts.shape[:3], bool)
    for b in range(5):
        pad[b, :al[b], :ll[b] + 1] = False
    assert not grads[pad].any()
