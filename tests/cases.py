"""Seeded synthetic RNN-T lattices shared by the tests (numpy, CPU)."""
import numpy as np


def make_case(B, T, U, V, seed=0, ragged=True, sigma=1.0, peaked=False, blank=0, min_T=1, min_U=0):
    """acts [B,T,U+1,V] f32, labels [B,U] i32 (never blank), act_lens, label_lens.

    One utterance always has (T, U) so the tensor dims equal the maxima, as the reference's
    collation guarantees (/root/reference/src/data/dataset.py:173-179).
    """
    rng = np.random.default_rng(seed)
    acts = (rng.standard_normal((B, T, U + 1, V)) * sigma).astype(np.float32)
    choices = [v for v in range(V) if v != blank]
    labels = rng.choice(choices, size=(B, max(U, 0))).astype(np.int32) if U > 0 else np.zeros((B, 0), np.int32)
    if ragged:
        act_lens = rng.integers(min_T, T + 1, size=B).astype(np.int32)
        label_lens = rng.integers(min_U, U + 1, size=B).astype(np.int32)
        act_lens[rng.integers(0, B)] = T
        label_lens[rng.integers(0, B)] = U
    else:
        act_lens = np.full(B, T, np.int32)
        label_lens = np.full(B, U, np.int32)
    for b in range(B):
        labels[b, label_lens[b]:] = 0  # zero padding as the reference's collate does
    if peaked:
        # trained-model-like: +8 on blank at 90 % of nodes, on the next label elsewhere
        pick = rng.random((B, T, U + 1)) < 0.9
        for b in range(B):
            for u in range(U + 1):
                y = labels[b, u] if u < label_lens[b] else blank
                acts[b, :, u, blank] += 8.0 * pick[b, :, u]
                acts[b, :, u, y] += 8.0 * (~pick[b, :, u])
    return acts, labels, act_lens, label_lens


_KAT1_ACTS = [[[[0.1, 0.6, 0.1, 0.1, 0.1], [0.1, 0.1, 0.6, 0.1, 0.1], [0.1, 0.1, 0.2, 0.8, 0.1]],
               [[0.1, 0.6, 0.1, 0.1, 0.1], [0.1, 0.1, 0.2, 0.1, 0.1], [0.7, 0.1, 0.2, 0.1, 0.1]]]]
# Known-answer vector recalled from the upstream package's public unit test.
KAT1 = dict(
    acts=np.array(_KAT1_ACTS, dtype=np.float32),
    acts64=np.array(_KAT1_ACTS, dtype=np.float64),
    labels=np.array([[1, 2]], dtype=np.int32),
    act_lens=np.array([2], dtype=np.int32),
    label_lens=np.array([2], dtype=np.int32),
    cost=4.495666773770733,
    grads=np.array([[[[-0.13116686, -0.39992680, 0.17703122, 0.17703122, 0.17703122],
                      [-0.18572753, 0.12247054, -0.18168408, 0.12247054, 0.12247054],
                      [-0.32091246, 0.06269139, 0.06928471, 0.12624497, 0.06269139]],
                     [[0.05456068, -0.21824272, 0.05456068, 0.05456068, 0.05456068],
                      [0.12073957, 0.12073957, -0.48295828, 0.12073957, 0.12073957],
                      [-0.69258820, 0.16871117, 0.18645468, 0.16871117, 0.16871117]]]], dtype=np.float64),
)


def assert_grad_parity(g_gpu, g_ref32, g_ref64=None, atol=1e-4):
    """North-star bar: |g - g_cpu32| <= 1e-4 absolute.  Where |ll| is in the thousands one fp32 ulp of
    alpha+beta-ll is ~1e-4, so O(1) gradients of two fp32 implementations differ by ~1e-4 RELATIVE
    (the CPU oracle's own error against float64 is that large).  In that regime, and only when a
    float64 reference is supplied, the bar is: within 1e-4 + 1e-3|g| of the fp32 oracle AND no
    further from float64 than twice the fp32 oracle's own distance."""
    err = np.abs(g_gpu - g_ref32)
    if err.max() <= atol:
        return
    assert g_ref64 is not None, f"max |dg| = {err.max():.3e} > {atol}"
    assert np.all(err <= atol + 1e-3 * np.abs(g_ref32)), f"max |dg| = {err.max():.3e}"
    e_gpu, e_cpu = np.abs(g_gpu - g_ref64).max(), np.abs(g_ref32 - g_ref64).max()
    assert e_gpu <= max(2 * e_cpu, atol), f"gpu {e_gpu:.3e} vs cpu-f32 {e_cpu:.3e} from float64"
