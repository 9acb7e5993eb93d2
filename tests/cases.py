"""Seeded synthetic RNN-T lattices shared by the tests (numpy, CPU)."""
import json
import os

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

# KAT-2: upstream's two-utterance unit test (B2 T4 U2 V3, 17-digit inputs), likewise recalled.  The
# recall is validated by the answer itself: three independent float64 statements of the loss
# (tests/test_oracle.py) reproduce both expected costs to 1e-12, which no mis-remembered input
# digit could do.  Expected gradients: only the first node's three rows are recalled (8 digits).
_KAT2_ACTS = [
    [[[0.06535690384862791, 0.7875301411923206, 0.08159176605666074],
      [0.5297155426466327, 0.7506749639230854, 0.7541348379087998],
      [0.6097641124736383, 0.8681404965673826, 0.6225318186056529]],
     [[0.6685222872103057, 0.8580392805336061, 0.16453892311765583],
      [0.989779515236694, 0.944298460961015, 0.6031678586829663],
      [0.9467833543605416, 0.666202507295747, 0.28688179752461884]],
     [[0.09418426230195986, 0.3666735970751962, 0.736168049462793],
      [0.1666804425271342, 0.7141542198635192, 0.3993997272216727],
      [0.5359823524146038, 0.29182076440286386, 0.6126422611507932]],
     [[0.3242405528768486, 0.8007644367291621, 0.5241057606558068],
      [0.779194617063042, 0.18331417220174862, 0.113745182072432],
      [0.24022162381327106, 0.3394695622533106, 0.1341595066017014]]],
    [[[0.5055615569388828, 0.051597282072282646, 0.6402903936686337],
      [0.43073311517251, 0.8294731834714112, 0.1774668847323424],
      [0.3207001991262245, 0.04288308912457006, 0.30280282975568984]],
     [[0.6751777088333762, 0.569537369330242, 0.5584738347504452],
      [0.08313242153985256, 0.06016544344162322, 0.10795752845152584],
      [0.7486153608562472, 0.943918041459349, 0.4863558118797222]],
     [[0.4181986264486809, 0.6524078485043804, 0.024242983423721887],
      [0.13458171554507403, 0.3663418070512402, 0.2958297395361563],
      [0.9236695822497084, 0.6899291482654177, 0.7418981733448822]],
     [[0.25000547599982104, 0.6034295486281007, 0.9872887878887768],
      [0.5926057265215715, 0.8846724004467684, 0.5434495396894328],
      [0.6607698886038497, 0.3771277082495921, 0.3580209022231813]]]]
KAT2 = dict(
    acts=np.array(_KAT2_ACTS, dtype=np.float32),
    acts64=np.array(_KAT2_ACTS, dtype=np.float64),
    labels=np.array([[1, 2], [1, 1]], dtype=np.int32),
    act_lens=np.array([4, 4], dtype=np.int32),
    label_lens=np.array([2, 2], dtype=np.int32),
    costs=np.array([4.2806528590890736, 3.9384369822503591]),
    grads_b0_t0=np.array([[-1.86843902e-01, -6.25548810e-02, 2.49398798e-01],
                          [-2.03376666e-01, 2.02399328e-01, 9.77333169e-04],
                          [-1.41016081e-01, 7.91234672e-02, 6.18926100e-02]]),
)


RELAXED_LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out",
                           "relaxed_grad_parity.jsonl")


def assert_grad_parity(g_gpu, g_ref32, g_ref64=None, atol=1e-4, reduction="mean", label=""):
    """North-star bar: |g - g_cpu32| <= 1e-4 absolute.  For reduction='mean' — the only mode the
    reference uses (/root/reference/src/rnnt/model.py:87) — that bar is strict and nothing else is accepted.

    For the UNSCALED gradients of 'sum' / 'none' on lattices where |ll| is in the thousands, one fp32 ulp of
    alpha+beta-ll is ~1e-4, so O(1) gradients of two fp32 implementations differ by ~1e-4 RELATIVE
    (the CPU oracle's own error against float64 is that large).  There, and only when a float64
    reference is supplied, the bar is: within 1e-4 + 1e-3|g| of the fp32 oracle AND no further from
    float64 than twice the fp32 oracle's own distance.  Every use of that branch is appended to
    gpurun_out/relaxed_grad_parity.jsonl (copied to profiles/ per round) so the list can be audited."""
    err = np.abs(g_gpu - g_ref32)
    if err.max() <= atol:
        return
    assert reduction != "mean", f"{label}: reduction='mean' must meet the strict bar, max |dg| = {err.max():.3e} > {atol}"
    assert g_ref64 is not None, f"{label}: max |dg| = {err.max():.3e} > {atol}"
    e_gpu, e_cpu = float(np.abs(g_gpu - g_ref64).max()), float(np.abs(g_ref32 - g_ref64).max())
    try:
        os.makedirs(os.path.dirname(RELAXED_LOG), exist_ok=True)
        with open(RELAXED_LOG, "a") as f:
            f.write(json.dumps({"case": label, "reduction": reduction, "shape": list(g_gpu.shape),
                                "max_abs_dg_vs_f32_oracle": float(err.max()), "max_abs_g": float(np.abs(g_ref32).max()),
                                "gpu_vs_f64": e_gpu, "f32_oracle_vs_f64": e_cpu}) + "\n")
    except OSError:
        pass
    assert np.all(err <= atol + 1e-3 * np.abs(g_ref32)), f"{label}: max |dg| = {err.max():.3e}"
    assert e_gpu <= max(2 * e_cpu, atol), f"{label}: gpu {e_gpu:.3e} vs cpu-f32 {e_cpu:.3e} from float64"
