"""Randomised shapes against the CPU oracle: every kernel shape (small rows with 1..16 chunks per lane,
one wave per node with and without 16-byte aligned rows, several waves per node, the streaming
wide-row kernel, the element-granular fallback) meets ragged lengths, arbitrary blanks, all three
dtypes and all three reductions.  Seeds are fixed; tolerances are the ones of test_gpu_parity.py."""
import numpy as np
import pytest
import torch

import oracle
from cases import assert_grad_parity, make_case

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
VOCABS = [1, 2, 3, 4, 5, 8, 11, 16, 27, 29, 31, 32, 33, 60, 64, 65, 100, 127, 128, 129, 200, 255, 256, 257, 300, 400,
          511, 512, 513, 600, 777, 1000, 1023, 1024, 1025, 1500, 2048, 2051, 3000, 4232, 4233, 4234, 5455, 8192, 9001,
          17408, 17409, 20000, 33000]


@pytest.mark.parametrize("seed", range(160))
def test_random_shape_against_oracle(seed):
    import rnnt_amd

    rng = np.random.default_rng(10_000 + seed)
    V = int(VOCABS[seed % len(VOCABS)] if seed < 2 * len(VOCABS) else rng.integers(1, 700))
    big = V > 6000
    B = int(rng.integers(1, 3 if big else 6))
    T = int(rng.integers(1, 9 if big else 41))
    U = int(rng.integers(0, 4 if big else 13)) if V > 1 else 0
    dtype = [torch.float32, torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 4))]
    blank = int(rng.integers(0, V))
    reduction = ["mean", "sum", "none"][int(rng.integers(0, 3))]
    acts, labels, al, ll = make_case(B, T, U, V, seed=seed, ragged=True, sigma=float(rng.choice([1.0, 2.5])),
                                     blank=blank) if V > 1 else (
        rng.standard_normal((B, T, 1, 1)).astype(np.float32), np.zeros((B, 0), np.int32),
        rng.integers(1, T + 1, size=B).astype(np.int32), np.zeros(B, np.int32))
    if V == 1:
        al[0] = T
    q = torch.from_numpy(acts).to(dtype).float().numpy()          # what the device really sees
    go = 0.5 if reduction != "none" else np.linspace(0.25, 1.5, B).astype(np.float32)

    a = torch.from_numpy(q).to(DEV).to(dtype).requires_grad_(True)
    loss = rnnt_amd.RNNTLoss(blank=blank, reduction=reduction)(
        a, torch.from_numpy(labels).to(DEV), torch.from_numpy(al).to(DEV), torch.from_numpy(ll).to(DEV))
    loss.backward(torch.as_tensor(go, dtype=torch.float32, device=DEV).reshape(loss.shape))
    got_loss = loss.detach().cpu().numpy()
    got_g = a.grad.float().cpu().numpy()

    c, g, _ = oracle.rnnt_loss_c(q, labels, al, ll, blank=blank)
    ref_loss, ref_g = oracle.reduce_like_upstream(c, g, reduction, go)
    np.testing.assert_allclose(got_loss, ref_loss, rtol=1e-5, atol=1e-6)
    eps = {torch.float32: 0.0, torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
    if eps == 0.0:
        _, g64, _ = oracle.rnnt_loss_c(q, labels, al, ll, blank=blank, dtype=np.float64)
        _, ref_g64 = oracle.reduce_like_upstream(c, g64, reduction, go)
        assert_grad_parity(got_g, ref_g, ref_g64, reduction=reduction,
                           label=f"fuzz seed={seed} B={B} T={T} U={U} V={V} blank={blank}")
    else:
        assert np.all(np.abs(got_g - ref_g) <= 1e-4 + eps * np.abs(ref_g))
    for b in range(B):                                             # padding is exactly zero
        assert np.all(got_g[b, al[b]:] == 0) and np.all(got_g[b, :, ll[b] + 1:] == 0)
