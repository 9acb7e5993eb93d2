"""One-off: the loop of tests/test_gpu_fuzz.py over wider shapes (V = 2 ... 40 000, B <= 7, T <= 59, U <= 19), 800 cases
against the C oracle at the suite's bars.  Run from the repository root: python tools/stress/loss_wide.py"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import oracle, rnnt_amd
from cases import assert_grad_parity, make_case
DEV = torch.device("cuda:0")
bad = 0
t0 = time.time()
for seed in range(5000, 5800):
    rng = np.random.default_rng(seed)
    kind = int(rng.integers(0, 3))
    V = int([rng.integers(2, 700), rng.integers(700, 6000), rng.integers(6000, 40000)][kind])
    big = V > 6000
    B = int(rng.integers(1, 3 if big else 8))
    T = int(rng.integers(1, 9 if big else 60))
    U = int(rng.integers(0, 4 if big else 20))
    dtype = [torch.float32, torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 4))]
    blank = int(rng.integers(0, V))
    reduction = ["mean", "sum", "none"][int(rng.integers(0, 3))]
    acts, labels, al, ll = make_case(B, T, U, V, seed=seed, ragged=True, sigma=float(rng.choice([1.0, 2.5])), blank=blank)
    q = torch.from_numpy(acts).to(dtype).float().numpy()
    go = 0.5 if reduction != "none" else np.linspace(0.25, 1.5, B).astype(np.float32)
    a = torch.from_numpy(q).to(DEV).to(dtype).requires_grad_(True)
    loss = rnnt_amd.RNNTLoss(blank=blank, reduction=reduction)(
        a, torch.from_numpy(labels).to(DEV), torch.from_numpy(al).to(DEV), torch.from_numpy(ll).to(DEV))
    loss.backward(torch.as_tensor(go, dtype=torch.float32, device=DEV).reshape(loss.shape))
    got_loss = loss.detach().cpu().numpy(); got_g = a.grad.float().cpu().numpy()
    c, g, _ = oracle.rnnt_loss_c(q, labels, al, ll, blank=blank)
    ref_loss, ref_g = oracle.reduce_like_upstream(c, g, reduction, go)
    try:
        np.testing.assert_allclose(got_loss, ref_loss, rtol=1e-5, atol=1e-6)
        eps = {torch.float32: 0.0, torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
        if eps == 0.0:
            _, g64, _ = oracle.rnnt_loss_c(q, labels, al, ll, blank=blank, dtype=np.float64)
            _, ref_g64 = oracle.reduce_like_upstream(c, g64, reduction, go)
            assert_grad_parity(got_g, ref_g, ref_g64, reduction=reduction, label=f"seed={seed}")
        else:
            assert np.all(np.abs(got_g - ref_g) <= 1e-4 + eps * np.abs(ref_g))
        for b in range(B):
            assert np.all(got_g[b, al[b]:] == 0) and np.all(got_g[b, :, ll[b] + 1:] == 0)
    except AssertionError as e:
        bad += 1
        print("FAIL seed", seed, B, T, U, V, dtype, reduction, str(e)[:200], flush=True)
    if seed % 100 == 0:
        print("seed", seed, "failures", bad, "%.0f s" % (time.time() - t0), flush=True)
print("done: 800 cases, failures", bad)
sys.exit(1 if bad else 0)
