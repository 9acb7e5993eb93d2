"""One-off: rnnt_amd.joint_hidden(..., nodes=n) (the compacted hidden layer) against the dense pair on random shapes,
forward and backward, bit for bit.  Run from the repository root on the GPU box: python tools/stress/hidden_compact.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import rnnt_amd
DEV = torch.device("cuda:0")
rng = np.random.default_rng(77)
bad = 0
for it in range(200):
    B = int(rng.integers(1, 70)); T = int(rng.integers(1, 90)); U1 = int(rng.integers(1, 20))
    H = int(rng.choice([1, 7, 64, 320, 512, 513, 640, 1024]))
    dtype = [torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 3))]
    al = rng.integers(1, T + 1, size=B).astype(np.int32); ll = rng.integers(0, U1, size=B).astype(np.int32)
    if rng.random() < 0.3: al[int(rng.integers(0, B))] = T + 3
    if rng.random() < 0.2: ll[int(rng.integers(0, B))] = U1
    rows = []
    for b in range(B):
        if 1 <= al[b] <= T and 0 <= ll[b] < U1:
            for t in range(al[b]): rows.extend((b * T + t) * U1 + u for u in range(ll[b] + 1))
    n = len(rows)
    if n == 0: continue
    assert n == rnnt_amd.count_lattice_nodes(al, ll, T, U1)
    rows = torch.tensor(rows, device=DEV)
    enc = torch.randn(B, T, H, device=DEV).to(dtype); dec = torch.randn(B, U1, H, device=DEV).to(dtype)
    ald, lld = torch.from_numpy(al).to(DEV), torch.from_numpy(ll).to(DEV)
    e1, d1 = enc.clone().requires_grad_(True), dec.clone().requires_grad_(True)
    dense = rnnt_amd.joint_hidden(e1, d1)
    e2, d2 = enc.clone().requires_grad_(True), dec.clone().requires_grad_(True)
    nodes = rnnt_amd.joint_hidden(e2, d2, ald, lld, nodes=n)
    ok = torch.equal(nodes, dense.view(-1, H)[rows])
    go = torch.randn(n, H, device=DEV).to(dtype)
    full = torch.zeros(B * T * U1, H, dtype=dtype, device=DEV); full[rows] = go
    dense.backward(full.view(B, T, U1, H)); nodes.backward(go)
    ok = ok and torch.equal(e2.grad, e1.grad) and torch.equal(d2.grad, d1.grad)
    if not ok:
        bad += 1; print("MISMATCH", it, B, T, U1, H, dtype, flush=True)
    if it % 40 == 0: print("it", it, "ok", flush=True)
print("done, mismatches:", bad)
sys.exit(1 if bad else 0)
