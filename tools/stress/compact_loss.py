"""One-off: rnnt_loss_compact against rnnt_loss on random ragged batches, every K1/K3 shape."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch, rnnt_amd
from cases import make_case
from test_gpu_compact_loss import host_rowmap, same
DEV = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(99)
for it in range(N):
    B = int(rng.integers(1, 7)); T = int(rng.integers(1, 60)); U = int(rng.integers(0, 12))
    V = int(rng.choice([2, 3, 17, 29, 64, 128, 129, 255, 512, 600, 1000, 1027, 2048, 4232, 4233, 4234, 5455, 8200, 9000, 20001, 40000, int(rng.integers(2, 6000))]))
    dtype = [torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 3))]
    if V * B * T * (U + 1) > 6e7:
        T = max(1, int(6e7 / (V * B * (U + 1))))
    acts_np, labels, al, ll = make_case(B, T, U, V, seed=7000 + it, ragged=True)
    if rng.random() < 0.25:
        al = al.copy(); al[int(rng.integers(0, B))] = T + 1
    acts = torch.from_numpy(acts_np).to(dtype).to(DEV).requires_grad_(True)
    lab, ald, lld = (torch.from_numpy(x).to(DEV) for x in (labels, al, ll))
    fit = torch.from_numpy((al >= 1) & (al <= T)).to(DEV)
    go = torch.from_numpy(rng.uniform(0.2, 1.5, B).astype(np.float32)).to(DEV) * fit
    costs = rnnt_amd.rnnt_loss(acts, lab, ald, lld, reduction="none", check_lengths=False)
    costs.backward(go)
    rows = host_rowmap(al, ll, T, U + 1)
    n = len(rows)
    what = (it, B, T, U, V, str(dtype), n)
    if n == 0:
        continue
    assert n == rnnt_amd.count_lattice_nodes(al, ll, T, U + 1)
    slack = int(rng.choice([0, 0, 3]))
    rowmap = rnnt_amd.compact_rowmap(ald, lld, T, U + 1, n)
    assert np.array_equal(rowmap.cpu().numpy(), rows), what
    idx = rowmap.long()
    nodes = torch.cat((acts.detach().view(-1, V).index_select(0, idx), torch.full((slack, V), float("nan"), dtype=dtype, device=DEV))).contiguous().requires_grad_(True)
    cc = rnnt_amd.rnnt_loss_compact(nodes, lab, ald, lld, T, reduction="none", exact_rows=slack == 0)
    exact = (V * acts.element_size()) % 128 == 0 and V * acts.element_size() > 2048
    ok = torch.isfinite(costs)
    assert torch.equal(torch.isnan(cc), torch.isnan(costs)), what
    assert same(cc[ok], costs[ok], exact), (what, cc, costs)
    cc.backward(go)
    ref = acts.grad.view(-1, V).index_select(0, idx)
    fin = torch.isfinite(ref).all(1)
    assert same(nodes.grad[:n][fin], ref[fin], exact), what
    if slack:
        assert float(nodes.grad[n:].float().abs().sum()) == 0.0, what
    if it % 20 == 0:
        print("ok", what, flush=True)
print("all", N, "passed")
