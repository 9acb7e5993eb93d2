"""One-off: the loop of tests/test_gpu_fused_compact.py::test_compact_random_shapes over many more seeds."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import test_gpu_fused_compact as t
from test_gpu_fused import DEV, forward_c_abi
from test_gpu_fused_bwd import check_against_k3, k3_c_abi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(2025)
for it in range(N):
    hidden, weight, bias, labels, al, ll, rows, n, cap, ld, V, dtype = t._random_compact_case(rng, 5000 + it)
    B, T, U1, H = hidden.shape
    what = (it, B, T, U1 - 1, V, n, cap, ld)
    st, dense, dense_costs, _ = forward_c_abi(hidden, weight, bias, labels, al, ll, ld=ld)
    assert st == 0, what
    st, logits, costs, ws, cws = t.forward_compact(t.poison_padding(hidden, al, ll), weight, bias, labels, al, ll, ld, cap)
    assert st == 0, what
    idx = torch.from_numpy(rows).to(DEV)
    assert torch.equal(logits[:n, :V].contiguous().view(torch.int16), dense.view(-1, ld)[idx][:, :V].contiguous().view(torch.int16)), what
    np.testing.assert_allclose(costs.cpu().numpy(), dense_costs.cpu().numpy(), rtol=2e-6, atol=0, equal_nan=True, err_msg=str(what))
    if n == 0:
        continue
    full = torch.zeros(B * T * U1, V, dtype=dtype, device=DEV)
    full[idx] = logits[:n, :V]
    go = rng.uniform(0.2, 1.5, B).astype(np.float32)
    fit = (al >= 1) & (al <= T)
    go[~fit] = 0.0
    g = k3_c_abi(full.view(B, T, U1, V).contiguous(), labels, al, ll, ws, go, 1, 1.0)
    g = torch.where(torch.from_numpy(fit).to(DEV)[:, None, None, None], g, torch.zeros_like(g))
    st, dh, dw, db = t.backward_compact(t.poison_padding(hidden, al, ll), weight, logits, ld, cap, labels, al, ll, ws, cws, go, 1, 1.0)
    assert st == 0, what
    check_against_k3(hidden, weight, g, dh, dw, db)
    if it % 20 == 0:
        print("ok", it, what, flush=True)
print("all", N, "passed")
