"""compute_rnnt_loss_static (the entry point that allocates nothing, include/rnnt_mi355.h) against
compute_rnnt_loss on the same inputs, through the C ABI: bit-identical costs and gradients, on two
calls in a row (the second on a smaller minibatch, so the device array still holds the first call's
lattice), for every logit type; and the size limit is refused with its own status before any launch."""
import ctypes

import numpy as np
import pytest
import torch

from cases import make_case

pytestmark = pytest.mark.gpu


def _call(lib, fn_name, acts, labels, act_lens, label_lens, dtype_code, with_ws):
    from rnnt_amd import _lib

    B, T, U1, V = acts.shape
    dev = acts.device
    grads = torch.full_like(acts, float("nan"))
    costs = torch.full((B,), float("nan"), dtype=torch.float32, device=dev)
    opts = _lib.RnntOptions(torch.cuda.current_stream(dev).cuda_stream, 0, T, U1, dtype_code, None)
    args = [acts.data_ptr(), grads.data_ptr(), labels.data_ptr(), label_lens.data_ptr(), act_lens.data_ptr(),
            V, B, costs.data_ptr()]
    if with_ws:
        ws = torch.empty(_lib.workspace_bytes(T, U1, B), dtype=torch.uint8, device=dev)
        args.append(ws.data_ptr())
    status = getattr(lib, fn_name)(*args, opts)
    torch.cuda.synchronize(dev)
    return status, costs.cpu().numpy(), grads.float().cpu().numpy()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_static_workspace_matches_compute_rnnt_loss_twice_in_a_row(dtype):
    from rnnt_amd import _lib
    from rnnt_amd.loss import _DTYPES

    lib = _lib.load()
    dev = torch.device("cuda:0")
    # first a larger ragged minibatch, then a smaller one that lands inside the first one's lattice
    for (B, T, U, V, seed) in [(6, 61, 17, 515, 11), (3, 23, 5, 515, 12)]:
        a, l, al, ll = make_case(B, T, U, V, seed=seed, ragged=True)
        acts = torch.from_numpy(a).to(dev).to(dtype).contiguous()
        tl, tal, tll = (torch.from_numpy(x).to(dev) for x in (l, al, ll))
        s0, c0, g0 = _call(lib, "compute_rnnt_loss", acts, tl, tal, tll, _DTYPES[dtype], True)
        s1, c1, g1 = _call(lib, "compute_rnnt_loss_static", acts, tl, tal, tll, _DTYPES[dtype], False)
        assert s0 == 0 and s1 == 0, (s0, s1)
        assert np.isfinite(c0).all() and np.isfinite(g0).all()
        assert np.array_equal(c0, c1), (c0, c1)      # same kernels on the same inputs: bit for bit
        assert np.array_equal(g0, g1), float(np.abs(g0 - g1).max())


def test_static_workspace_checks_against_the_oracle_and_refuses_what_does_not_fit():
    import oracle
    from rnnt_amd import _lib

    lib = _lib.load()
    dev = torch.device("cuda:0")
    a, l, al, ll = make_case(4, 37, 9, 4232, seed=5, ragged=True)
    acts = torch.from_numpy(a).to(dev)
    tl, tal, tll = (torch.from_numpy(x).to(dev) for x in (l, al, ll))
    s, c, g = _call(lib, "compute_rnnt_loss_static", acts, tl, tal, tll, _lib.DTYPE_F32, False)
    assert s == 0
    rc, rg, _ = oracle.rnnt_loss_c(a, l, al, ll)
    assert np.all(np.abs(c - rc) <= 1e-5 * np.abs(rc))   # north-star tolerances (fp32)
    assert np.abs(g - rg).max() <= 1e-4

    # a lattice beyond RNNT_STATIC_WORKSPACE_BYTES: its own status, nothing launched, outputs untouched
    B, T, U1 = 64, 800, 61
    assert _lib.workspace_bytes(T, U1, B) > _lib.STATIC_WORKSPACE_BYTES
    V = 8
    big = torch.zeros((B, T, U1, V), dtype=torch.float32, device=dev)
    lab = torch.ones((B, U1 - 1), dtype=torch.int32, device=dev)
    tal = torch.full((B,), T, dtype=torch.int32, device=dev)
    tll = torch.full((B,), U1 - 1, dtype=torch.int32, device=dev)
    s, c, g = _call(lib, "compute_rnnt_loss_static", big, lab, tal, tll, _lib.DTYPE_F32, False)
    assert s == _lib.RNNT_STATUS_STATIC_WORKSPACE_EXCEEDED
    assert b"static workspace" in lib.rnntGetStatusString(s)
    assert np.isnan(c).all() and np.isnan(g).all()
    # ... and the same minibatch is fine through the allocating entry point
    s, c, g = _call(lib, "compute_rnnt_loss", big, lab, tal, tll, _lib.DTYPE_F32, True)
    assert s == 0 and np.isfinite(c).all()


def test_static_entry_point_under_graph_capture_after_a_warm_up_call():
    """The use the header names: one call per device BEFORE capture (the first call looks the device array up, which
    is not capturable), then the call is captured into a HIP graph and replayed on new inputs; the replays allocate
    nothing and equal compute_rnnt_loss bit for bit."""
    from rnnt_amd import _lib

    lib = _lib.load()
    dev = torch.device("cuda:0")
    B, T, U, V = 3, 29, 6, 515
    a, l, al, ll = make_case(B, T, U, V, seed=21, ragged=True)
    acts = torch.from_numpy(a).to(dev)
    tl, tal, tll = (torch.from_numpy(x).to(dev) for x in (l, al, ll))
    grads = torch.empty_like(acts)
    costs = torch.empty(B, dtype=torch.float32, device=dev)

    def enqueue():
        opts = _lib.RnntOptions(torch.cuda.current_stream(dev).cuda_stream, 0, T, U + 1, _lib.DTYPE_F32, None)
        return lib.compute_rnnt_loss_static(acts.data_ptr(), grads.data_ptr(), tl.data_ptr(), tll.data_ptr(),
                                            tal.data_ptr(), V, B, costs.data_ptr(), opts)

    assert enqueue() == 0                      # warm-up: the per-device lookup happens here
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        assert enqueue() == 0
    for seed in (22, 23):
        a2, _, _, _ = make_case(B, T, U, V, seed=seed, ragged=True)
        acts.copy_(torch.from_numpy(a2))
        grads.fill_(float("nan"))
        costs.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize(dev)
        s, c, g = _call(lib, "compute_rnnt_loss", acts, tl, tal, tll, _lib.DTYPE_F32, True)
        assert s == 0 and np.array_equal(costs.cpu().numpy(), c) and np.array_equal(grads.cpu().numpy(), g)
