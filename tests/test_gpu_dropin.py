"""GPU: behaviour at the reference's call site (src/rnnt/model.py:99-133, bin/train.py:22-51,85-91)
and the execution contexts a drop-in has to survive: other streams, DataParallel-style threads,
HIP graph capture, B=1 retries, U=0, in-place accumulation onto the returned loss."""
import threading

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from cases import assert_grad_parity, make_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def d(x):
    return torch.from_numpy(x).to(DEV)


def oracle_mean(acts, labels, al, ll, go=1.0):
    c, g, _ = oracle.rnnt_loss_c(acts, labels, al, ll)
    return oracle.reduce_like_upstream(c, g, "mean", go)


def test_reference_call_site_sequence():
    """model.py:106-116 + train.py:34-43: int64 collated tensors, .int() casts, warprnnt_pytorch import,
    in-place += of an auxiliary loss, / num_gpu, / accumulation_steps, backward()."""
    from warprnnt_pytorch import RNNTLoss  # exactly the reference's import

    acts, labels, al, ll = make_case(4, 30, 8, 4232, seed=1, ragged=True)
    tokens = d(labels).long()                      # collate gives int64 (src/data/dataset.py:176)
    tokens_length = d(ll).long()
    enc_output_lengths = d(al).long()
    logits = d(acts).requires_grad_(True)
    transducer_loss = RNNTLoss()                   # model.py:87
    tokens_with_bos = F.pad(tokens, pad=[1, 0, 0, 0], value=0)  # model.py:106 (decoder input only)
    assert tokens_with_bos.shape[1] == logits.shape[2]
    output_length = enc_output_lengths.int().cuda()             # model.py:109
    loss = transducer_loss(logits, tokens.int(), output_length, tokens_length.int())  # model.py:116
    assert loss.shape == (1,) and loss.dtype == torch.float32
    loss += torch.tensor(0.25, device=DEV) * 0.5   # model.py:124: in-place add of an aux loss
    loss = loss / 1                                # train.py:35  loss / num_gpu
    assert not (torch.isnan(loss) or torch.isinf(loss))         # train.py:36-41
    loss = loss / 2                                # train.py:42  / accumulation_steps
    loss.backward()                                # train.py:43
    ref_loss, ref_g = oracle_mean(acts, labels, al, ll, go=0.5)
    assert abs(loss.item() - (ref_loss[0] + 0.125) / 2) <= 1e-5 * abs(ref_loss[0])
    assert np.abs(logits.grad.cpu().numpy() - ref_g).max() <= 1e-4


def test_batch_of_one_and_no_labels():
    """train.py:85-91 retries a failed batch one utterance at a time; U=0 is a legal lattice."""
    import rnnt_amd

    acts, labels, al, ll = make_case(1, 13, 5, 300, seed=2, ragged=False)
    a = d(acts).requires_grad_(True)
    loss = rnnt_amd.RNNTLoss()(a, d(labels), d(al), d(ll))
    loss.backward()
    ref_loss, ref_g = oracle_mean(acts, labels, al, ll)
    assert abs(loss.item() - ref_loss[0]) <= 1e-5 * abs(ref_loss[0])
    assert np.abs(a.grad.cpu().numpy() - ref_g).max() <= 1e-4

    acts0, labels0, al0, ll0 = make_case(3, 6, 0, 50, seed=3, ragged=True)
    assert labels0.shape == (3, 0)
    a0 = d(acts0).requires_grad_(True)
    loss0 = rnnt_amd.RNNTLoss()(a0, torch.zeros((3, 0), dtype=torch.int32, device=DEV), d(al0), d(ll0))
    loss0.backward()
    ref_loss0, ref_g0 = oracle_mean(acts0, labels0, al0, ll0)
    assert abs(loss0.item() - ref_loss0[0]) <= 1e-5 * abs(ref_loss0[0])
    assert np.abs(a0.grad.cpu().numpy() - ref_g0).max() <= 1e-4


@pytest.mark.parametrize("U", [64, 65, 300, 511])
def test_wide_lattices_multi_wave_k2(U):
    """maxU = U+1 up to 512 columns: 2..8 waves per direction in K2 with the LDS edge hand-off."""
    import rnnt_amd

    acts, labels, al, ll = make_case(2, 7, U, 16, seed=U, ragged=True, min_U=U // 2)
    a = d(acts).requires_grad_(True)
    loss = rnnt_amd.RNNTLoss()(a, d(labels), d(al), d(ll))
    loss.backward()
    ref_loss, ref_g = oracle_mean(acts, labels, al, ll)
    assert abs(loss.item() - ref_loss[0]) <= 1e-5 * abs(ref_loss[0])
    _, g64, _ = oracle.rnnt_loss_c(acts, labels, al, ll, dtype=np.float64)
    assert_grad_parity(a.grad.cpu().numpy(), ref_g, (g64 / 2).astype(np.float64), reduction="mean",
                       label=f"wide lattice U={U}")


@pytest.mark.parametrize("U,T,ragged", [(512, 5, True), (600, 9, True), (1030, 3, False), (2500, 4, True), (4095, 2, False)])
def test_lattices_beyond_512_columns_take_the_wide_k2(U, T, ragged):
    """More than 511 labels (upstream has no limit): rnnt_k2_lattice_wide, 1024 threads per utterance, several
    columns per thread, previous anti-diagonal in LDS.  Same bar as every other lattice."""
    import rnnt_amd

    acts, labels, al, ll = make_case(2, T, U, 8, seed=U, ragged=ragged, min_U=U // 2)
    a = d(acts).requires_grad_(True)
    costs = rnnt_amd.RNNTLoss(reduction="none")(a, d(labels), d(al), d(ll))
    costs.backward(torch.ones_like(costs))
    c, g, _ = oracle.rnnt_loss_c(acts, labels, al, ll)
    np.testing.assert_allclose(costs.detach().cpu().numpy(), c, rtol=1e-5)
    _, g64, _ = oracle.rnnt_loss_c(acts, labels, al, ll, dtype=np.float64)
    got = a.grad.cpu().numpy()
    e_gpu, e_cpu = float(np.abs(got - g64).max()), float(np.abs(g - g64).max())
    if e_cpu <= 2e-5:
        assert np.abs(got - g).max() <= 1e-4
    else:
        # |ll| is in the thousands here (V = 8, thousands of labels): one fp32 ulp of alpha + beta - ll is ~1e-3,
        # and the fp32 ORACLE is that far from float64 (1.3e-2 at U = 4095).  The kernel may be no further.
        assert e_gpu <= 1.5 * e_cpu + 1e-4, (e_gpu, e_cpu)


def test_wide_and_fast_k2_give_the_same_bits():
    """A 300-label lattice through both lattice kernels: the tensor is padded to 600 columns for the wide one."""
    import rnnt_amd

    acts, labels, al, ll = make_case(3, 11, 300, 8, seed=4, ragged=True, min_U=100)
    crit = rnnt_amd.RNNTLoss(reduction="none")
    a = d(acts).requires_grad_(True)
    c0 = crit(a, d(labels), d(al), d(ll))
    c0.sum().backward()
    wide = np.full((3, 11, 601, 8), np.nan, np.float32)      # padding content must not matter
    wide[:, :, :301] = acts
    lab = np.zeros((3, 600), np.int32)
    lab[:, :300] = labels
    a1 = d(wide).requires_grad_(True)
    c1 = crit(a1, d(lab), d(al), d(ll))
    c1.sum().backward()
    assert torch.equal(c0, c1)
    assert torch.equal(a.grad, a1.grad[:, :, :301]) and not a1.grad[:, :, 301:].any()


def test_too_wide_lattice_is_an_error_not_a_fault():
    import rnnt_amd

    a = torch.zeros((1, 2, 4098, 4), device=DEV)
    lab = torch.ones((1, 4097), dtype=torch.int32, device=DEV)
    al, ll = (torch.tensor([v], dtype=torch.int32, device=DEV) for v in (2, 4097))
    # the operator: ValueError before any launch (a RuntimeError would send the reference's loop into its
    # out-of-memory retry, bin/train.py:80-91)
    with pytest.raises(ValueError, match="at most 4095 labels"):
        rnnt_amd.RNNTLoss()(a, lab, al, ll)
    # the C ABI underneath: a status
    from rnnt_amd import _lib
    import ctypes

    n = ctypes.c_size_t(0)
    assert _lib.load().rnnt_get_workspace_size(2, 4098, 1, ctypes.byref(n)) == 2
    costs = torch.empty(1, device=DEV)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    opts = _lib.RnntOptions(torch.cuda.current_stream().cuda_stream, 0, 2, 4098, _lib.DTYPE_F32, None)
    assert _lib.load().rnnt_forward(a.data_ptr(), lab.data_ptr(), al.data_ptr(), ll.data_ptr(), 4, 1,
                                    costs.data_ptr(), None, 0, ws.data_ptr(), opts) == 2


def test_forward_only_and_double_backward():
    import rnnt_amd

    acts, labels, al, ll = make_case(2, 9, 4, 64, seed=5, ragged=True)
    with torch.no_grad():
        loss = rnnt_amd.RNNTLoss()(d(acts), d(labels), d(al), d(ll))
    assert not loss.requires_grad
    ref_loss, _ = oracle_mean(acts, labels, al, ll)
    assert abs(loss.item() - ref_loss[0]) <= 1e-5 * abs(ref_loss[0])
    a = d(acts).requires_grad_(True)
    loss = rnnt_amd.RNNTLoss()(a, d(labels), d(al), d(ll))
    (g,) = torch.autograd.grad(loss, a, create_graph=True)
    with pytest.raises(RuntimeError):
        g.sum().backward()  # once_differentiable: no silent wrong second derivative
    # gradient accumulates into an existing .grad like any autograd op
    a2 = d(acts).requires_grad_(True)
    for _ in range(2):
        rnnt_amd.RNNTLoss()(a2, d(labels), d(al), d(ll)).backward()
    assert torch.allclose(a2.grad, 2 * g.detach(), rtol=0, atol=1e-7)


def test_non_default_stream_and_threads():
    """DataParallel (bin/train.py:244) calls the loss from one thread per replica, each on its
    own stream; the library is re-entrant per (stream, workspace)."""
    import rnnt_amd

    cases = [make_case(3, 40, 9, 1000, seed=10 + i, ragged=True) for i in range(4)]
    refs = [oracle_mean(*c) for c in cases]
    out = [None] * 4

    def work(i):
        s = torch.cuda.Stream(device=DEV)
        with torch.cuda.stream(s):
            acts, labels, al, ll = cases[i]
            for _ in range(5):
                a = d(acts).requires_grad_(True)
                loss = rnnt_amd.RNNTLoss()(a, d(labels), d(al), d(ll))
                loss.backward()
            s.synchronize()
            out[i] = (loss.item(), a.grad.cpu().numpy())

    ths = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    for i in range(4):
        assert abs(out[i][0] - refs[i][0][0]) <= 1e-5 * abs(refs[i][0][0])
        assert np.abs(out[i][1] - refs[i][1]).max() <= 1e-4


def test_hip_graph_capture_of_forward_and_backward():
    """No allocation, no host sync inside the C ABI: fwd+bwd can be captured and replayed."""
    import rnnt_amd

    acts, labels, al, ll = make_case(4, 25, 6, 512, seed=20, ragged=True)
    crit = rnnt_amd.RNNTLoss(check_lengths=False)  # the host check is the only synchronising part
    static_a = d(acts).requires_grad_(True)
    lab, ald, lld = d(labels), d(al), d(ll)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):  # warm-up outside capture (library load, allocator pools)
        for _ in range(2):
            static_a.grad = None
            crit(static_a, lab, ald, lld).backward()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    static_a.grad = None
    with torch.cuda.graph(graph):
        static_loss = crit(static_a, lab, ald, lld)
        static_loss.backward()
    acts2, _, _, _ = make_case(4, 25, 6, 512, seed=21, ragged=True)
    for src in (acts, acts2):
        with torch.no_grad():
            static_a.copy_(d(src))
        graph.replay()
        torch.cuda.synchronize()
        ref_loss, ref_g = oracle_mean(src, labels, al, ll)
        assert abs(static_loss.item() - ref_loss[0]) <= 1e-5 * abs(ref_loss[0])
        assert np.abs(static_a.grad.cpu().numpy() - ref_g).max() <= 1e-4
