"""CPU tests (no GPU) for the joint-network widening: the identity the HIP path relies on, and the
C ABI's host-side argument checks."""
import ctypes

import pytest
import torch

from oracle.joint_oracle import joint_logits


@pytest.mark.parametrize("joint", ["concat", "sum"])
def test_split_linear_identity_float64(joint):
    """Linear(cat(e, d)) == W_e e + W_d d + b: the decomposition rnnt_amd.JointNet uses."""
    g = torch.Generator().manual_seed(3)
    B, T, U1, He, Hd, H, V = 2, 5, 4, 6, (6 if joint == "sum" else 3), 8, 7
    enc = torch.randn(B, T, He, generator=g, dtype=torch.float64)
    dec = torch.randn(B, U1, Hd, generator=g, dtype=torch.float64)
    nin = He if joint == "sum" else He + Hd
    w1 = torch.randn(H, nin, generator=g, dtype=torch.float64)
    b1 = torch.randn(H, generator=g, dtype=torch.float64)
    w2 = torch.randn(V, H, generator=g, dtype=torch.float64)
    b2 = torch.randn(V, generator=g, dtype=torch.float64)
    ref, hid = joint_logits(enc, dec, w1, b1, w2, b2, joint, return_hidden=True)
    we, wd = (w1, w1) if joint == "sum" else (w1[:, :He], w1[:, He:])
    ep = torch.nn.functional.linear(enc, we, b1)
    dp = torch.nn.functional.linear(dec, wd)
    mine = torch.tanh(ep[:, :, None, :] + dp[:, None, :, :])
    assert (mine - hid).abs().max() < 1e-13
    assert ref.shape == (B, T, U1, V)


def test_module_mirrors_reference_parameter_names():
    import rnnt_amd

    m = rnnt_amd.JointNet(640, 512, 4232)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {
        "mlp.0.weight": (512, 640), "mlp.0.bias": (512,), "mlp.2.weight": (4232, 512), "mlp.2.bias": (4232,)}
    with pytest.raises(ValueError, match="must be have same size"):
        m(torch.zeros(1, 2, 320), torch.zeros(320))
    with pytest.raises(RuntimeError, match="no CPU"):   # the training path has no CPU implementation
        m(torch.zeros(1, 2, 320), torch.zeros(1, 3, 320))
    # the evaluation path (model.py:51-54, 1-D / 2-D states) is plain torch and runs anywhere
    out = m(torch.zeros(320), torch.zeros(320))
    assert out.shape == (4232,)
    with pytest.raises(AttributeError):                 # model.py:60-61: self.softmax does not exist upstream either
        m(torch.zeros(320), torch.zeros(320), softmax=True)


def test_joint_abi_rejects_bad_arguments_without_a_gpu():
    from rnnt_amd import _lib

    lib = _lib.load()
    n = ctypes.c_size_t(0)
    assert lib.rnnt_joint_hidden_workspace_size(32, 400, 41, 512, ctypes.byref(n)) == 0
    assert n.value == 32 * 25 * 41 * 512 * 4            # one fp32 row per (b, 16-frame chunk, u)
    assert lib.rnnt_joint_hidden_workspace_size(0, 400, 41, 512, ctypes.byref(n)) != 0
    assert lib.rnnt_joint_hidden_forward(None, None, None, 1, 1, 1, 1, 0, None) != 0
    assert lib.rnnt_joint_hidden_backward(None, None, None, None, 1, 1, 1, 1, 0, None, None) != 0


@pytest.mark.parametrize("kind", ["concat", "sum"])
def test_restatement_reproduces_the_reference_joint(kind):
    """oracle.joint_oracle against the REFERENCE'S OWN JointNet (model.py:20-63) run in float64 by
    tests/golden/make_ref_fixtures.py: logits and every gradient, to float64 rounding."""
    from ref_fixtures import joint_case

    c = joint_case(kind)
    enc, dec = c["enc"].clone().requires_grad_(True), c["dec"].clone().requires_grad_(True)
    params = [p.clone().requires_grad_(True) for p in c["params"]]
    logits = joint_logits(enc, dec, *params, joint=kind)
    assert logits.shape == c["logits"].shape
    assert (logits.detach() - c["logits"]).abs().max() <= 1e-13
    (logits * c["go"]).sum().backward()
    assert (enc.grad - c["grad_enc"]).abs().max() <= 1e-12
    assert (dec.grad - c["grad_dec"]).abs().max() <= 1e-12
    for p, g, name in zip(params, c["grads"], c["names"]):
        assert (p.grad - g).abs().max() <= 1e-12, name


def test_restated_greedy_loop_reproduces_the_reference_recognize():
    """oracle.decode_oracle.greedy_decode_reference against the token lists Transducer.recognize
    (model.py:135-172) returned for the fixture's encoder states and parameters; the decoder is rebuilt
    from the fixture's weights with stock torch layers, the joint is oracle.joint_oracle."""
    from oracle.decode_oracle import greedy_decode_reference
    from ref_fixtures import recognize_case

    enc, lengths, tokens, decoder, jp, margin = recognize_case()
    assert margin > 1e-3                  # every argmax in the fixture is far from a tie
    assert [len(t) for t in tokens] == [21, 17, 9]
    for i, want in enumerate(tokens):
        got, m = greedy_decode_reference(
            enc[i], lengths[i], decoder,
            lambda e, d: joint_logits(e.view(1, 1, -1), d.view(1, 1, -1), *jp).view(-1))
        assert got == want, i
        assert m >= margin * 0.99


@pytest.mark.parametrize("vocab,count", [(4232, 2499208), (4234, 2500234)])
def test_parameter_count_matches_the_reference_training_log(vocab, count):
    """The reference's own log records its JointNet's size: "# the number of parameters in the JointNet: 2499208"
    (egs/aishell/exp/4blstm_320henc_1l512dec/train.log:14, V = 4232) and 2500234 (:47, V = 4234), for
    JointNet(input_size=640, inner_dim=512) of config/aishell.yaml:36-39."""
    import rnnt_amd

    assert sum(p.numel() for p in rnnt_amd.JointNet(640, 512, vocab).parameters()) == count
