"""GuardedStep against the reference's consumer of the loss restated with its host-side decisions
(oracle/train_step_oracle.py, bin/train.py:34-51).  Same model, same micro-batches, two of them NaN
and one infinite."""
import copy

import pytest
import torch
from torch import nn

from oracle.train_step_oracle import consume_loss_reference
from rnnt_amd.train_step import GuardedStep


def tiny_model(seed):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Linear(6, 8), nn.Tanh(), nn.Linear(8, 3))


def batches(n, seed, poison):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        x = torch.randn(5, 6, generator=g) * (3.0 if i % 2 else 0.5)
        out.append((x, poison.get(i)))
    return out


def loss_of(model, x, poison):
    y = model(x).pow(2).sum().reshape(1)  # shape [1] like RNNTLoss(reduction='mean')
    if poison == "nan":
        y = y * float("nan")
    elif poison == "inf":
        y = y + float("inf")
    return y


@pytest.mark.parametrize("accum,max_norm,num_gpu", [(1, None, 1), (2, 1.0, 1), (3, 200.0, 2), (2, 0.05, 4)])
def test_matches_reference_consumer(accum, max_norm, num_gpu):
    ref_model, model = tiny_model(3), tiny_model(3)
    opt_ref = torch.optim.SGD(ref_model.parameters(), lr=0.05)
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    guard = GuardedStep(model.parameters(), accumulation_steps=accum, max_grad_norm=max_norm, num_gpu=num_gpu)
    total_ref = 0.0
    opt_ref.zero_grad()
    opt.zero_grad()
    for step, (x, poison) in enumerate(batches(9, 11, {2: "nan", 5: "inf", 6: "nan"})):
        lv_ref, gn_ref = consume_loss_reference(loss_of(ref_model, x, poison), ref_model.parameters(), num_gpu, accum, max_norm)
        total_ref += lv_ref
        lv, gn = guard(loss_of(model, x, poison))
        assert lv.shape == () and gn.shape == ()
        assert float(lv) == pytest.approx(float(lv_ref), rel=1e-6, abs=0)
        assert float(gn) == pytest.approx(float(gn_ref), rel=1e-6, abs=0)
        for p_ref, p in zip(ref_model.parameters(), model.parameters()):
            if p_ref.grad is None:
                assert p.grad is None or not p.grad.any()
            else:
                torch.testing.assert_close(p.grad, p_ref.grad, rtol=1e-6, atol=0)
                assert torch.isfinite(p.grad).all()
        if (step + 1) % accum == 0:  # bin/train.py:93-96
            opt_ref.step()
            opt_ref.zero_grad()
            opt.step()
            opt.zero_grad()
    assert int(guard.skipped) == 3
    assert float(guard.total_loss) == pytest.approx(total_ref, rel=1e-5)
    assert float(guard.average_loss()) == pytest.approx(total_ref / 9, rel=1e-5)
    for p_ref, p in zip(ref_model.parameters(), model.parameters()):
        torch.testing.assert_close(p, p_ref, rtol=1e-6, atol=0)


def test_first_batch_skipped_leaves_zero_grads():
    model = tiny_model(1)
    guard = GuardedStep(model.parameters(), max_grad_norm=1.0)
    lv, gn = guard(loss_of(model, torch.randn(2, 6), "nan"))
    assert float(lv) == 0 and float(gn) == 0
    assert all(p.grad is not None and not p.grad.any() for p in model.parameters())


def test_unused_parameter_and_argument_checks():
    model = tiny_model(2)
    extra = nn.Parameter(torch.ones(3))
    guard = GuardedStep(list(model.parameters()) + [extra])
    guard(loss_of(model, torch.randn(2, 6), None))
    assert extra.grad is None
    with pytest.raises(ValueError):
        GuardedStep([torch.zeros(3)])
    with pytest.raises(ValueError):
        GuardedStep(model.parameters(), accumulation_steps=0)
