"""GPU: the loss and its consumer (rnnt_amd.GuardedStep) run a training micro-step without a single
host synchronisation, and end where the reference's loop (bin/train.py:34-51,93-96) ends — including
a micro-batch whose loss is NaN."""
import copy

import pytest
import torch
from torch import nn

import rnnt_amd
from rnnt_amd.train_step import GuardedStep
from oracle.train_step_oracle import consume_loss_reference

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class TinyTransducer(nn.Module):
    """enc/dec projections + tanh + Linear(H -> V): the joint of src/rnnt/model.py:9-63 in small."""

    def __init__(self, feat, hid, vocab):
        super().__init__()
        self.enc = nn.Linear(feat, hid)
        self.emb = nn.Embedding(vocab, hid)
        self.out = nn.Linear(hid, vocab)

    def forward(self, x, tokens_with_bos):
        h = torch.tanh(self.enc(x)[:, :, None, :] + self.emb(tokens_with_bos)[:, None, :, :])
        return self.out(h)  # [B, T, U+1, V]


def micro_batches(n, B, T, U, feat, vocab, seed):
    g = torch.Generator().manual_seed(seed)
    for i in range(n):
        x = torch.randn(B, T, feat, generator=g)
        if i == 2:
            x[1, 3, 0] = float("nan")  # one frame of one utterance poisons the batch mean
        labels = torch.randint(1, vocab, (B, U), generator=g, dtype=torch.int32)
        al = torch.randint(T // 2, T + 1, (B,), generator=g, dtype=torch.int32)
        ll = torch.randint(1, U + 1, (B,), generator=g, dtype=torch.int32)
        al[0], ll[0] = T, U
        yield x.to(DEV), labels.to(DEV), al.to(DEV), ll.to(DEV)


def test_micro_steps_without_host_sync_match_reference_loop():
    B, T, U, feat, vocab, accum, max_norm = 3, 12, 5, 7, 33, 2, 0.5
    torch.manual_seed(0)
    model = TinyTransducer(feat, 16, vocab).to(DEV)
    ref_model = copy.deepcopy(model)
    crit = rnnt_amd.RNNTLoss(check_lengths=False)  # the length check is the loss's only host read
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    opt_ref = torch.optim.SGD(ref_model.parameters(), lr=0.1)
    guard = GuardedStep(model.parameters(), accumulation_steps=accum, max_grad_norm=max_norm)
    data = list(micro_batches(6, B, T, U, feat, vocab, seed=5))
    bos = [torch.nn.functional.pad(lab.long(), [1, 0], value=0) for _, lab, _, _ in data]
    torch.cuda.synchronize()

    returned = []
    torch.cuda.set_sync_debug_mode("error")  # any host synchronisation raises from here on
    try:
        for step, ((x, labels, al, ll), tok) in enumerate(zip(data, bos)):
            loss = crit(model(x, tok), labels, al, ll)
            returned.append(guard(loss))
            if (step + 1) % accum == 0:
                opt.step()
                opt.zero_grad()
    finally:
        torch.cuda.set_sync_debug_mode("default")

    total_ref = 0.0
    for step, ((x, labels, al, ll), tok) in enumerate(zip(data, bos)):
        loss = crit(ref_model(x, tok), labels, al, ll)
        lv, gn = consume_loss_reference(loss, ref_model.parameters(), 1, accum, max_norm)
        total_ref += lv
        assert float(returned[step][0]) == pytest.approx(float(lv), rel=1e-5, abs=1e-7)
        assert float(returned[step][1]) == pytest.approx(float(gn), rel=1e-5, abs=1e-7)
        if (step + 1) % accum == 0:
            opt_ref.step()
            opt_ref.zero_grad()
    assert int(guard.skipped) == 1
    assert float(returned[2][0]) == 0.0 and float(returned[2][1]) == 0.0
    assert float(guard.total_loss) == pytest.approx(total_ref, rel=1e-5)
    for p, p_ref in zip(model.parameters(), ref_model.parameters()):
        assert torch.isfinite(p).all()
        torch.testing.assert_close(p, p_ref, rtol=1e-5, atol=1e-7)
