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


class TinyReferenceModel(nn.Module):
    """The tail of the reference's ``Transducer.forward`` (src/rnnt/model.py:100-117) in small: an encoder and a
    decoder stand-in (one Linear, one Embedding fed the BOS-padded tokens of :113), the joint network, the loss."""

    def __init__(self, feat, he, hd, inner, vocab):
        super().__init__()
        self.enc = nn.Linear(feat, he)
        self.emb = nn.Embedding(vocab, hd)
        self.joint = rnnt_amd.JointNet(he + hd, inner, vocab)

    def states(self, x, labels):
        tok = torch.nn.functional.pad(labels.long(), [1, 0], value=0)
        return self.enc(x), self.emb(tok)

    def loss_one_call(self, x, labels, al_host, ll_host):
        enc, dec = self.states(x, labels)
        return self.joint.transducer_loss(enc, dec, labels, al_host, ll_host)   # lengths where the reference has them

    def loss_reference_formulation(self, x, labels, al, ll, crit):
        from oracle.joint_oracle import joint_logits                           # model.py:39-59 in stock torch ops
        enc, dec = self.states(x, labels)
        first, out = self.joint.mlp[0], self.joint.mlp[2]
        logits = joint_logits(enc, dec, first.weight, first.bias, out.weight, out.bias)
        return crit(logits, labels, al, ll)                                     # model.py:116


@pytest.mark.parametrize("dtype,inner,tol", [(torch.float32, 96, 1e-4), (torch.bfloat16, 512, 3e-2)])
def test_ragged_micro_steps_through_the_one_call_joint_and_loss(dtype, inner, tol):
    """Rows (f1)+(f2)+(f3) of SURVEY §8 in one loop: `JointNet.transducer_loss` with the lengths on the host (ragged
    batch: compacted node order — fp32 through the gathered library GEMMs and `rnnt_loss_compact`, bf16 through the
    fused matrix-core kernels) feeding `GuardedStep`, without a host synchronisation, against the reference's
    formulation (repeat, cat, Linear, Tanh, Linear; the loss on the padded tensor; bin/train.py:34-51) on a copy of the
    model.  Same per-step losses and gradient norms, same parameters after six micro-steps (fp32), within the
    summation-order differences of GEMMs over different row sets (16-bit: of one rounding of every logit)."""
    B, T, U, feat, he, hd, vocab, accum, max_norm = 6, 120, 30, 20, 24, 16, 300, 2, 5.0
    torch.manual_seed(3)
    model = TinyReferenceModel(feat, he, hd, inner, vocab).to(DEV).to(dtype)
    ref_model = copy.deepcopy(model)
    crit = rnnt_amd.RNNTLoss(check_lengths=False)
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    opt_ref = torch.optim.SGD(ref_model.parameters(), lr=0.05)
    guard = GuardedStep(model.parameters(), accumulation_steps=accum, max_grad_norm=max_norm)
    g = torch.Generator().manual_seed(11)
    data = []
    for i in range(6):
        x = torch.randn(B, T, feat, generator=g)
        if i == 3:
            x[2, 5, 1] = float("nan")                                           # inside utterance 2: the batch is skipped
        labels = torch.randint(1, vocab, (B, U), generator=g, dtype=torch.int32)
        al = torch.randint(T // 4, T + 1, (B,), generator=g, dtype=torch.int32)
        ll = torch.randint(1, U + 1, (B,), generator=g, dtype=torch.int32)
        al[i % B], ll[i % B] = T, U
        data.append((x.to(DEV).to(dtype), labels.to(DEV), al, ll, al.to(DEV), ll.to(DEV)))
    # warm the libraries up outside the no-synchronisation region (first GEMM calls load code objects)
    model.loss_one_call(*data[0][:4]).backward()
    model.zero_grad(set_to_none=True)
    torch.cuda.synchronize()

    returned = []
    torch.cuda.set_sync_debug_mode("error")
    try:
        for step, (x, labels, al, ll, _, _) in enumerate(data):
            returned.append(guard(model.loss_one_call(x, labels, al, ll)))
            if (step + 1) % accum == 0:
                opt.step()
                opt.zero_grad()
    finally:
        torch.cuda.set_sync_debug_mode("default")

    for step, (x, labels, _, _, ald, lld) in enumerate(data):
        loss = ref_model.loss_reference_formulation(x, labels, ald, lld, crit)
        lv, gn = consume_loss_reference(loss, ref_model.parameters(), 1, accum, max_norm)
        assert float(returned[step][0]) == pytest.approx(float(lv), rel=tol, abs=1e-7), step
        assert float(returned[step][1]) == pytest.approx(float(gn), rel=10 * tol, abs=1e-7), step
        if (step + 1) % accum == 0:
            opt_ref.step()
            opt_ref.zero_grad()
    assert int(guard.skipped) == 1 and float(returned[3][0]) == 0.0
    for p, p_ref in zip(model.parameters(), ref_model.parameters()):
        assert torch.isfinite(p).all()
        if dtype is torch.float32:
            torch.testing.assert_close(p, p_ref, rtol=1e-4, atol=1e-5)
        else:   # bf16 parameters: a step of lr * gradient below half an ulp is lost on either side; hold the bulk
            a, b = p.detach().float(), p_ref.detach().float()
            assert float((a - b).abs().max()) <= 2.0 ** -6 * max(1.0, float(b.abs().max()))
