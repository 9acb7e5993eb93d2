"""The batch-sharded harness (rnnt_amd/parallel.py) on CPU: world_size 2 and 3 over gloo.

The HIP kernels cannot run here, so the per-utterance cost function is injected: an autograd
wrapper around the CPU oracle.  What is under test is the harness: shard bounds, the single
all-reduce of per-utterance costs, the fixed-order scalar, and that each rank's gradient equals
its slice of the unsharded gradient bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from rnnt_amd.parallel import (ShardedRNNTLoss, balanced_assignment, ordered_reduce, shard_batch,  # noqa: E402
                               shard_bounds)


class OracleCosts(torch.autograd.Function):
    """[B] per-utterance costs from the CPU oracle, with its gradient."""

    @staticmethod
    def forward(ctx, acts, labels, act_lens, label_lens, blank):
        import oracle

        c, g, _ = oracle.rnnt_loss_c(acts.detach().numpy(), labels.numpy(), act_lens.numpy(),
                                     label_lens.numpy(), blank=blank, num_threads=1)
        ctx.save_for_backward(torch.from_numpy(g))
        return torch.from_numpy(c)

    @staticmethod
    def backward(ctx, go):
        (g,) = ctx.saved_tensors
        return g * go.reshape(-1, 1, 1, 1), None, None, None, None


def oracle_loss_fn(acts, labels, act_lens, label_lens, blank):
    return OracleCosts.apply(acts, labels, act_lens, label_lens, blank)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def worker(rank, world, port, B, reduction, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cases import make_case

        acts, labels, al, ll = make_case(B, 9, 4, 12, seed=77, ragged=True, sigma=2.0)
        full = [torch.from_numpy(x) for x in (acts, labels, al, ll)]
        a, lab, alr, llr = shard_batch(*full, world, rank)
        a = a.clone().requires_grad_(True)
        crit = ShardedRNNTLoss(blank=0, reduction=reduction, loss_fn=oracle_loss_fn)
        loss = crit(a, lab, alr, llr, global_batch=B)
        if reduction == "none":
            go = torch.linspace(0.5, 1.5, B)
            loss.backward(go)
        else:
            loss.backward(torch.tensor([0.5]))
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), loss=loss.detach().numpy(), grad=a.grad.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,B,reduction", [(2, 4, "mean"), (2, 5, "sum"), (3, 7, "mean"), (2, 4, "none")])
def test_sharded_equals_unsharded_bitwise(tmp_path, world, B, reduction):
    import oracle
    from cases import make_case

    port = free_port()
    mp.spawn(worker, args=(world, port, B, reduction, str(tmp_path)), nprocs=world, join=True)

    acts, labels, al, ll = make_case(B, 9, 4, 12, seed=77, ragged=True, sigma=2.0)
    c, g, _ = oracle.rnnt_loss_c(acts, labels, al, ll, num_threads=1)
    if reduction == "none":
        ref_loss = c
        ref_g = g * torch.linspace(0.5, 1.5, B).numpy().reshape(B, 1, 1, 1)
    else:
        ref_loss = ordered_reduce(torch.from_numpy(c), reduction).numpy()
        factor = np.float32(1.0 / B) if reduction == "mean" else np.float32(1.0)
        ref_g = g * (np.float32(0.5) * factor)
    for r in range(world):
        z = np.load(tmp_path / f"r{r}.npz")
        lo, hi = shard_bounds(B, world, r)
        np.testing.assert_array_equal(z["loss"], ref_loss)       # identical bits on every rank
        np.testing.assert_array_equal(z["grad"], ref_g[lo:hi])       # and identical gradient bits
        assert np.all((z["grad"] == 0) == (ref_g[lo:hi] == 0))   # padding stays exactly zero


def balanced_worker(rank, world, port, B, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cases import make_case

        acts, labels, al, ll = make_case(B, 9, 4, 12, seed=78, ragged=True, sigma=2.0)
        mine = balanced_assignment(al, ll, world)[rank]
        a = torch.from_numpy(acts[mine]).requires_grad_(True)
        crit = ShardedRNNTLoss(blank=0, reduction="mean", loss_fn=oracle_loss_fn)
        loss = crit(a, torch.from_numpy(labels[mine]), torch.from_numpy(al[mine]), torch.from_numpy(ll[mine]),
                    global_batch=B, index=mine)
        loss.backward(torch.tensor([0.5]))
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), loss=loss.detach().numpy(), grad=a.grad.numpy(),
                 index=np.asarray(mine))
    finally:
        dist.destroy_process_group()


def test_balanced_partition_equals_unsharded_bitwise(tmp_path):
    import oracle
    from cases import make_case

    world, B = 2, 7
    mp.spawn(balanced_worker, args=(world, free_port(), B, str(tmp_path)), nprocs=world, join=True)
    acts, labels, al, ll = make_case(B, 9, 4, 12, seed=78, ragged=True, sigma=2.0)
    c, g, _ = oracle.rnnt_loss_c(acts, labels, al, ll, num_threads=1)
    ref_loss = ordered_reduce(torch.from_numpy(c), "mean").numpy()
    ref_g = g * (np.float32(0.5) * np.float32(1.0 / B))
    seen = []
    for r in range(world):
        z = np.load(tmp_path / f"r{r}.npz")
        np.testing.assert_array_equal(z["loss"], ref_loss)
        np.testing.assert_array_equal(z["grad"], ref_g[z["index"]])
        seen += z["index"].tolist()
    assert sorted(seen) == list(range(B))


class TinyTransducer(torch.nn.Module):
    """enc/dec features -> logits [B,T,U+1,V] the way the reference's joint does (model.py:39-59)."""

    def __init__(self, D=6, V=7, mixed=False):
        super().__init__()
        torch.manual_seed(3)
        self.enc = torch.nn.Linear(D, 8)
        if mixed:   # parameters of two dtypes: GuardedStep all-reduces per dtype, in the same order on every rank
            self.enc = self.enc.double()
        self.dec = torch.nn.Embedding(V, 8)
        self.out = torch.nn.Linear(8, V)
        self.unused = torch.nn.Parameter(torch.zeros(2))  # never reached by the loss: .grad stays None

    def forward(self, feats, tokens_with_bos):
        e = self.enc(feats.to(self.enc.weight.dtype)).float()
        h = torch.tanh(e.unsqueeze(2) + self.dec(tokens_with_bos).unsqueeze(1))
        return self.out(h)


def _training_batches(B, steps):
    from cases import make_case

    out = []
    for k in range(steps):
        _, labels, al, ll = make_case(B, 6, 3, 7, seed=500 + k, ragged=True)
        feats = np.random.default_rng(600 + k).standard_normal((B, 6, 6)).astype(np.float32)
        if k == 1:
            feats[0, 0, 0] = np.nan  # a NaN loss: the reference skips this update (bin/train.py:36-41)
        out.append((feats, labels, al, ll))
    return out


def _train(model, batches, world, rank, mode):
    """Three optimiser steps; `mode`: 'single' (unsharded), 'helper' (loss.backward + all_reduce_gradients)
    or 'guarded' (GuardedStep(group=True))."""
    from rnnt_amd import GuardedStep, all_reduce_gradients

    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    guard = GuardedStep(model.parameters(), accumulation_steps=1, max_grad_norm=5.0,
                        group=True if mode.startswith("guarded") else None)
    for feats, labels, al, ll in batches:
        B = feats.shape[0]
        full = [torch.from_numpy(x) for x in (feats, labels, al, ll)]
        lo, hi = shard_bounds(B, world, rank) if world > 1 else (0, B)
        f, lab, alr, llr = (x[lo:hi] for x in full)
        logits = model(f, torch.nn.functional.pad(lab.long(), [1, 0, 0, 0], value=0))
        if world > 1:
            loss = ShardedRNNTLoss(loss_fn=oracle_loss_fn)(logits, lab, alr, llr, global_batch=B)
        else:
            c = oracle_loss_fn(logits, lab, alr, llr, 0)
            loss = ordered_reduce(c, "mean") + (c.sum() - c.sum().detach()) / B
        if mode == "helper":
            if bool(torch.isfinite(loss).all()):
                loss.backward()
                all_reduce_gradients(model.parameters())
                torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        else:
            guard(loss)
        opt.step()
        opt.zero_grad()
    return {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}


def training_worker(rank, world, port, mode, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params = _train(TinyTransducer(mixed=mode.endswith("mixed")), _training_batches(5, 3), world, rank, mode)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), **params)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["helper", "guarded", "guarded-mixed"])
def test_two_ranks_train_like_one(tmp_path, mode):
    """sharded loss + SUM all-reduce of the parameter gradients reproduces the
    unsharded training run (what DataParallel's reduce-add gave the reference), including a skipped
    NaN batch and gradient clipping; both ranks end with the same parameters."""
    world = 2
    mp.spawn(training_worker, args=(world, free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    mixed = mode.endswith("mixed")
    ref = _train(TinyTransducer(mixed=mixed), _training_batches(5, 3), 1, 0, "single")
    start = {k: v.detach().numpy() for k, v in TinyTransducer(mixed=mixed).state_dict().items()}
    assert any(np.abs(ref[k] - start[k]).max() > 1e-3 for k in ref)      # training moved the parameters
    r0, r1 = (np.load(tmp_path / f"r{r}.npz") for r in range(world))
    for k in ref:
        np.testing.assert_array_equal(r0[k], r1[k])                       # replicas stay in step
        np.testing.assert_allclose(r0[k], ref[k], rtol=0, atol=2e-6)      # summation order differs


def test_balanced_assignment_balances_lattice_work():
    rng = np.random.default_rng(5)
    al = rng.integers(100, 801, size=64)
    ll = rng.integers(10, 61, size=64)
    parts = balanced_assignment(al, ll, 8)
    assert sorted(i for p in parts for i in p) == list(range(64))
    assert all(len(p) == 8 for p in parts)
    work = np.array([sum(int(al[i]) * (int(ll[i]) + 1) for i in p) for p in parts], float)
    contiguous = np.array([sum(int(al[i]) * (int(ll[i]) + 1) for i in range(8 * r, 8 * r + 8)) for r in range(8)], float)
    assert work.max() / work.mean() < 1.05 < contiguous.max() / contiguous.mean()
    assert [len(p) for p in balanced_assignment([5, 4, 3, 2, 1], [0] * 5, 2)] == [3, 2]
    with pytest.raises(ValueError):
        balanced_assignment([1], [1], 2)


def test_shard_bounds():
    assert [shard_bounds(7, 3, r) for r in range(3)] == [(0, 3), (3, 5), (5, 7)]
    assert [shard_bounds(256, 8, r) for r in range(8)] == [(32 * r, 32 * r + 32) for r in range(8)]
    with pytest.raises(ValueError):
        shard_bounds(2, 3, 0)
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def test_ordered_reduce_is_sequential_fp32():
    c = torch.tensor([1e8, 1.0, -1e8, 1.0], dtype=torch.float32)
    # ((1e8 + 1) - 1e8) + 1 in fp32 = 1, not 2: proves index order, fp32 accumulation
    assert float(ordered_reduce(c, "sum")[0]) == 1.0
    assert float(ordered_reduce(c, "mean")[0]) == 0.25


def test_needs_process_group():
    crit = ShardedRNNTLoss(loss_fn=oracle_loss_fn)
    with pytest.raises(RuntimeError, match="process group"):
        crit(torch.zeros(1, 1, 1, 2), torch.zeros(1, 0, dtype=torch.int32), torch.ones(1, dtype=torch.int32),
             torch.zeros(1, dtype=torch.int32))
