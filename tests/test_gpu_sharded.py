"""GPU: the sharded harness over RCCL (backend "nccl") with the HIP kernels as loss_fn.

Only one GPU is available to the tests, so the process group has one rank: this covers the RCCL
all-reduce call, the device-side ordered reduction and the autograd wiring on the device; the
multi-rank arithmetic is covered on CPU by tests/test_parallel_gloo.py."""
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from cases import make_case

pytestmark = pytest.mark.gpu


def test_single_rank_rccl_equals_unsharded_bitwise():
    import rnnt_amd

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        acts, labels, al, ll = make_case(5, 20, 7, 256, seed=3, ragged=True, sigma=2.0)
        d = lambda x: torch.from_numpy(x).to(dev)
        for reduction in ("mean", "sum", "none"):
            a1 = d(acts).requires_grad_(True)
            a2 = d(acts).requires_grad_(True)
            l1 = rnnt_amd.RNNTLoss(reduction=reduction)(a1, d(labels), d(al), d(ll))
            l2 = rnnt_amd.ShardedRNNTLoss(reduction=reduction)(a2, d(labels), d(al), d(ll), global_batch=5)
            go = torch.full_like(l1, 0.5)
            l1.backward(go)
            l2.backward(go)
            assert torch.equal(l1, l2), reduction
            assert torch.equal(a1.grad, a2.grad), reduction
        # a shard in the middle of a larger global batch: gradient scaled by 1/B_global
        a3 = d(acts).requires_grad_(True)
        l3 = rnnt_amd.ShardedRNNTLoss(reduction="mean")(a3, d(labels), d(al), d(ll), global_batch=5)
        l3.backward()
        a4 = d(acts).requires_grad_(True)
        rnnt_amd.RNNTLoss(reduction="mean")(a4, d(labels), d(al), d(ll)).backward()
        assert torch.equal(a3.grad, a4.grad)
    finally:
        dist.destroy_process_group()
