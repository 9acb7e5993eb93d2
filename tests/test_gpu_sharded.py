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
            go = torch.full_like(l1, 0.5)
            l1.backward(go)
            args = (a2, d(labels), d(al), d(ll))
            # the sharded default waits for nothing on the host: no length check (check_lengths=False), the
            # all-reduce asynchronous behind K2, the ordered reduction on the device
            torch.cuda.set_sync_debug_mode("error")
            try:
                l2 = rnnt_amd.ShardedRNNTLoss(reduction=reduction)(*args, global_batch=5)
                l2.backward(go)
                if reduction == "mean":  # a non-contiguous partition, too (its index table goes up without a wait)
                    l5 = rnnt_amd.ShardedRNNTLoss(reduction=reduction)(
                        args[0].detach(), *args[1:], global_batch=5, index=[0, 1, 2, 3, 4])
            finally:
                torch.cuda.set_sync_debug_mode("default")
            if reduction == "mean":
                assert torch.equal(l5, l2.detach())
            assert torch.equal(l1, l2), reduction
            assert torch.equal(a1.grad, a2.grad), reduction
        # check_lengths=True brings the host-side check back (one host wait): a length beyond T raises
        with pytest.raises(ValueError, match="Input length mismatch"):
            rnnt_amd.ShardedRNNTLoss(check_lengths=True)(d(acts), d(labels), d(al) + 1, d(ll), global_batch=5)
        # a shard in the middle of a larger global batch: gradient scaled by 1/B_global
        a3 = d(acts).requires_grad_(True)
        l3 = rnnt_amd.ShardedRNNTLoss(reduction="mean")(a3, d(labels), d(al), d(ll), global_batch=5)
        l3.backward()
        a4 = d(acts).requires_grad_(True)
        rnnt_amd.RNNTLoss(reduction="mean")(a4, d(labels), d(al), d(ll)).backward()
        assert torch.equal(a3.grad, a4.grad)
    finally:
        dist.destroy_process_group()
