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
        # the fused projection + loss under the same harness: loss_fn receives the shard's hidden activations
        torch.manual_seed(2)
        H, V = 512, 520
        proj = torch.nn.Linear(H, V).to(dev).to(torch.bfloat16)
        hidden = torch.tanh(torch.randn(5, 20, 8, H, device=dev)).to(torch.bfloat16)
        fused_costs = lambda h, lab, tl, ul, blank: rnnt_amd.fused_joint_loss(
            h, proj.weight, proj.bias, lab, tl, ul, blank=blank, reduction="none", min_nodes=0)
        h1, h2 = hidden.clone().requires_grad_(True), hidden.clone().requires_grad_(True)
        l1 = rnnt_amd.FusedJointLoss(min_nodes=0)(h1, proj, d(labels), d(al), d(ll))
        l1.backward()
        gw = proj.weight.grad.clone()
        proj.zero_grad()
        l2 = rnnt_amd.ShardedRNNTLoss(reduction="mean", loss_fn=fused_costs)(h2, d(labels), d(al), d(ll), global_batch=5)
        l2.backward()
        assert torch.equal(l1, l2) and torch.equal(h1.grad, h2.grad) and torch.equal(gw, proj.weight.grad)
    finally:
        dist.destroy_process_group()
