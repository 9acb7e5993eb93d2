"""The loss on logits kept in the COMPACTED node order — for ragged batches whose logits the caller produces itself.

The reference's joint computes the logits of all ``B*T*(U+1)`` rows (/root/reference/src/rnnt/model.py:46-59) and the
loss it calls (:116) ignores the rows outside the utterances' lattices (``t >= act_lens[b]`` or ``u > label_lens[b]``).
``compact_rowmap`` numbers the nodes INSIDE the lattices on the device (utterance by utterance, frame by frame, label
position fastest) and returns, per node, its row of the ``[B, T, U+1]`` tensor; the caller gathers those hidden vectors,
projects only them (a library GEMM over ``n`` rows instead of ``B*T*(U+1)``) and hands the ``[n, V]`` logits to
``rnnt_loss_compact``: ``rnnt_forward_compact`` / ``rnnt_backward_compact`` of include/rnnt_mi355.h, the loss's three
passes with one row per node.  Same costs as ``rnnt_loss`` on the padded tensor, bit for bit; the gradient holds the
node rows of ``rnnt_loss``'s, bit for bit (rows of whole 128-byte lines, longer than 2 KiB; otherwise to the last bit of each row's
log-sum), and no zeros for padding are written.  ``rnnt_amd.JointNet.transducer_loss``
takes this route for fp32 states (the reference's own precision) when the lengths arrive on the host; its own hidden
layer is written in the compacted order directly (``rnnt_amd.joint_hidden(..., nodes=n)``), so it needs no map.
Rows shorter than 2 KiB (V < 512 in fp32): the padded call batches several nodes per wave, this one takes one wave per
node — at V = 29 / 128 the three passes alone are 2.5x / 1.8x slower than on the padded tensor (0.28 against 0.11 / 0.16 ms
at B32 T400 U40, 38 % nodes), faster from V = 512 on; what the compacted order saves around the loss (hidden layer and
projection over n rows) outweighs that in ``JointNet.transducer_loss``.
No CPU path, no fallback.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from .loss import _DTYPES, _REDUCTIONS, MAX_LATTICE_COLUMNS


def compact_rowmap(act_lens: torch.Tensor, label_lens: torch.Tensor, maxT: int, maxU1: int, nodes: int) -> torch.Tensor:
    """int32 ``[nodes]`` on the lengths' device: entry ``k`` = row ``(b*maxT + t)*maxU1 + u`` of compacted node ``k``.

    ``nodes``: the number of lattice nodes, known on the host (``rnnt_amd.count_lattice_nodes``); entries past the
    true count are left uninitialised, nodes past ``nodes`` are dropped."""
    if not act_lens.is_cuda or label_lens.device != act_lens.device:
        raise RuntimeError("compact_rowmap numbers the nodes on the MI355X: the length tensors must be on one ROCm device")
    for name, t in (("act_lens", act_lens), ("label_lens", label_lens)):
        if t.dtype is not torch.int32 or t.dim() != 1 or not t.is_contiguous():
            raise TypeError(f"{name} must be a contiguous 1-D torch.int32 tensor")
    B = act_lens.shape[0]
    if label_lens.shape[0] != B or nodes < 1:
        raise ValueError("one length per utterance and at least one node")
    dev = act_lens.device
    with torch.cuda.device(dev):
        rowmap = torch.empty(int(nodes), dtype=torch.int32, device=dev)
        scratch = torch.empty(16 * (B + 1), dtype=torch.uint8, device=dev)
        _lib.check(_lib.load().rnnt_compact_rows(act_lens.data_ptr(), label_lens.data_ptr(), B, int(maxT), int(maxU1),
                                                 int(nodes), rowmap.data_ptr(), scratch.data_ptr(),
                                                 torch.cuda.current_stream(dev).cuda_stream), "rnnt_compact_rows")
    return rowmap


class _CompactLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, acts, labels, act_lens, label_lens, maxT, blank, reduction, exact_rows):
        lib = _lib.load()
        n, V = acts.shape
        B, U1 = labels.shape[0], labels.shape[1] + 1
        dev = acts.device
        red = _REDUCTIONS[reduction]
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            ws = torch.empty(_lib.workspace_bytes(maxT, U1, B), dtype=torch.uint8, device=dev)
            costs = torch.empty(B, dtype=torch.float32, device=dev)
            loss = torch.empty(B if red == _lib.REDUCTION_NONE else 1, dtype=torch.float32, device=dev)
            opts = _lib.RnntOptions(stream, int(blank), int(maxT), U1, _DTYPES[acts.dtype], None)
            _lib.check(lib.rnnt_forward_compact(acts.data_ptr(), n, labels.data_ptr(), act_lens.data_ptr(),
                                                label_lens.data_ptr(), V, B, costs.data_ptr(), loss.data_ptr(), red,
                                                ws.data_ptr(), opts), "rnnt_forward_compact")
        ctx.save_for_backward(acts, labels, act_lens, label_lens, ws)
        ctx.args = (int(maxT), int(blank), red)
        ctx.slack = not exact_rows
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        acts, labels, act_lens, label_lens, ws = ctx.saved_tensors
        maxT, blank, red = ctx.args
        lib = _lib.load()
        n, V = acts.shape
        B, U1 = labels.shape[0], labels.shape[1] + 1
        dev = acts.device
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            go = grad_output.to(device=dev, dtype=torch.float32).contiguous().reshape(-1)
            if red == _lib.REDUCTION_NONE:
                if go.numel() != B:
                    raise RuntimeError("grad_output must have one element per utterance")
                stride, scale = 1, 1.0
            else:
                stride, scale = 0, (1.0 / B if red == _lib.REDUCTION_MEAN else 1.0)
            # rows past the last node (a buffer with slack) belong to no utterance: zero, like the padding of the tensor
            grads = torch.zeros_like(acts) if ctx.slack else torch.empty_like(acts)
            opts = _lib.RnntOptions(stream, blank, maxT, U1, _DTYPES[acts.dtype], None)
            _lib.check(lib.rnnt_backward_compact(acts.data_ptr(), grads.data_ptr(), n, labels.data_ptr(),
                                                 act_lens.data_ptr(), label_lens.data_ptr(), V, B, go.data_ptr(), stride,
                                                 ctypes.c_float(scale), ws.data_ptr(), opts), "rnnt_backward_compact")
        return grads, None, None, None, None, None, None, None


def rnnt_loss_compact(acts, labels, act_lens, label_lens, maxT, blank=0, reduction="mean", exact_rows=False):
    """``rnnt_amd.rnnt_loss`` on ``acts`` ``[n, V]``: the logits of the lattice nodes only, in the compacted order
    (row ``k`` = node ``k`` of :func:`compact_rowmap`).

    Args:
        acts: ``[n, V]`` fp32 / bf16 / fp16, contiguous, on the MI355X; ``n`` >= the number of nodes
            ``sum_b act_lens[b] * (label_lens[b] + 1)`` (a smaller ``n`` gives NaN costs for the utterances it cuts).
        labels: ``[B, U]`` int32 (``U + 1`` is the width of the lattice tensor the map was made for).
        act_lens, label_lens: ``[B]`` int32.
        maxT: frames of the lattice tensor the map was made for.
        exact_rows: ``n`` is exactly the number of nodes (the gradient then needs no zero-fill for surplus rows).
    Returns the loss as :func:`rnnt_amd.rnnt_loss` does; ``acts.grad`` gets ``[n, V]``.
    """
    if reduction not in _REDUCTIONS:
        raise ValueError("reduction must be one of 'none', 'mean', 'sum'")
    if acts.dtype not in _DTYPES:
        raise TypeError("acts must be torch.float32, torch.bfloat16 or torch.float16")
    if not acts.is_cuda:
        raise RuntimeError("rnnt_amd.rnnt_loss_compact computes on MI355X (ROCm device tensors) only; there is no CPU "
                           "path or PyTorch fallback in this package")
    if acts.dim() != 2 or labels.dim() != 2 or acts.shape[0] < 1:
        raise ValueError("acts must be [n, V] with n >= 1 and labels [B, U]")
    for name, t in (("labels", labels), ("lengths", act_lens), ("label_lengths", label_lens)):
        if t.dtype is not torch.int32:
            raise TypeError(f"{name} must be torch.int32")
        if t.device != acts.device:
            raise RuntimeError(f"{name} must be on the same device as acts")
        if not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous")
    B = labels.shape[0]
    if act_lens.shape != (B,) or label_lens.shape != (B,):
        raise ValueError("must have a length per example.")
    if labels.shape[1] + 1 > MAX_LATTICE_COLUMNS:
        raise ValueError(f"rnnt_amd supports at most {MAX_LATTICE_COLUMNS - 1} labels per utterance")
    if not (0 <= int(blank) < acts.shape[1]) or int(maxT) < 1:
        raise ValueError("blank label or maxT out of range")
    if B * int(maxT) * (labels.shape[1] + 1) >= 2 ** 31:
        raise ValueError("the lattice tensor must have fewer than 2^31 rows")
    if not acts.is_contiguous():
        acts = acts.contiguous()
    return _CompactLossFunction.apply(acts, labels, act_lens, label_lens, int(maxT), int(blank), reduction,
                                      bool(exact_rows))
