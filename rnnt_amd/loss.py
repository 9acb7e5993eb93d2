"""``RNNTLoss`` — drop-in for ``warprnnt_pytorch.RNNTLoss`` on MI355X.

Mirrors the operator the reference constructs and calls at
/root/reference/src/rnnt/model.py:87 (``RNNTLoss()``) and :116
(``loss(acts, labels.int(), act_lens, label_lens.int())``):
same positional arguments, same argument checks and exception types, same
``[1]``-shaped fp32 result for the default ``reduction='mean'``.

The work is done by ``librnnt_mi355.so`` through its C ABI
(``include/rnnt_mi355.h``): ``rnnt_forward`` in ``forward`` (K1 + K2) and
``rnnt_backward`` in ``backward`` (K3, with ``grad_output`` and the ``1/B`` of
'mean' fused in).  PyTorch only provides device memory and the current stream.
There is no CPU path and no fallback: CPU tensors raise.

Two bindings drive that C ABI.  The default, ``"torch"``, is the compiled extension
``librnnt_amd_torch.so`` (``rnnt_amd/csrc/torch_binding.cpp``): one ``torch.ops.rnnt_amd.rnnt_loss`` call
per step, backward recorded as a C++ autograd node.  ``"ctypes"`` is the same sequence written in Python
(``_RNNTLossFunction`` below), kept as the worked example of ``INTEGRATION.md`` §2 and for A/B runs
(``tools/bench_host_overhead.py``); select it with ``set_binding("ctypes")`` or ``binding="ctypes"``.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib, _torch_binding

_DTYPES = {torch.float32: _lib.DTYPE_F32, torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}
_REDUCTIONS = {"none": _lib.REDUCTION_NONE, "mean": _lib.REDUCTION_MEAN, "sum": _lib.REDUCTION_SUM}

MAX_LATTICE_COLUMNS = 4096  # maxU = U + 1 the library accepts (include/rnnt_mi355.h: RNNT_MAX_LATTICE_COLUMNS)

# bench.py sets these to _lib.StageEvents instances to time K1/K2/K3 with hipEvents recorded by
# the library on the launch stream.  None in normal use.
_profile_fwd: Optional[_lib.StageEvents] = None
_profile_bwd: Optional[_lib.StageEvents] = None


_binding = "torch"


def set_binding(name: str) -> None:
    """Process-wide default binding of :func:`rnnt_loss`: ``"torch"`` (compiled extension) or ``"ctypes"``."""
    global _binding
    if name not in ("torch", "ctypes"):
        raise ValueError("binding must be 'torch' or 'ctypes'")
    _binding = name


def get_binding() -> str:
    return _binding


def set_stage_events(fwd: Optional[_lib.StageEvents], bwd: Optional[_lib.StageEvents]) -> None:
    global _profile_fwd, _profile_bwd
    _profile_fwd, _profile_bwd = fwd, bwd


# --- argument checks: names, order and exception types follow upstream's certify_inputs -----------
def _check_type(var, t, name):
    if var.dtype is not t:
        raise TypeError("{} must be {}".format(name, t))


def _check_contiguous(var, name):
    if not var.is_contiguous():
        raise ValueError("{} must be contiguous".format(name))


def _check_dim(var, dim, name):
    if len(var.shape) != dim:
        raise ValueError("{} must be {}D".format(name, dim))


def certify_inputs(acts, labels, act_lens, label_lens, check_lengths=True):
    if acts.dtype not in _DTYPES:
        raise TypeError("acts must be torch.float32, torch.bfloat16 or torch.float16")
    _check_type(labels, torch.int32, "labels")
    _check_type(label_lens, torch.int32, "label_lengths")
    _check_type(act_lens, torch.int32, "lengths")
    _check_contiguous(acts, "acts")
    _check_contiguous(labels, "labels")
    _check_contiguous(label_lens, "label_lengths")
    _check_contiguous(act_lens, "lengths")
    _check_dim(acts, 4, "acts")
    _check_dim(labels, 2, "labels")
    _check_dim(act_lens, 1, "lenghts")
    _check_dim(label_lens, 1, "label_lenghts")
    if act_lens.shape[0] != acts.shape[0]:
        raise ValueError("must have a length per example.")
    if label_lens.shape[0] != acts.shape[0]:
        raise ValueError("must have a label length per example.")
    if labels.shape[0] != acts.shape[0]:
        raise ValueError("must have a label row per example.")
    if acts.shape[0] < 1 or acts.shape[1] < 1 or acts.shape[2] < 1 or acts.shape[3] < 1:
        raise ValueError("acts must not be empty")
    if labels.shape[1] != acts.shape[2] - 1:
        raise ValueError("Output length mismatch")  # labels must be [B, U] for acts [B, T, U+1, V]
    if acts.shape[2] > MAX_LATTICE_COLUMNS:
        # a limit upstream does not have (the wide alpha/beta kernel keeps two anti-diagonals per direction in
        # 64 KiB of LDS).  ValueError, not the C ABI's RuntimeError: the reference's loop takes every RuntimeError
        # for an out-of-memory condition and retries utterance by utterance (bin/train.py:80-91).
        raise ValueError("rnnt_amd supports at most {} labels per utterance (acts.shape[2] <= {}), got "
                         "acts.shape[2] = {}".format(MAX_LATTICE_COLUMNS - 1, MAX_LATTICE_COLUMNS, acts.shape[2]))
    if check_lengths:
        _validate_lengths(*_start_length_fetch(acts, act_lens, label_lens))


_copy_streams = {}  # device index -> the side stream the length check copies on


def _start_length_fetch(acts, act_lens, label_lens):
    """Enqueue a copy of both length vectors to pinned host memory; returns what the check needs.

    Upstream synchronises on ``torch.max(lengths)`` before its native call.  Here the two copies run on a side
    stream (ordered behind whatever produced the lengths on the current one) and are waited for AFTER the kernels
    are launched: the launch stream carries no extra work and the host check never drains it (the kernels guard
    themselves against bad lengths, so launching first is safe).
    """
    B = acts.shape[0]
    if act_lens.is_cuda:
        dev = act_lens.device
        side = _copy_streams.get(dev.index)
        if side is None:
            side = _copy_streams.setdefault(dev.index, torch.cuda.Stream(device=dev))
        host = torch.empty(2 * B, dtype=torch.int32, pin_memory=True)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            host[:B].copy_(act_lens.reshape(-1), non_blocking=True)
            host[B:].copy_(label_lens.reshape(-1), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        # (act_lens / label_lens stay referenced by the caller until _validate_lengths has waited for `ev`)
    else:
        host, ev = torch.cat((act_lens.reshape(-1), label_lens.reshape(-1))), None
    return host, ev, B, acts.shape[1], acts.shape[2]


def _validate_lengths(host, ev, B, T, U1):
    """Upstream demands equality; a batch shard of a padded batch legitimately has T > max(lengths)
    (SURVEY.md §4, §8b), so the tensor only has to be large enough."""
    if ev is not None:
        ev.synchronize()
    al, ll = host[:B], host[B:]
    if int(al.max()) > T or int(al.min()) < 1:
        raise ValueError("Input length mismatch")
    if int(ll.max()) + 1 > U1 or int(ll.min()) < 0:
        raise ValueError("Output length mismatch")


class _RNNTLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, acts, labels, act_lens, label_lens, blank, reduction):
        lib = _lib.load()
        B, T, U1, V = acts.shape
        dev = acts.device
        red = _REDUCTIONS[reduction]
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            ws = torch.empty(_lib.workspace_bytes(T, U1, B), dtype=torch.uint8, device=dev)
            costs = torch.empty(B, dtype=torch.float32, device=dev)
            loss = torch.empty(B if red == _lib.REDUCTION_NONE else 1, dtype=torch.float32, device=dev)
            opts = _lib.RnntOptions(stream, int(blank), T, U1, _DTYPES[acts.dtype],
                                    _profile_fwd.ptr() if _profile_fwd is not None else None)
            _lib.check(lib.rnnt_forward(acts.data_ptr(), labels.data_ptr(), act_lens.data_ptr(),
                                        label_lens.data_ptr(), V, B, costs.data_ptr(), loss.data_ptr(),
                                        red, ws.data_ptr(), opts), "rnnt_forward")
        ctx.save_for_backward(acts, labels, act_lens, label_lens, ws)
        ctx.blank = int(blank)
        ctx.reduction = red
        ctx.mark_non_differentiable(labels, act_lens, label_lens)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        acts, labels, act_lens, label_lens, ws = ctx.saved_tensors
        lib = _lib.load()
        B, T, U1, V = acts.shape
        dev = acts.device
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            go = grad_output.to(device=dev, dtype=torch.float32).contiguous().reshape(-1)
            if ctx.reduction == _lib.REDUCTION_NONE:
                if go.numel() != B:
                    raise RuntimeError("grad_output must have one element per utterance")
                stride, scale = 1, 1.0
            else:
                stride, scale = 0, (1.0 / B if ctx.reduction == _lib.REDUCTION_MEAN else 1.0)
            grads = torch.empty_like(acts)
            opts = _lib.RnntOptions(stream, ctx.blank, T, U1, _DTYPES[acts.dtype],
                                    _profile_bwd.ptr() if _profile_bwd is not None else None)
            _lib.check(lib.rnnt_backward(acts.data_ptr(), grads.data_ptr(), labels.data_ptr(),
                                         act_lens.data_ptr(), label_lens.data_ptr(), V, B, go.data_ptr(),
                                         stride, ctypes.c_float(scale), ws.data_ptr(), opts),
                       "rnnt_backward")
        return grads, None, None, None, None, None


def reduce_costs(costs: torch.Tensor, reduction: str = "mean") -> torch.Tensor:
    """``sum(costs)/B`` ('mean'), ``sum`` or a copy, summed in index order on the device.

    Same kernel ``rnnt_forward`` uses for its ``loss_out``; no host synchronisation.
    """
    if not costs.is_cuda:
        raise RuntimeError("rnnt_amd.reduce_costs runs on ROCm device tensors only")
    if costs.dtype is not torch.float32 or costs.dim() != 1 or not costs.is_contiguous():
        raise ValueError("costs must be a contiguous 1-D float32 tensor")
    red = _REDUCTIONS[reduction]
    B = costs.shape[0]
    with torch.cuda.device(costs.device):
        out = torch.empty(B if red == _lib.REDUCTION_NONE else 1, dtype=torch.float32, device=costs.device)
        _lib.check(_lib.load().rnnt_reduce_costs(costs.data_ptr(), out.data_ptr(), B, red,
                                                 torch.cuda.current_stream(costs.device).cuda_stream),
                   "rnnt_reduce_costs")
    return out


def _apply(acts, labels, act_lens, label_lens, blank, reduction, binding):
    if binding == "ctypes":
        return _RNNTLossFunction.apply(acts, labels, act_lens, label_lens, blank, reduction)
    return _torch_binding.load().rnnt_loss(
        acts, labels, act_lens, label_lens, int(blank), _REDUCTIONS[reduction],
        _profile_fwd.address() if _profile_fwd is not None else 0,
        _profile_bwd.address() if _profile_bwd is not None else 0)


def rnnt_loss(acts, labels, act_lens, label_lens, blank=0, reduction="mean", check_lengths=True, binding=None):
    """RNN Transducer loss (functional form of :class:`RNNTLoss`).

    Args:
        acts: ``[B, T, U+1, V]`` raw joint-network logits (log-softmax is applied inside).
        labels: ``[B, U]`` int32, zero padded, without BOS.
        act_lens: ``[B]`` int32 frames per utterance.
        label_lens: ``[B]`` int32 labels per utterance.
        blank: blank label index (default 0, as the reference uses).
        reduction: ``'mean'`` (sum / B, shape ``[1]``), ``'sum'`` (``[1]``) or ``'none'`` (``[B]``).
        binding: ``"torch"`` | ``"ctypes"`` | None (the process default, see :func:`set_binding`).
    """
    binding = _binding if binding is None else binding
    if binding not in ("torch", "ctypes"):
        raise ValueError("binding must be 'torch' or 'ctypes'")
    if reduction not in _REDUCTIONS:
        raise ValueError("reduction must be one of 'none', 'mean', 'sum'")
    certify_inputs(acts, labels, act_lens, label_lens, check_lengths=False)
    if not acts.is_cuda:
        raise RuntimeError(
            "rnnt_amd.RNNTLoss computes on MI355X (ROCm device tensors) only; got acts on "
            f"{acts.device}. There is no CPU path or PyTorch fallback in this package.")
    for name, t in (("labels", labels), ("lengths", act_lens), ("label_lengths", label_lens)):
        if t.device != acts.device:
            raise RuntimeError(f"{name} must be on the same device as acts ({acts.device}), got {t.device}")
    if not (0 <= int(blank) < acts.shape[3]):
        raise ValueError("blank label out of range")
    if not check_lengths:
        return _apply(acts, labels, act_lens, label_lens, blank, reduction, binding)
    with torch.cuda.device(acts.device):
        pending = _start_length_fetch(acts, act_lens, label_lens)
        loss = _apply(acts, labels, act_lens, label_lens, blank, reduction, binding)
        _validate_lengths(*pending)  # raises ValueError like upstream; the launched kernels were harmless
    return loss


class RNNTLoss(torch.nn.Module):
    """Drop-in for ``warprnnt_pytorch.RNNTLoss``; the reference builds it with no arguments.

    Parameters:
        blank (int, default 0): blank label.
        reduction (str, default 'mean'): 'none' | 'mean' | 'sum'; 'mean' divides by the batch size.
        binding (str or None): ``"torch"`` (compiled extension, the default) or ``"ctypes"``.
        check_lengths (bool, default True): verify on the host that the length tensors fit the
            logits tensor (one tiny device-to-host copy, as upstream).  The kernels guard
            themselves either way: an utterance with out-of-range lengths or labels yields a NaN
            cost and zero gradient instead of an out-of-bounds access.
    """

    def __init__(self, blank: int = 0, reduction: str = "mean", check_lengths: bool = True, binding=None):
        super().__init__()
        if reduction not in _REDUCTIONS:
            raise ValueError("reduction must be one of 'none', 'mean', 'sum'")
        if binding not in (None, "torch", "ctypes"):
            raise ValueError("binding must be 'torch' or 'ctypes'")
        self.blank = blank
        self.reduction = reduction
        self.check_lengths = check_lengths
        self.binding = binding  # None: the process default (rnnt_amd.loss.set_binding)

    def forward(self, acts, labels, act_lens, label_lens):
        return rnnt_loss(acts, labels, act_lens, label_lens, self.blank, self.reduction,
                         self.check_lengths, self.binding)

    def extra_repr(self) -> str:
        return f"blank={self.blank}, reduction='{self.reduction}'"
