"""``FusedJointLoss`` — the joint network's output projection and the RNN-T loss as one operator
(SURVEY.md §8f rank 1).

The reference runs ``Linear(inner_dim -> vocab)`` as the last layer of ``JointNet.mlp``
(/root/reference/src/rnnt/model.py:14-18, applied at :59) and hands the ``[B,T,U+1,V]`` logits to
``RNNTLoss`` (:116).  For 16-bit operands this module does both with ``rnnt_joint_loss_forward``
(include/rnnt_mi355.h): one MFMA kernel writes the logits once and produces the log-softmax statistics
in its epilogue, so the loss's first pass over the logits (K1) does not exist.  Backward is
``rnnt_joint_loss_backward``: two MFMA kernels that form the loss's gradient tile by tile from the stored logits
(K3's arithmetic as their prologue) and multiply it at once, so the gradient w.r.t. the logits is never written
(inner size 512, vocabulary >= 128; other shapes: K3 followed by the two plain library GEMMs of a Linear layer's
backward).  On the fused-backward path the logits are kept with their rows padded to a multiple of 128 entries:
every row then starts on a 256-byte boundary, which the forward kernel's stores and the backward kernels' LDS-DMA
reads want (13 % of the forward at V = 4232, profiles/r03_tuning.md), and any vocabulary size is taken.
On that path the kernels also walk the COMPACTED node order (``rnnt_joint_loss_forward_compact``): only the lattice
nodes inside the utterances (t < act_lens[b], u <= label_lens[b]) are multiplied, stored and read back, numbered on the
device from the two length tensors; a ragged batch costs what its own nodes cost, not what the padded tensor would.

This is an EXTENSION next to the drop-in, not part of it: its inputs are the hidden activations and the
projection's parameters, not logits.  No CPU path, no fallback.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from .loss import _DTYPES, _REDUCTIONS, MAX_LATTICE_COLUMNS

SUPPORTED_HIDDEN = (128, 256, 512)
# Below this many lattice nodes B*T*(U+1) the fused kernel loses to the library GEMM + the three-pass loss on the
# same device: a workgroup walks the vocabulary tile by tile, and with few row tiles that chain (0.1-0.2 ms) is
# all there is.  Measured crossover at V=4232, H=512 (tools/bench_fused.py): 4 200 nodes 0.128 ms fused against
# 0.105 ms unfused, 33 600 nodes 0.269 against 0.297 (profiles/r02_bench_fused_B1.json).
DEFAULT_MIN_NODES = 16384


def fused_backward_supported(H: int, V: int) -> bool:
    """Shapes ``rnnt_joint_loss_backward`` takes (include/rnnt_mi355.h); others run K3 + two library GEMMs."""
    return H == 512 and V >= 128


def _row_stride(H: int, V: int, fused_backward: bool) -> int:
    """Elements from one node's row of the stored logits to the next."""
    if fused_backward and fused_backward_supported(H, V):
        return (V + 127) // 128 * 128
    return V  # K3 (rnnt_backward) reads contiguous rows


class _FusedJointLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hidden, weight, bias, labels, act_lens, label_lens, blank, reduction, fused_backward, valid_nodes,
                compact_order=True, zero_padding_grad=True):
        lib = _lib.load()
        B, T, U1, H = hidden.shape
        V = weight.shape[0]
        dev = hidden.device
        red = _REDUCTIONS[reduction]
        rows = B * T * U1
        # the compacted node order goes with the fused backward (K3 wants the logits of every row of the tensor)
        compact = bool(compact_order) and bool(fused_backward) and fused_backward_supported(H, V) and rows < 2 ** 31
        cap = rows if (valid_nodes is None or not compact) else max(1, min(int(valid_nodes), rows))
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            bias32 = bias.detach().to(torch.float32).contiguous()
            ld = _row_stride(H, V, fused_backward)
            logits = torch.empty((cap, ld), dtype=hidden.dtype, device=dev)  # (columns V .. ld-1: never touched)
            ws = torch.empty(_lib.workspace_bytes(T, U1, B), dtype=torch.uint8, device=dev)
            costs = torch.empty(B, dtype=torch.float32, device=dev)
            loss = torch.empty(B if red == _lib.REDUCTION_NONE else 1, dtype=torch.float32, device=dev)
            opts = _lib.RnntOptions(stream, int(blank), T, U1, _DTYPES[hidden.dtype], None)
            if compact:
                n = ctypes.c_size_t(0)
                _lib.check(lib.rnnt_joint_loss_compact_workspace_size(B, T, U1, ctypes.byref(n)),
                           "rnnt_joint_loss_compact_workspace_size")
                cws = torch.empty(int(n.value), dtype=torch.uint8, device=dev)
                _lib.check(lib.rnnt_joint_loss_forward_compact(hidden.data_ptr(), weight.data_ptr(), bias32.data_ptr(),
                                                               logits.data_ptr(), ld, cap, labels.data_ptr(),
                                                               act_lens.data_ptr(), label_lens.data_ptr(), H, V, B,
                                                               costs.data_ptr(), loss.data_ptr(), red, ws.data_ptr(),
                                                               cws.data_ptr(), opts),
                           "rnnt_joint_loss_forward_compact")
            else:
                cws = torch.empty(0, dtype=torch.uint8, device=dev)
                _lib.check(lib.rnnt_joint_loss_forward_strided(hidden.data_ptr(), weight.data_ptr(), bias32.data_ptr(),
                                                               logits.data_ptr(), ld, labels.data_ptr(),
                                                               act_lens.data_ptr(), label_lens.data_ptr(), H, V, B,
                                                               costs.data_ptr(), loss.data_ptr(), red, ws.data_ptr(), opts),
                           "rnnt_joint_loss_forward_strided")
        ctx.save_for_backward(hidden, weight, logits, labels, act_lens, label_lens, ws, cws)
        ctx.blank = int(blank)
        ctx.reduction = red
        ctx.bias_dtype = bias.dtype
        ctx.fused_backward = bool(fused_backward)
        ctx.compact = compact
        ctx.zero_padding_grad = bool(zero_padding_grad)
        ctx.dims = (B, T, U1, V, ld, cap)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        hidden, weight, logits, labels, act_lens, label_lens, ws, cws = ctx.saved_tensors
        lib = _lib.load()
        B, T, U1, V, ld, cap = ctx.dims
        H = hidden.shape[-1]
        dev = logits.device
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            go = grad_output.to(device=dev, dtype=torch.float32).contiguous().reshape(-1)
            if ctx.reduction == _lib.REDUCTION_NONE:
                stride, scale = 1, 1.0
            else:
                stride, scale = 0, (1.0 / B if ctx.reduction == _lib.REDUCTION_MEAN else 1.0)
            opts = _lib.RnntOptions(stream, ctx.blank, T, U1, _DTYPES[logits.dtype], None)
            need_h, need_w, need_b = ctx.needs_input_grad[:3]
            g_hidden = g_weight = g_bias = None
            if ctx.fused_backward and fused_backward_supported(H, V):
                # K3 as the prologue of the two products: the gradient w.r.t. the logits is never written
                n = ctypes.c_size_t(0)
                if ctx.compact:
                    _lib.check(lib.rnnt_joint_loss_backward_compact_workspace_size(H, V, cap, ctypes.byref(n)),
                               "rnnt_joint_loss_backward_compact_workspace_size")
                else:
                    _lib.check(lib.rnnt_joint_loss_backward_workspace_size(H, V, B, T, U1, ctypes.byref(n)),
                               "rnnt_joint_loss_backward_workspace_size")
                bws = torch.empty(int(n.value), dtype=torch.uint8, device=dev)
                if need_h:
                    g_hidden = torch.empty_like(hidden)
                if need_w:
                    g_weight = torch.empty_like(weight)
                gb32 = torch.empty(V, dtype=torch.float32, device=dev) if need_b else None
                outs = (g_hidden.data_ptr() if need_h else None, g_weight.data_ptr() if need_w else None,
                        gb32.data_ptr() if need_b else None)
                if ctx.compact:
                    _lib.check(lib.rnnt_joint_loss_backward_compact(
                        hidden.data_ptr(), weight.data_ptr(), logits.data_ptr(), ld, cap, labels.data_ptr(),
                        act_lens.data_ptr(), label_lens.data_ptr(), H, V, B, go.data_ptr(), stride,
                        ctypes.c_float(scale), outs[0], 1 if ctx.zero_padding_grad else 0, outs[1], outs[2],
                        ws.data_ptr(), cws.data_ptr(), bws.data_ptr(), opts),
                        "rnnt_joint_loss_backward_compact")
                else:
                    _lib.check(lib.rnnt_joint_loss_backward_strided(
                        hidden.data_ptr(), weight.data_ptr(), logits.data_ptr(), ld, labels.data_ptr(),
                        act_lens.data_ptr(), label_lens.data_ptr(), H, V, B, go.data_ptr(), stride,
                        ctypes.c_float(scale), *outs, ws.data_ptr(), bws.data_ptr(), opts),
                        "rnnt_joint_loss_backward_strided")
                if need_b:
                    g_bias = gb32.to(ctx.bias_dtype)
            else:
                assert ld == V
                g_logits = torch.empty_like(logits)
                _lib.check(lib.rnnt_backward(logits.data_ptr(), g_logits.data_ptr(), labels.data_ptr(),
                                             act_lens.data_ptr(), label_lens.data_ptr(), V, B, go.data_ptr(), stride,
                                             ctypes.c_float(scale), ws.data_ptr(), opts), "rnnt_backward")
                g2 = g_logits.view(-1, V)
                h2 = hidden.reshape(-1, H)
                if need_h:
                    g_hidden = torch.mm(g2, weight).view_as(hidden)        # plain library GEMMs (hipBLASLt)
                if need_w:
                    g_weight = torch.mm(g2.t(), h2)
                if need_b:
                    g_bias = g2.sum(0, dtype=torch.float32).to(ctx.bias_dtype)
        return g_hidden, g_weight, g_bias, None, None, None, None, None, None, None, None, None


def fused_joint_loss_skips_padding(hidden, weight, min_nodes=DEFAULT_MIN_NODES) -> bool:
    """Whether :func:`fused_joint_loss` on these operands walks the compacted node order, i.e. never reads the rows of
    ``hidden`` outside the lattices (and may leave their gradient unwritten, ``zero_padding_grad=False``)."""
    B, T, U1, H = hidden.shape
    rows = B * T * U1
    return (hidden.dtype in (torch.bfloat16, torch.float16) and fused_backward_supported(H, weight.shape[0])
            and min_nodes <= rows < 2 ** 31)


def count_lattice_nodes(act_lens, label_lens, maxT=None, maxU1=None) -> int:
    """Sum of ``T_b * (U_b + 1)`` over the batch from HOST copies of the two length tensors (any integer sequences or
    CPU tensors): the ``valid_nodes`` argument of :func:`fused_joint_loss`.  The reference has both on the host when
    it calls the loss (``enc_output_lengths`` comes out of ``pad_packed_sequence``, ``tokens_length`` out of the data
    loader: /root/reference/src/rnnt/model.py:104-116), so the count costs no device round trip there.
    With ``maxT`` / ``maxU1`` (the lattice tensor's frames and columns) utterances whose lengths do not fit are left
    out, as the kernels leave them out: the count is then exact, not an upper bound."""
    t = torch.as_tensor(act_lens, device="cpu").to(torch.int64)
    u = torch.as_tensor(label_lens, device="cpu").to(torch.int64)
    n = t.clamp_min(0) * (u.clamp_min(0) + 1)
    if maxT is not None and maxU1 is not None:
        n = n * ((t >= 1) & (t <= int(maxT)) & (u >= 0) & (u < int(maxU1)))
    return int(n.sum())


def fused_joint_loss(hidden, weight, bias, labels, act_lens, label_lens, blank=0, reduction="mean",
                     min_nodes=DEFAULT_MIN_NODES, fused_backward=True, valid_nodes=None, compact_order=True,
                     zero_padding_grad=True):
    """``RNNTLoss(blank, reduction)(hidden @ weight.T + bias, labels, act_lens, label_lens)`` in one pass.

    Lattices of fewer than ``min_nodes`` nodes (``B*T*(U+1)``) are computed as exactly that composition instead
    — the library GEMM and ``rnnt_amd.rnnt_loss`` on the same device, which is the faster forward there
    (``DEFAULT_MIN_NODES``); ``min_nodes=0`` always runs the fused kernel.

    Args:
        hidden: ``[B, T, U+1, H]`` bf16 or fp16, contiguous: what ``JointNet``'s ``Tanh`` returns
            (``rnnt_amd.joint_hidden``); ``H`` in ``SUPPORTED_HIDDEN``.
        weight, bias: the parameters of ``Linear(H -> V)`` (``mlp.2`` of the reference's ``JointNet``);
            weight ``[V, H]`` (another floating dtype than ``hidden``'s, e.g. an fp32 parameter under autocast, is cast).
        labels, act_lens, label_lens, blank, reduction: as :func:`rnnt_amd.rnnt_loss`.
        valid_nodes: optional upper bound of the number of lattice nodes ``sum_b T_b (U_b + 1)``, known on the host
            (:func:`count_lattice_nodes`).  The kernels skip the padding of a ragged batch either way (they number
            the nodes on the device); with the bound the logits buffer and the launch grids are cut for the nodes
            instead of for ``B*T*(U+1)`` rows.  A bound below the true count makes the affected costs NaN.
        compact_order: False runs the kernels over every row of the tensor, padding included (for A/B runs).
        zero_padding_grad: False leaves the rows of ``hidden.grad`` outside the lattices UNINITIALISED instead of
            zero — only for a producer of ``hidden`` whose backward skips those rows (``joint_hidden`` with lengths);
            honoured on the compacted path only (``fused_joint_loss_skips_padding``).
    """
    if reduction not in _REDUCTIONS:
        raise ValueError("reduction must be one of 'none', 'mean', 'sum'")
    if hidden.dtype not in (torch.bfloat16, torch.float16):
        raise TypeError("fused_joint_loss takes bfloat16 or float16 activations (fp32: use JointNet + RNNTLoss; "
                        "the fp32 matrix rate of the MI355X is 1/16 of the 16-bit one)")
    if bias is None:
        raise ValueError("fused_joint_loss needs the projection's bias (nn.Linear(bias=True)); pass a zero vector for a "
                         "projection without one")
    if weight.dtype is not hidden.dtype:
        if not weight.is_floating_point():
            raise TypeError("weight must be a floating-point tensor")
        weight = weight.to(hidden.dtype)  # e.g. an fp32 parameter under autocast: its gradient comes back through the cast
    if hidden.dim() != 4 or weight.dim() != 2 or bias.dim() != 1:
        raise ValueError("hidden must be 4D, weight 2D, bias 1D")
    if not hidden.is_cuda:
        raise RuntimeError("rnnt_amd.fused_joint_loss computes on MI355X (ROCm device tensors) only; "
                           "there is no CPU path or PyTorch fallback in this package.")
    B, T, U1, H = hidden.shape
    V = weight.shape[0]
    if weight.shape[1] != H or bias.shape[0] != V:
        raise ValueError("weight must be [V, H] and bias [V]")
    if H not in SUPPORTED_HIDDEN:
        raise ValueError(f"fused_joint_loss supports inner sizes {SUPPORTED_HIDDEN}, got {H}")
    if V < 64:
        raise ValueError(f"fused_joint_loss needs a vocabulary of at least 64 entries (one tile), got {V}")
    if U1 > MAX_LATTICE_COLUMNS:
        raise ValueError(f"rnnt_amd supports at most {MAX_LATTICE_COLUMNS - 1} labels per utterance")
    for name, t in (("labels", labels), ("lengths", act_lens), ("label_lengths", label_lens)):
        if t.dtype is not torch.int32:
            raise TypeError(f"{name} must be torch.int32")
        if t.device != hidden.device:
            raise RuntimeError(f"{name} must be on the same device as hidden")
        if not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous")
    if labels.dim() != 2 or labels.shape[0] != B or labels.shape[1] != U1 - 1:
        raise ValueError("Output length mismatch")
    if act_lens.shape != (B,) or label_lens.shape != (B,):
        raise ValueError("must have a length per example.")
    if not (0 <= int(blank) < V):
        raise ValueError("blank label out of range")
    if not hidden.is_contiguous():
        hidden = hidden.contiguous()
    if not weight.is_contiguous():
        weight = weight.contiguous()
    if B * T * U1 < min_nodes:
        from .loss import rnnt_loss

        # as the kernel: fp32 accumulation, the bias added in fp32 (whatever its dtype), one rounding to the 16-bit type
        logits = torch.nn.functional.linear(hidden.float(), weight.float(), bias.float()).to(hidden.dtype)
        return rnnt_loss(logits, labels, act_lens, label_lens, blank=blank, reduction=reduction, check_lengths=False)
    return _FusedJointLossFunction.apply(hidden, weight, bias, labels, act_lens, label_lens, blank, reduction,
                                         fused_backward, valid_nodes, compact_order, zero_padding_grad)


class FusedJointLoss(torch.nn.Module):
    """``Linear(H -> V)`` of the joint network + ``RNNTLoss`` as one module; owns no parameters: pass the
    projection (``joint.mlp[2]`` of the reference's or this package's ``JointNet``) at call time."""

    def __init__(self, blank: int = 0, reduction: str = "mean", min_nodes: int = DEFAULT_MIN_NODES,
                 fused_backward: bool = True, compact_order: bool = True):
        super().__init__()
        if reduction not in _REDUCTIONS:
            raise ValueError("reduction must be one of 'none', 'mean', 'sum'")
        self.blank = blank
        self.reduction = reduction
        self.min_nodes = min_nodes  # smaller lattices take the library GEMM + RNNTLoss (see fused_joint_loss)
        self.fused_backward = fused_backward  # False: K3 + two library GEMMs (the round-2 backward), for A/B runs
        self.compact_order = compact_order    # False: every row of the tensor, padding included, for A/B runs

    def forward(self, hidden, projection: torch.nn.Linear, labels, act_lens, label_lens, valid_nodes=None):
        return fused_joint_loss(hidden, projection.weight, projection.bias, labels, act_lens, label_lens,
                                self.blank, self.reduction, self.min_nodes, self.fused_backward, valid_nodes,
                                self.compact_order)
