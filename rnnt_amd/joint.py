"""``JointNet`` with its hidden layer computed the MI355X way.

Mirror of the reference's ``JointNet`` (/root/reference/src/rnnt/model.py:9-63): same constructor
arguments, same parameter names (``mlp.0.weight`` ``[inner, input]``, ``mlp.0.bias``,
``mlp.2.weight`` ``[vocab, inner]``, ``mlp.2.bias``) so a reference checkpoint loads with
``load_state_dict``, same ``forward(enc_state, dec_state, softmax=False)`` returning the raw logits
``[B, T, U+1, V]`` the loss takes (model.py:115-116).

What changes is how the training path (3-D inputs) gets there.  The reference repeats both states
to ``[B,T,U+1,·]``, concatenates them (model.py:49-54) and runs ``Linear(He+Hd -> H)`` and ``Tanh``
over all B·T·(U+1) rows.  Here ``Linear(cat(e, d)) = W_e e + W_d d + b`` is evaluated as two small
GEMMs on ``[B,T,He]`` and ``[B,U+1,Hd]`` and one HIP kernel forms
``tanh(enc_proj[b,t] + dec_proj[b,u])`` (``rnnt_joint_hidden_forward``); backward is one streaming
kernel plus a small ordered reduction (``rnnt_joint_hidden_backward``).  The output projection
``Linear(H -> V)`` stays a plain library GEMM.  No CPU path for the 3-D case: CPU tensors raise.
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

_DTYPES = {torch.float32: _lib.DTYPE_F32, torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}


def _workspace_bytes(B, T, U1, H) -> int:
    n = ctypes.c_size_t(0)
    _lib.check(_lib.load().rnnt_joint_hidden_workspace_size(B, T, U1, H, ctypes.byref(n)),
               "rnnt_joint_hidden_workspace_size")
    return int(n.value)


class _JointHidden(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc_proj, dec_proj, act_lens=None, label_lens=None):
        B, T, H = enc_proj.shape
        U1 = dec_proj.shape[1]
        dev = enc_proj.device
        enc_proj = enc_proj.contiguous()
        dec_proj = dec_proj.contiguous()
        ctx.ragged = act_lens is not None
        with torch.cuda.device(dev):
            hidden = torch.empty((B, T, U1, H), dtype=enc_proj.dtype, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            if ctx.ragged:  # rows outside the lattices are neither written here nor read in backward
                _lib.check(_lib.load().rnnt_joint_hidden_forward_ragged(
                    enc_proj.data_ptr(), dec_proj.data_ptr(), hidden.data_ptr(), act_lens.data_ptr(),
                    label_lens.data_ptr(), B, T, U1, H, _DTYPES[enc_proj.dtype], stream),
                    "rnnt_joint_hidden_forward_ragged")
                ctx.save_for_backward(hidden, act_lens, label_lens)
            else:
                _lib.check(_lib.load().rnnt_joint_hidden_forward(
                    enc_proj.data_ptr(), dec_proj.data_ptr(), hidden.data_ptr(), B, T, U1, H,
                    _DTYPES[enc_proj.dtype], stream), "rnnt_joint_hidden_forward")
                ctx.save_for_backward(hidden)
        return hidden

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_hidden):
        hidden = ctx.saved_tensors[0]
        B, T, U1, H = hidden.shape
        dev = hidden.device
        grad_hidden = grad_hidden.to(hidden.dtype).contiguous()
        with torch.cuda.device(dev):
            grad_enc = torch.empty((B, T, H), dtype=hidden.dtype, device=dev)
            grad_dec = torch.empty((B, U1, H), dtype=hidden.dtype, device=dev)
            ws = torch.empty(_workspace_bytes(B, T, U1, H), dtype=torch.uint8, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            if ctx.ragged:
                _, act_lens, label_lens = ctx.saved_tensors
                _lib.check(_lib.load().rnnt_joint_hidden_backward_ragged(
                    hidden.data_ptr(), grad_hidden.data_ptr(), grad_enc.data_ptr(), grad_dec.data_ptr(),
                    act_lens.data_ptr(), label_lens.data_ptr(), B, T, U1, H, _DTYPES[hidden.dtype], ws.data_ptr(),
                    stream), "rnnt_joint_hidden_backward_ragged")
            else:
                _lib.check(_lib.load().rnnt_joint_hidden_backward(
                    hidden.data_ptr(), grad_hidden.data_ptr(), grad_enc.data_ptr(), grad_dec.data_ptr(), B, T, U1, H,
                    _DTYPES[hidden.dtype], ws.data_ptr(), stream), "rnnt_joint_hidden_backward")
        return grad_enc, grad_dec, None, None


class _JointHiddenCompact(torch.autograd.Function):
    """``joint_hidden`` with the result in the compacted node order: ``[nodes, H]``, one row per lattice node."""

    @staticmethod
    def forward(ctx, enc_proj, dec_proj, act_lens, label_lens, nodes):
        B, T, H = enc_proj.shape
        U1 = dec_proj.shape[1]
        dev = enc_proj.device
        enc_proj = enc_proj.contiguous()
        dec_proj = dec_proj.contiguous()
        with torch.cuda.device(dev):
            hidden = torch.empty((int(nodes), H), dtype=enc_proj.dtype, device=dev)
            _lib.check(_lib.load().rnnt_joint_hidden_forward_compact(
                enc_proj.data_ptr(), dec_proj.data_ptr(), hidden.data_ptr(), int(nodes), act_lens.data_ptr(),
                label_lens.data_ptr(), B, T, U1, H, _DTYPES[enc_proj.dtype], torch.cuda.current_stream(dev).cuda_stream),
                "rnnt_joint_hidden_forward_compact")
        ctx.save_for_backward(hidden, act_lens, label_lens)
        ctx.shape = (B, T, U1, H)
        return hidden

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_hidden):
        hidden, act_lens, label_lens = ctx.saved_tensors
        B, T, U1, H = ctx.shape
        dev = hidden.device
        grad_hidden = grad_hidden.to(hidden.dtype).contiguous()
        with torch.cuda.device(dev):
            grad_enc = torch.empty((B, T, H), dtype=hidden.dtype, device=dev)
            grad_dec = torch.empty((B, U1, H), dtype=hidden.dtype, device=dev)
            ws = torch.empty(_workspace_bytes(B, T, U1, H), dtype=torch.uint8, device=dev)
            _lib.check(_lib.load().rnnt_joint_hidden_backward_compact(
                hidden.data_ptr(), grad_hidden.data_ptr(), hidden.shape[0], grad_enc.data_ptr(), grad_dec.data_ptr(),
                act_lens.data_ptr(), label_lens.data_ptr(), B, T, U1, H, _DTYPES[hidden.dtype], ws.data_ptr(),
                torch.cuda.current_stream(dev).cuda_stream), "rnnt_joint_hidden_backward_compact")
        return grad_enc, grad_dec, None, None, None


class _Projection(torch.autograd.Function):
    """``F.linear(x, weight, bias)`` over the rows of ``x`` ``[n, H]`` (fp32), with the weight gradient formed as a
    split-K product.  ``grad_weight = g^T x`` contracts over n = 10^5 .. 10^6 rows into a ``[V, H]`` result of a few
    dozen 256 x 256 tiles: as ONE library GEMM it runs on that many workgroups of a 256-CU chip (V = 4232, H = 512,
    n = 65 773: 4.45 ms = 64 TFLOP/s).  Cut into 32 row ranges as a batched GEMM whose partial products are added in
    range order it fills the chip: 2.09 ms there, 8.6 -> 6.4 ms at n = 197 944, 18.9 -> 16.9 ms at n = 524 800
    (profiles/r03_tuning.md §7).  Same sums in another association (fp32 throughout; deterministic)."""

    RANGES = 32
    MIN_ROWS = 8192

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.mm(g, weight)
        if ctx.needs_input_grad[1]:
            n, S = g.shape[0], _Projection.RANGES
            m = n // S
            if n >= _Projection.MIN_ROWS:
                gw = torch.bmm(g[:m * S].view(S, m, -1).transpose(1, 2), x[:m * S].view(S, m, -1)).sum(0)
                if m * S < n:
                    gw = gw + torch.mm(g[m * S:].t(), x[m * S:])
            else:
                gw = torch.mm(g.t(), x)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(0)
        return gx, gw, gb


def _project(x, linear):
    """``linear(x)`` for ``x`` ``[..., H]``; fp32 on the device takes :class:`_Projection`."""
    if (x.dtype is torch.float32 and x.is_cuda and linear.weight.dtype is torch.float32
            and not torch.is_autocast_enabled("cuda")):  # (under autocast the library's Linear casts its operands)
        return _Projection.apply(x.reshape(-1, x.shape[-1]), linear.weight, linear.bias).view(*x.shape[:-1], -1)
    return linear(x)


def joint_hidden(enc_proj: torch.Tensor, dec_proj: torch.Tensor, act_lens=None, label_lens=None, nodes=None) -> torch.Tensor:
    """``tanh(enc_proj[:, :, None, :] + dec_proj[:, None, :, :])`` → ``[B, T, U+1, H]`` on the MI355X.

    ``enc_proj`` ``[B, T, H]``, ``dec_proj`` ``[B, U+1, H]``, same dtype (fp32, bf16 or fp16) and device.
    With ``act_lens`` / ``label_lens`` (int32 ``[B]`` on the device): the ragged form for a consumer that ignores the
    rows outside the utterances' lattices — only rows ``t < act_lens[b]``, ``u <= label_lens[b]`` are written (the
    others are UNINITIALISED), and backward reads ``hidden`` and its gradient only there.
    With ``nodes`` as well (the number of lattice nodes, ``rnnt_amd.count_lattice_nodes``): the result is ``[nodes, H]``
    in the COMPACTED node order — row ``k`` is node ``k`` of ``rnnt_amd.compact_rowmap``, where a projection over the
    nodes and ``rnnt_amd.rnnt_loss_compact`` read it; no ``[B, T, U+1, H]`` tensor exists in either direction.
    """
    if enc_proj.dim() != 3 or dec_proj.dim() != 3:
        raise ValueError("enc_proj must be [B,T,H] and dec_proj [B,U+1,H]")
    if enc_proj.shape[0] != dec_proj.shape[0] or enc_proj.shape[2] != dec_proj.shape[2]:
        raise ValueError("enc_proj and dec_proj must agree in batch and hidden size")
    if enc_proj.dtype not in _DTYPES or dec_proj.dtype is not enc_proj.dtype:
        raise TypeError("enc_proj and dec_proj must both be torch.float32, torch.bfloat16 or torch.float16")
    if min(enc_proj.shape) < 1 or min(dec_proj.shape) < 1:
        raise ValueError("enc_proj and dec_proj must not be empty")
    if not enc_proj.is_cuda or dec_proj.device != enc_proj.device:
        raise RuntimeError("rnnt_amd.joint_hidden computes on MI355X (ROCm device tensors) only; there is no CPU "
                           "path or PyTorch fallback in this package")
    if (act_lens is None) != (label_lens is None):
        raise ValueError("pass both act_lens and label_lens, or neither")
    if act_lens is not None:
        for name, t in (("act_lens", act_lens), ("label_lens", label_lens)):
            if t.dtype is not torch.int32 or t.device != enc_proj.device or t.shape != (enc_proj.shape[0],) \
                    or not t.is_contiguous():
                raise ValueError(f"{name} must be a contiguous int32 [B] tensor on enc_proj's device")
    if nodes is not None:
        if act_lens is None or int(nodes) < 1:
            raise ValueError("nodes= needs the two length tensors and at least one node")
        return _JointHiddenCompact.apply(enc_proj, dec_proj, act_lens, label_lens, int(nodes))
    return _JointHidden.apply(enc_proj, dec_proj, act_lens, label_lens)


# JointNet.transducer_loss, lengths known on the host: a batch with fewer lattice nodes than this share of the rows of
# its [B, T, U+1] tensor takes the compacted route (hidden layer, projection GEMMs and the loss on one row per node).
COMPACT_BELOW = 0.9


class JointNet(nn.Module):
    """Drop-in for the reference's ``JointNet`` (model.py:9-63); see the module docstring."""

    def __init__(self, input_size, inner_dim, vocab_size, joint="concat"):
        super().__init__()
        self.joint = joint
        self.mlp = nn.Sequential(
            nn.Linear(input_size, inner_dim, bias=True),
            nn.Tanh(),
            nn.Linear(inner_dim, vocab_size, bias=True),
        )

    def hidden(self, enc_state, dec_state, act_lens=None, label_lens=None, nodes=None):
        """``[B, T, U+1, inner_dim]``: what the reference's ``mlp[:2]`` (Linear + Tanh) returns on the repeated and
        concatenated (or summed) states (model.py:46-58), from ``enc_state`` ``[B, T, He]`` and ``dec_state``
        ``[B, U+1, Hd]`` without building the ``[B, T, U+1, He+Hd]`` tensor.  With the two length tensors: the ragged
        form of :func:`joint_hidden` (rows outside the lattices uninitialised); with ``nodes`` too: ``[nodes, inner_dim]``
        in the compacted node order."""
        first = self.mlp[0]
        if self.joint == "concat":
            he = enc_state.size(-1)
            if he + dec_state.size(-1) != first.in_features:
                raise RuntimeError("enc/dec state sizes do not add up to the joint's input_size")
            w_enc, w_dec = first.weight[:, :he], first.weight[:, he:]
        else:
            w_enc = w_dec = first.weight
        enc_proj = F.linear(enc_state, w_enc, first.bias)      # [B, T, H]
        dec_proj = F.linear(dec_state, w_dec)                  # [B, U+1, H]
        return joint_hidden(enc_proj, dec_proj, act_lens, label_lens, nodes)

    def transducer_loss(self, enc_state, dec_state, labels, act_lens, label_lens, blank=0, reduction="mean"):
        """The reference's two lines ``logits = self.joint(enc_state, dec_state)`` and
        ``self.transducer_loss(logits, tokens.int(), output_length, tokens_length.int())`` (model.py:115-116) as one
        call.  16-bit states (bf16/fp16 training): the output projection and the loss run fused
        (``rnnt_amd.FusedJointLoss``: the logits are written once and never read by a softmax pass, their gradient is
        never written), over the lattice nodes of the batch only (the compacted node order); fp32 states: the
        projection as a library GEMM and ``rnnt_amd.rnnt_loss`` on its logits.

        ``act_lens`` / ``label_lens`` may be CPU tensors — where the reference has them when it makes this call
        (``enc_output_lengths``, ``tokens_length``: model.py:104-111).  The lattice nodes are then counted on the host
        (no device round trip), the lengths go to the device by one asynchronous copy, buffers and launch grids are cut
        for the nodes, and the fp32 route too skips everything outside the lattices: hidden layer, projection GEMMs and
        the loss's three passes run on one row per node (``rnnt_amd.rnnt_loss_compact``) — the same costs bit for bit.
        With device tensors nothing is read back: the fused route still skips the padding (the nodes are numbered on the
        device), the fp32 route works on the padded tensor.

        Under ``torch.autocast("cuda", dtype=torch.bfloat16 / torch.float16)`` fp32 states and parameters take the
        16-bit fused route (states and projection weight are cast; gradients arrive in fp32)."""
        if enc_state.dim() != 3 or dec_state.dim() != 3:
            raise ValueError("transducer_loss takes enc_state [B, T, He] and dec_state [B, U+1, Hd]")
        if self.joint not in ("sum", "concat"):
            raise NotImplementedError
        out = self.mlp[2]
        if enc_state.is_cuda and enc_state.dtype is torch.float32 and torch.is_autocast_enabled("cuda"):
            # fp32 parameters and states under torch.autocast("cuda", dtype=bf16 / fp16): as a Linear + loss would run
            # there, the states and (below) the projection's weight go to the 16-bit type and the fused route is taken;
            # gradients come back in fp32 through the casts
            amp = torch.get_autocast_dtype("cuda")
            if amp in (torch.bfloat16, torch.float16):
                enc_state, dec_state = enc_state.to(amp), dec_state.to(amp)
        # Lengths still on the host (the reference's are: model.py:104-111 moves them to the device for the loss): the
        # number of lattice nodes is counted there, for free, and the fused kernels are cut for the nodes of a ragged
        # batch instead of for the padded tensor; then the lengths go to the device as the reference's do.
        valid_nodes = None
        if not act_lens.is_cuda and not label_lens.is_cuda:
            from .fused import count_lattice_nodes

            valid_nodes = count_lattice_nodes(act_lens, label_lens, enc_state.size(1), dec_state.size(1))
        if valid_nodes is not None and enc_state.is_cuda:
            # one asynchronous copy from pinned memory: `lengths.cuda()` from pageable memory (model.py:109) makes the host
            # wait for everything queued on the device before it, every step
            both = torch.stack((act_lens.to(torch.int32), label_lens.to(torch.int32))).pin_memory()
            both = both.to(enc_state.device, non_blocking=True)
            act_lens, label_lens = both[0], both[1]
        else:
            act_lens = act_lens.to(device=enc_state.device, dtype=torch.int32)
            label_lens = label_lens.to(device=enc_state.device, dtype=torch.int32)
        if enc_state.dtype in (torch.bfloat16, torch.float16):
            from .fused import SUPPORTED_HIDDEN, fused_joint_loss, fused_joint_loss_skips_padding

            if out.in_features in SUPPORTED_HIDDEN and out.out_features >= 64 and out.bias is not None:
                shape = torch.empty((enc_state.size(0), enc_state.size(1), dec_state.size(1), out.in_features),
                                    dtype=enc_state.dtype, device="meta")
                # the fused kernels walk the nodes inside the lattices only: the hidden layer then computes, and
                # differentiates, only those rows too
                ragged = fused_joint_loss_skips_padding(shape, out.weight)
                hidden = self.hidden(enc_state, dec_state, *((act_lens, label_lens) if ragged else ()))
                # (an fp32 weight under autocast is cast to the states' type inside)
                return fused_joint_loss(hidden, out.weight, out.bias, labels, act_lens, label_lens, blank, reduction,
                                        valid_nodes=valid_nodes, zero_padding_grad=not ragged)
        B, T, U1 = enc_state.size(0), enc_state.size(1), dec_state.size(1)
        if valid_nodes is not None and 0 < valid_nodes <= COMPACT_BELOW * B * T * U1 and B * T * U1 < 2 ** 31:
            # fp32 states (the reference's own precision) and the shapes the fused kernels leave to the library: the
            # projection and the loss's three passes run over the lattice nodes only (rnnt_amd/compact.py) — the two
            # fp32 GEMMs of the projection's forward and backward are most of this step, and they shrink with the nodes
            from .compact import rnnt_loss_compact

            nodes = self.hidden(enc_state, dec_state, act_lens, label_lens, valid_nodes)   # [n, H]: one row per node
            return rnnt_loss_compact(_project(nodes, out), labels, act_lens, label_lens, T, blank=blank,
                                     reduction=reduction, exact_rows=True)
        hidden = self.hidden(enc_state, dec_state)
        from .loss import rnnt_loss

        return rnnt_loss(_project(hidden, out), labels, act_lens, label_lens, blank=blank, reduction=reduction)

    def forward(self, enc_state, dec_state, softmax=False):
        if enc_state.dim() != dec_state.dim():
            raise ValueError("input_TN and input_PN must be have same size."
                             "3 for training, or 1,2 for evaluation")
        if self.joint not in ("sum", "concat"):
            raise NotImplementedError
        if enc_state.dim() == 4 and enc_state.size(2) == 1 and dec_state.size(1) == 1:
            # already unsqueezed by the caller ([B,T,1,He] / [B,1,U+1,Hd]), model.py:46-50
            enc_state, dec_state = enc_state.squeeze(2), dec_state.squeeze(1)
        if enc_state.dim() == 3:
            joint = _project(self.hidden(enc_state, dec_state), self.mlp[2])     # [B, T, U+1, V]
        else:
            # evaluation: one frame against one decoder state ([B *] H), model.py:51-54 — a few hundred
            # numbers; greedy/beam decoding is outside this package's scope
            if self.joint == "sum":
                concat_state = enc_state + dec_state
            else:
                concat_state = torch.cat((enc_state, dec_state), dim=-1)
            joint = self.mlp(concat_state)
        if softmax:
            # the reference calls self.softmax here, an attribute its JointNet never defines
            # (model.py:60-61): the call site raises AttributeError there, and so it does here
            joint = self.softmax(joint)
        return joint
