"""Greedy transducer decoding with the joint evaluated over a window of frames.

The reference's ``Transducer.recognize`` (/root/reference/src/rnnt/model.py:145-165) walks the frames
one at a time: joint -> softmax -> argmax -> ``.item()`` per frame, and a decoder step whenever the
argmax is not blank.  The decoder state only changes at an emission, so all frames up to the next
emission see the SAME decoder state: this module evaluates the joint for a window of ``window``
frames in one go (one ``tanh(enc_proj[t] + dec_proj)`` kernel, one library GEMM with the output
projection, one ``rnnt_row_argmax`` launch), reads the window's argmaxes back once, and resumes
after the first non-blank frame.  Same tokens as the reference's loop (at most one symbol per
frame), with one host round trip per window or emission instead of one per frame.
"""
from __future__ import annotations

from typing import Callable, List, Tuple

import torch
import torch.nn.functional as F

from . import _lib
from .joint import JointNet, joint_hidden

_DTYPES = {torch.float32: _lib.DTYPE_F32, torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}


def row_argmax(x: torch.Tensor) -> torch.Tensor:
    """``int32[R]``: first index of the maximum of every row of ``x[R, V]`` (NaN counts as maximum)."""
    if x.dim() != 2 or x.shape[0] < 1 or x.shape[1] < 1:
        raise ValueError("x must be a non-empty [rows, V] tensor")
    if x.dtype not in _DTYPES:
        raise TypeError("x must be torch.float32, torch.bfloat16 or torch.float16")
    if not x.is_cuda:
        raise RuntimeError("rnnt_amd.row_argmax computes on MI355X (ROCm device tensors) only")
    x = x.contiguous()
    with torch.cuda.device(x.device):
        out = torch.empty(x.shape[0], dtype=torch.int32, device=x.device)
        _lib.check(_lib.load().rnnt_row_argmax(x.data_ptr(), x.shape[0], x.shape[1], _DTYPES[x.dtype], out.data_ptr(),
                                               torch.cuda.current_stream(x.device).cuda_stream), "rnnt_row_argmax")
    return out


def row_max_logp(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """``(int32[R], float32[R])``: argmax of every row of ``x[R, V]`` as :func:`row_argmax` gives it, and
    ``max(log_softmax(row))``, i.e. ``torch.max(F.log_softmax(x, -1), dim=1)`` in one launch."""
    if x.dim() != 2 or x.shape[0] < 1 or x.shape[1] < 1:
        raise ValueError("x must be a non-empty [rows, V] tensor")
    if x.dtype not in _DTYPES:
        raise TypeError("x must be torch.float32, torch.bfloat16 or torch.float16")
    if not x.is_cuda:
        raise RuntimeError("rnnt_amd.row_max_logp computes on MI355X (ROCm device tensors) only")
    x = x.contiguous()
    with torch.cuda.device(x.device):
        out = torch.empty(x.shape[0], dtype=torch.int32, device=x.device)
        logp = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().rnnt_row_max_logp(x.data_ptr(), x.shape[0], x.shape[1], _DTYPES[x.dtype],
                                                 out.data_ptr(), logp.data_ptr(),
                                                 torch.cuda.current_stream(x.device).cuda_stream), "rnnt_row_max_logp")
    return out, logp


def _joint_weights(joint: JointNet, he: int):
    first, out = joint.mlp[0], joint.mlp[2]
    if joint.joint == "concat":
        return first, out, first.weight[:, :he], first.weight[:, he:]
    return first, out, first.weight, first.weight


@torch.no_grad()
def greedy_search_batch(encoder_output: torch.Tensor, decoder: Callable, joint: JointNet, blank: int = 0,
                        window: int = 32):
    """The reference's batched greedy search, ``TransducerBeamSearcher.transducer_greedy_decode``
    (/root/reference/src/rnnt/beam_searcher.py:65-127), with one host round trip per window of frames.

    The reference walks the frames of the whole batch together and reads ``positions[i].item()`` for every
    utterance at every frame.  An utterance's decoder state only changes when it emits, so here every
    utterance keeps its own frame pointer: each iteration evaluates the joint for the next ``window`` frames
    of all unfinished utterances against their current decoder outputs (one ``tanh`` kernel, one library
    GEMM, one ``rnnt_row_max_logp`` launch), copies the ``[A, window]`` predictions and scores to the host
    once, advances each utterance to just after its first non-blank frame, and steps the decoder for those
    that emitted (one batched call, hidden states gathered and scattered as ``:107-120`` does).

    Args:
        encoder_output: ``[B, T, He]``; as in the reference all ``T`` frames of every utterance are decoded.
        decoder: called as the reference calls it: ``decoder(tokens[n,1])`` and
            ``decoder(tokens[n,1], hidden=hidden)`` -> ``(out[n,1,Hd], hidden)``, hidden a tensor or an
            ``(h, c)`` tuple shaped ``[layers, n, Hd]``.
    Returns:
        ``(predictions, mean(exp(score)), None, None)`` as the reference's 4-tuple (``:122-127``): token lists
        per utterance, and the batch mean of ``exp`` of each utterance's summed log-probabilities of its
        emitted symbols.
    """
    if encoder_output.dim() != 3:
        raise ValueError("encoder_output must be [B, T, He]")
    if window < 1:
        raise ValueError("window must be positive")
    B, T, he = encoder_output.shape
    dev = encoder_output.device
    first, out, w_enc, w_dec = _joint_weights(joint, he)
    enc_proj = F.linear(encoder_output, w_enc, first.bias)                      # [B, T, H], once

    predictions: List[List[int]] = [[] for _ in range(B)]
    scores = [0.0] * B
    input_decoder = torch.full((B, 1), blank, dtype=torch.long, device=dev)     # BOS = blank (:73-80)
    out_pn, hidden = decoder(input_decoder)                                     # [B, 1, Hd]
    dec_proj = F.linear(out_pn, w_dec)                                          # [B, 1, H]
    ptr = [0] * B
    offs = torch.arange(window, device=dev)
    while True:
        active = [b for b in range(B) if ptr[b] < T]
        if not active:
            break
        act = torch.tensor(active, device=dev)
        start = torch.tensor([ptr[b] for b in active], device=dev)
        frames = torch.clamp(start[:, None] + offs[None, :], max=T - 1)         # [A, W]; surplus frames ignored below
        win = enc_proj[act[:, None], frames]                                    # [A, W, H]
        h = joint_hidden(win.contiguous(), dec_proj[act].contiguous())          # [A, W, 1, H]
        pos, logp = row_max_logp(out(h.reshape(len(active) * window, -1)))
        host = torch.stack((pos, logp.view(torch.int32))).cpu()                 # the one host round trip
        pos_h = host[0].reshape(len(active), window).tolist()
        logp_h = host[1].view(torch.float32).reshape(len(active), window).tolist()
        emitted = []
        for a, b in enumerate(active):
            n = min(window, T - ptr[b])
            step = next((j for j in range(n) if pos_h[a][j] != blank), None)
            if step is None:
                ptr[b] += n
                continue
            predictions[b].append(pos_h[a][step])
            scores[b] += logp_h[a][step]
            ptr[b] += step + 1                                                  # one symbol per frame at most
            emitted.append(b)
        if emitted:
            sel = torch.tensor(emitted, device=dev)
            input_decoder[sel, 0] = torch.tensor([predictions[b][-1] for b in emitted], device=dev)
            if isinstance(hidden, tuple):
                sel_hidden = tuple(hx[:, sel, :].contiguous() for hx in hidden)
            else:
                sel_hidden = hidden[:, sel, :].contiguous()
            sel_out, sel_hidden = decoder(input_decoder[sel], hidden=sel_hidden)
            out_pn[sel] = sel_out
            if isinstance(hidden, tuple):
                for hx, new in zip(hidden, sel_hidden):
                    hx[:, sel, :] = new
            else:
                hidden[:, sel, :] = sel_hidden
            dec_proj[sel] = F.linear(sel_out, w_dec)
    return predictions, torch.tensor(scores).exp().mean(), None, None


@torch.no_grad()
def greedy_decode(enc_state: torch.Tensor, length: int, decoder: Callable, joint: JointNet, blank: int = 0,
                  window: int = 32) -> List[int]:
    """Tokens of one utterance, as ``decode(enc_state, lengths)`` of the reference (model.py:145-165).

    Args:
        enc_state: ``[T, He]`` encoder output of the utterance (device tensor).
        length: number of valid frames.
        decoder: the prediction network, called as the reference calls it: ``decoder(token[1,1])`` and
            ``decoder(token[1,1], hidden=hidden)`` -> ``(dec_state[1,1,Hd], hidden)``.
        joint: :class:`rnnt_amd.JointNet` (its two Linear layers are used directly).
        window: frames evaluated per launch.
    """
    if enc_state.dim() != 2:
        raise ValueError("enc_state must be [T, He]")
    if window < 1:
        raise ValueError("window must be positive")
    length = int(length)
    dev = enc_state.device
    first, out = joint.mlp[0], joint.mlp[2]
    if joint.joint == "concat":
        he = enc_state.size(-1)
        w_enc, w_dec = first.weight[:, :he], first.weight[:, he:]
    else:
        w_enc = w_dec = first.weight
    enc_proj = F.linear(enc_state[:length], w_enc, first.bias).unsqueeze(0)  # [1, T, H], once per utterance

    tokens: List[int] = []
    zero_token = torch.zeros((1, 1), dtype=torch.long, device=dev)
    dec_state, hidden = decoder(zero_token)
    dec_proj = F.linear(dec_state.reshape(1, 1, -1), w_dec)                  # [1, 1, H]
    t = 0
    while t < length:
        n = min(window, length - t)
        h = joint_hidden(enc_proj[:, t:t + n].contiguous(), dec_proj)        # [1, n, 1, H]
        pred = row_argmax(out(h.reshape(n, -1))).tolist()                    # the one host round trip
        step = next((i for i, p in enumerate(pred) if p != blank), None)
        if step is None:
            t += n
            continue
        tokens.append(pred[step])
        dec_state, hidden = decoder(torch.full((1, 1), pred[step], dtype=torch.long, device=dev), hidden=hidden)
        dec_proj = F.linear(dec_state.reshape(1, 1, -1), w_dec)
        t += step + 1                                                        # at most one symbol per frame
    return tokens
