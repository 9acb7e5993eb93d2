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

from typing import Callable, List

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
