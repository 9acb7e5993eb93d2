"""Restatement of the reference's greedy decoding loop — TEST INFRASTRUCTURE ONLY.

Follows ``decode`` inside ``Transducer.recognize`` (/root/reference/src/rnnt/model.py:145-165) line by
line with stock torch ops: start the decoder on token 0 (:141,148); per frame evaluate the joint on
the 1-D states (:151), softmax, argmax, ``.item()`` (:152-154); on a non-blank prediction append it
and step the decoder (:156-163).  ``joint_fn(enc_vec, dec_vec) -> logits[V]`` is the reference's
``JointNet`` evaluation path (``oracle.joint_oracle`` restates its layers).  PARITY UNPINNED: the
reference holds no decoding fixtures.  Also returns the smallest top-2 probability margin met, so a
test can tell a genuine disagreement from an argmax tie at rounding level.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


@torch.no_grad()
def greedy_decode_reference(enc_state, length, decoder, joint_fn):
    token_list = []
    margin = float("inf")
    zero_token = torch.zeros((1, 1), dtype=torch.long, device=enc_state.device)
    dec_state, hidden = decoder(zero_token)
    for t in range(int(length)):
        logits = joint_fn(enc_state[t].view(-1), dec_state.view(-1))
        out = F.softmax(logits, dim=0).detach()
        top2 = torch.topk(out, 2).values
        margin = min(margin, float(top2[0] - top2[1]))
        pred = int(torch.argmax(out, dim=0).item())
        if pred != 0:
            token_list.append(pred)
            token = torch.full((1, 1), pred, dtype=torch.long, device=enc_state.device)
            dec_state, hidden = decoder(token, hidden=hidden)
    return token_list, margin
