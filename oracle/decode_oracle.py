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


@torch.no_grad()
def greedy_search_batch_reference(encoder_output, decoder, joint_fn, blank_id=0):
    """``TransducerBeamSearcher.transducer_greedy_decode`` (/root/reference/src/rnnt/beam_searcher.py:65-127)
    restated step by step: BOS = blank for every utterance (:73-82); per frame the joint on
    ``[B,1,1,He]`` x ``[B,1,Hd]`` (:86-89), ``torch.max(log_softmax)`` (:91-93), per utterance
    ``.item()`` (:96-103), then one decoder call for the utterances that emitted with their hidden
    states gathered and scattered back (:104-120).  Returns the reference's 4-tuple (:122-127) and the
    smallest top-2 log-probability margin met.

    ``joint_fn(enc[B,He], dec[B,Hd]) -> logits[B,V]``.  As shipped the reference passes ``True`` as the
    joint's third argument and its own ``JointNet`` then raises (``src/rnnt/model.py:60-61`` reads
    ``self.softmax``, which is never set); the restatement takes what :92 applies anyway, a
    LogSoftmax over the vocabulary.  PARITY UNPINNED: no fixtures exist for this path.
    """
    B = encoder_output.size(0)
    hyp = {"prediction": [[] for _ in range(B)], "logp_scores": [0.0 for _ in range(B)]}
    margin = float("inf")
    input_decoder = torch.ones((B, 1), device=encoder_output.device, dtype=torch.int64) * blank_id
    out_PN, hidden = decoder(input_decoder)
    for t_step in range(encoder_output.size(1)):
        logits = joint_fn(encoder_output[:, t_step, :], out_PN[:, 0, :])
        log_probs = F.log_softmax(logits, dim=-1)
        logp_targets, positions = torch.max(log_probs, dim=1)
        top2 = torch.topk(log_probs, 2, dim=1).values
        margin = min(margin, float((top2[:, 0] - top2[:, 1]).min()))
        have_update_hyp = []
        for i in range(positions.size(0)):
            if positions[i].item() != blank_id:
                hyp["prediction"][i].append(positions[i].item())
                hyp["logp_scores"][i] += float(logp_targets[i])
                input_decoder[i][0] = positions[i]
                have_update_hyp.append(i)
        if len(have_update_hyp) > 0:
            selected_input = input_decoder[have_update_hyp, :]
            if isinstance(hidden, tuple):
                selected_hidden = (hidden[0][:, have_update_hyp, :].contiguous(),
                                   hidden[1][:, have_update_hyp, :].contiguous())
            else:
                selected_hidden = hidden[:, have_update_hyp, :].contiguous()
            selected_out, selected_hidden = decoder(selected_input, hidden=selected_hidden)
            out_PN[have_update_hyp] = selected_out
            if isinstance(hidden, tuple):
                hidden[0][:, have_update_hyp, :] = selected_hidden[0]
                hidden[1][:, have_update_hyp, :] = selected_hidden[1]
            else:
                hidden[:, have_update_hyp, :] = selected_hidden
    score = torch.Tensor(hyp["logp_scores"]).exp().mean()
    return (hyp["prediction"], score, None, None), margin
