"""Restatement of the reference's greedy decoding loop — TEST INFRASTRUCTURE ONLY.

Follows ``decode`` inside ``Transducer.recognize`` (/root/reference/src/rnnt/model.py:145-165) line by
line with stock torch ops: start the decoder on token 0 (:141,148); per frame evaluate the joint on
the 1-D states (:151), softmax, argmax, ``.item()`` (:152-154); on a non-blank prediction append it
and step the decoder (:156-163).  ``joint_fn(enc_vec, dec_vec) -> logits[V]`` is the reference's
``JointNet`` evaluation path (``oracle.joint_oracle`` restates its layers).  PINNED to the reference's own
code: tests/golden/make_ref_fixtures.py runs ``Transducer.recognize`` on a seeded tiny transducer in the
build container and writes its token lists, with the encoder states and the decoder's and joint's
parameters, to tests/golden/ref_recognize.npz; tests/test_joint_oracle.py holds this loop to them (the
batched search below stays unpinned).  Also returns the smallest top-2 probability margin met, so a
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
    n_utt, n_frames = encoder_output.size(0), encoder_output.size(1)
    tokens = [[] for _ in range(n_utt)]
    logp_sum = [0.0] * n_utt
    margin = float("inf")

    def take(state, rows):      # hidden state of the chosen utterances (LSTM pair or single tensor)
        if isinstance(state, tuple):
            return tuple(part[:, rows, :].contiguous() for part in state)
        return state[:, rows, :].contiguous()

    def put(state, rows, new):  # ... and written back in place
        if isinstance(state, tuple):
            for part, fresh in zip(state, new):
                part[:, rows, :] = fresh
        else:
            state[:, rows, :] = new

    last_token = torch.full((n_utt, 1), blank_id, device=encoder_output.device, dtype=torch.int64)  # :73-80
    pred_out, state = decoder(last_token)                                                          # :82
    for t in range(n_frames):                                                                      # :84
        log_probs = F.log_softmax(joint_fn(encoder_output[:, t, :], pred_out[:, 0, :]), dim=-1)    # :86-92
        best_logp, best = torch.max(log_probs, dim=1)                                              # :91-93
        two = torch.topk(log_probs, 2, dim=1).values
        margin = min(margin, float((two[:, 0] - two[:, 1]).min()))
        stepped = []
        for b in range(n_utt):                                                                     # :96-103
            sym = best[b].item()
            if sym != blank_id:
                tokens[b].append(sym)
                logp_sum[b] += float(best_logp[b])
                last_token[b, 0] = sym
                stepped.append(b)
        if stepped:                                                                                # :104-120
            new_out, new_state = decoder(last_token[stepped, :], hidden=take(state, stepped))
            pred_out[stepped] = new_out
            put(state, stepped, new_state)
    score = torch.Tensor(logp_sum).exp().mean()                                                    # :124
    return (tokens, score, None, None), margin
