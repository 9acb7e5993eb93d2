#!/usr/bin/env python3
"""Greedy decoding of one utterance at the reference's sizes (T=400 frames, He=Hd=320, H=512, V=4232,
1-layer LSTM decoder of 512 units as config/aishell.yaml): rnnt_amd.greedy_decode (windowed joint)
against the reference's frame-by-frame loop (model.py:145-165, restated below)
with the same modules on the same MI355X, and the batched search rnnt_amd.greedy_search_batch against the
reference's batched loop (beam_searcher.py:65-127, restated below).  Prints one JSON line."""
import json
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F  # noqa: E402

import rnnt_amd  # noqa: E402


@torch.no_grad()
def greedy_decode_reference(enc_state, length, decoder, joint_fn):
    """The reference's formulation (model.py:145-165): one joint/softmax/argmax/.item() per frame."""
    token_list = []
    dec_state, hidden = decoder(torch.zeros((1, 1), dtype=torch.long, device=enc_state.device))
    for t in range(int(length)):
        out = F.softmax(joint_fn(enc_state[t].view(-1), dec_state.view(-1)), dim=0)
        pred = int(torch.argmax(out, dim=0).item())
        if pred != 0:
            token_list.append(pred)
            dec_state, hidden = decoder(torch.full((1, 1), pred, dtype=torch.long, device=enc_state.device),
                                        hidden=hidden)
    return token_list


@torch.no_grad()
def greedy_search_batch_reference(encoder_output, decoder, joint_fn, blank_id=0):
    """The reference's batched formulation (beam_searcher.py:65-127): one joint + log_softmax + max per frame
    for the batch, `.item()` per utterance per frame, one decoder call per frame for those that emitted."""
    n_utt = encoder_output.size(0)
    tokens = [[] for _ in range(n_utt)]
    logp_sum = [0.0] * n_utt
    last_token = torch.full((n_utt, 1), blank_id, device=encoder_output.device, dtype=torch.int64)
    pred_out, (h, c) = decoder(last_token)
    for t in range(encoder_output.size(1)):
        best_logp, best = torch.max(F.log_softmax(joint_fn(encoder_output[:, t, :], pred_out[:, 0, :]), dim=-1), dim=1)
        stepped = []
        for b in range(n_utt):                      # one host read per utterance per frame, as the reference
            sym = best[b].item()
            if sym != blank_id:
                tokens[b].append(sym)
                logp_sum[b] += best_logp[b]
                last_token[b, 0] = sym
                stepped.append(b)
        if stepped:
            new_out, (nh, nc) = decoder(last_token[stepped, :],
                                        hidden=(h[:, stepped, :].contiguous(), c[:, stepped, :].contiguous()))
            pred_out[stepped] = new_out
            h[:, stepped, :] = nh
            c[:, stepped, :] = nc
    return tokens, torch.Tensor(logp_sum).exp().mean()


class Decoder(nn.Module):
    def __init__(self, hidden, vocab, out):
        super().__init__()
        self.embedding = nn.Embedding(vocab, hidden, padding_idx=0)
        self.lstm = nn.LSTM(hidden, hidden, 1, batch_first=True)
        self.output_proj = nn.Linear(hidden, out)

    def forward(self, inputs, length=None, hidden=None):
        o, hidden = self.lstm(self.embedding(inputs), hidden)
        return self.output_proj(o), hidden


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    T, He, Hd, H, V = 400, 320, 320, 512, 4232
    joint = rnnt_amd.JointNet(He + Hd, H, V).to(dev)
    with torch.no_grad():
        joint.mlp[2].bias[0] += float(sys.argv[1]) if len(sys.argv) > 1 else 4.0   # ~10 % of the frames emit
        joint.mlp[2].weight *= 4.0
    decoder = Decoder(512, V, Hd).to(dev)
    enc = torch.randn(T, He, device=dev)

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2] * 1e3, out

    t_ref, ref = timed(lambda: greedy_decode_reference(enc, T, decoder, lambda e, d: joint(e, d)))
    res = {"workload": f"greedy decode, one utterance, T={T} He={He} Hd={Hd} H={H} V={V}", "tokens": len(ref),
           "reference_loop_ms": round(t_ref, 3)}
    for w in (16, 32, 64):
        t, got = timed(lambda: rnnt_amd.greedy_decode(enc, T, decoder, joint, window=w))
        res[f"window{w}_ms"] = round(t, 3)
        res[f"window{w}_same_tokens"] = got == ref
    B = 16
    encb = torch.randn(B, T, He, device=dev)
    t_ref, (refb, _) = timed(lambda: greedy_search_batch_reference(encb, decoder, lambda e, d: joint(e, d)), reps=3)
    res["batch"] = {"workload": f"batched greedy search (beam_searcher.py:65-127), B={B}, same sizes",
                    "tokens": sum(len(p) for p in refb), "reference_loop_ms": round(t_ref, 3)}
    for w in (16, 32):
        t, got = timed(lambda: rnnt_amd.greedy_search_batch(encb, decoder, joint, window=w), reps=3)
        res["batch"][f"window{w}_ms"] = round(t, 3)
        res["batch"][f"window{w}_same_tokens"] = got[0] == refb
    print(json.dumps(res))


if __name__ == "__main__":
    main()
