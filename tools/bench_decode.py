#!/usr/bin/env python3
"""Greedy decoding of one utterance at the reference's sizes (T=400 frames, He=Hd=320, H=512, V=4232,
1-layer LSTM decoder of 512 units as config/aishell.yaml): rnnt_amd.greedy_decode (windowed joint)
against the reference's frame-by-frame loop (model.py:145-165, restated below)
with the same modules on the same MI355X.  Prints one JSON line."""
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
    print(json.dumps(res))


if __name__ == "__main__":
    main()
