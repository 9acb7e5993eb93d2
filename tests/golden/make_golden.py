#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the float64 oracle (run from the repo root).

The reference (jhvmhg/rnnt) ships no tests, fixtures or golden vectors for its RNNT loss and the
package it imports (warprnnt_pytorch) is neither vendored nor importable here, so these vectors
come from the build's own float64 restatement (oracle/rnnt_oracle.c, fp64 instantiation), which
tests/test_oracle.py cross-checks against autograd through a float64 DP and against brute-force
path enumeration.  PARITY UNPINNED with respect to the real warp-transducer; the one upstream
anchors are KAT-1 and KAT-2 (kat1.npz, kat2.npz), recalled from the upstream package's public unit test.

Each file: acts f32 [B,T,U+1,V], labels i32 [B,U], act_lens i32 [B], label_lens i32 [B],
costs f64 [B] (-log P per utterance), grads f64 [B,T,U+1,V] (d costs[b] / d acts[b]).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from cases import KAT1, KAT2, make_case  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: make_case kwargs
    "tiny_T1_U0": dict(B=2, T=1, U=0, V=5, seed=101, ragged=False),
    "tiny_T1": dict(B=2, T=1, U=3, V=6, seed=102, ragged=True),
    "tiny_U0": dict(B=3, T=6, U=0, V=7, seed=103, ragged=True),
    "ragged_small": dict(B=4, T=9, U=5, V=11, seed=104, ragged=True, sigma=2.0),
    "ragged_sigma5": dict(B=3, T=12, U=7, V=33, seed=105, ragged=True, sigma=5.0),
    "peaked": dict(B=2, T=16, U=6, V=40, seed=106, ragged=True, peaked=True),
    "wide_lattice_U70": dict(B=2, T=10, U=70, V=16, seed=107, ragged=True),
    "blank7": dict(B=2, T=8, U=4, V=12, seed=108, ragged=True, blank=7),
    "vocab_unaligned_4234": dict(B=1, T=3, U=2, V=4234, seed=109, ragged=False),
}


def main():
    for name, kw in CASES.items():
        acts, labels, al, ll = make_case(**kw)
        blank = kw.get("blank", 0)
        costs, grads, _ = oracle.rnnt_loss_c(acts, labels, al, ll, blank=blank, dtype=np.float64)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), acts=acts, labels=labels, act_lens=al,
                            label_lens=ll, costs=costs, grads=grads, blank=np.int32(blank))
        print(name, acts.shape, costs)
    np.savez_compressed(os.path.join(OUT, "kat1.npz"), acts=KAT1["acts64"], labels=KAT1["labels"],
                        act_lens=KAT1["act_lens"], label_lens=KAT1["label_lens"],
                        costs=np.array([KAT1["cost"]]), grads=KAT1["grads"], blank=np.int32(0))
    # KAT-2: inputs and costs are upstream's; the full gradient is the float64 oracle's (only its
    # first node is recalled from upstream, asserted in tests/test_oracle.py)
    c2, g2, _ = oracle.rnnt_loss_c(KAT2["acts64"], KAT2["labels"], KAT2["act_lens"], KAT2["label_lens"],
                                   dtype=np.float64)
    assert np.abs(c2 - KAT2["costs"]).max() < 1e-12
    np.savez_compressed(os.path.join(OUT, "kat2.npz"), acts=KAT2["acts64"], labels=KAT2["labels"],
                        act_lens=KAT2["act_lens"], label_lens=KAT2["label_lens"],
                        costs=KAT2["costs"], grads=g2, blank=np.int32(0))


if __name__ == "__main__":
    main()
