#!/usr/bin/env python3
"""The loss's three passes on a ragged batch: logits of the padded [B, T, U+1, V] tensor (rnnt_amd.rnnt_loss, what the
reference's call site hands over) against logits of the lattice nodes only (rnnt_amd.rnnt_loss_compact).  One JSON line.

    python tools/bench_compact_loss.py [--shape B,T,U,V] [--dtype f32|bf16|f16] [--seed 3] [--reps 7] [--out file.json]

Lengths as BASELINE config 3: T_b uniform in [T/8, T], U_b in [U/6, U], one utterance at (T, U).
Algorithmic bytes: padded 2 N_valid s + N_full s (K3 writes zeros into the padding), compacted 3 N_valid s.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def median_ms(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="64,800,60,4232")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f16"])
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import rnnt_amd

    B, T, U, V = (int(x) for x in args.shape.split(","))
    dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
    dev = torch.device("cuda:0")
    hg = torch.Generator().manual_seed(args.seed)
    al_h = torch.randint(max(1, T // 8), T + 1, (B,), generator=hg, dtype=torch.int32)
    ll_h = torch.randint(U // 6, U + 1, (B,), generator=hg, dtype=torch.int32)
    al_h[0], ll_h[0] = T, U
    al, ll = al_h.to(dev), ll_h.to(dev)
    n = rnnt_amd.count_lattice_nodes(al_h, ll_h, T, U + 1)
    g = torch.Generator(device=dev).manual_seed(1)
    labels = torch.randint(1, V, (B, U), device=dev, dtype=torch.int32, generator=g)
    rowmap = rnnt_amd.compact_rowmap(al, ll, T, U + 1, n).long()
    nodes = torch.randn(n, V, device=dev, generator=g).to(dt).requires_grad_(True)
    go = torch.full((1,), 0.5, device=dev)
    res = {"shape": dict(B=B, T=T, U=U, V=V), "dtype": args.dtype, "rows": B * T * (U + 1), "nodes": n}
    s = nodes.element_size()

    def compact():
        nodes.grad = None
        rnnt_amd.rnnt_loss_compact(nodes, labels, al, ll, T, exact_rows=True).backward(go)

    res["compacted_ms"] = round(median_ms(compact, args.reps), 3)
    res["compacted_GBs"] = round(3.0 * n * V * s / res["compacted_ms"] / 1e6, 1)
    loss_c = float(rnnt_amd.rnnt_loss_compact(nodes.detach(), labels, al, ll, T))
    # the same logits in the padded tensor (the rows outside the lattices hold NaN: never read)
    padded = torch.full((B * T * (U + 1), V), float("nan"), dtype=dt, device=dev)
    padded[rowmap] = nodes.detach()
    padded = padded.view(B, T, U + 1, V).requires_grad_(True)

    def dense():
        padded.grad = None
        rnnt_amd.rnnt_loss(padded, labels, al, ll, check_lengths=False).backward(go)

    res["padded_ms"] = round(median_ms(dense, args.reps), 3)
    res["padded_GBs"] = round((2.0 * n + B * T * (U + 1)) * V * s / res["padded_ms"] / 1e6, 1)
    res["loss"] = {"compacted": loss_c, "padded": float(rnnt_amd.rnnt_loss(padded.detach(), labels, al, ll, check_lengths=False))}
    res["speedup"] = round(res["padded_ms"] / res["compacted_ms"], 2)
    line = json.dumps(res)
    print(line)
    if args.out:
        with open(args.out, "w") as f:
            f.write(json.dumps(res, indent=1) + "\n")


if __name__ == "__main__":
    main()
