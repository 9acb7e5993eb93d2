#!/usr/bin/env python3
"""Joint network + loss, forward + backward, at config 2's shape (B32 T400 U40, He = Hd = 320, inner 512, V = 4232):
the reference's formulation in stock torch (repeat / cat / Linear / Tanh / Linear, model.py:46-59) + RNNTLoss, against
rnnt_amd.JointNet + RNNTLoss (two calls) and rnnt_amd.JointNet.transducer_loss (one call).  One JSON line.

    python tools/bench_joint_loss.py [--dtype bf16|f16|f32] [--shape B,T,U,V] [--reps 9] [--out file.json] [--ragged SEED]
                                       [--autocast bf16|f16]

--ragged: utterance lengths as in BASELINE config 3 scaled to the shape (T_b uniform in [T/8, T], U_b in [U/6, U], one
utterance at (T, U)), kept on the HOST as the reference has them when it calls the loss (model.py:104-111): the one-call
path counts the lattice nodes there and its fused kernels walk only those (the compacted node order).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def median_ms(fn, reps):
    for _ in range(3):
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
    ap.add_argument("--shape", default="32,400,40,4232")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--reps", type=int, default=9)
    ap.add_argument("--out", default=None)
    ap.add_argument("--ragged", type=int, default=None, metavar="SEED")
    ap.add_argument("--only-one-call", action="store_true", help="time JointNet.transducer_loss only (for profiler runs)")
    ap.add_argument("--autocast", default=None, choices=["bf16", "f16"],
                    help="run every formulation under torch.autocast of this dtype (use with --dtype f32: fp32 model)")
    args = ap.parse_args()
    import rnnt_amd

    B, T, U, V = (int(x) for x in args.shape.split(","))
    He = Hd = 320
    H = 512
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(7)
    net = rnnt_amd.JointNet(He + Hd, H, V).to(dev).to(dt)
    enc = torch.randn(B, T, He, device=dev, generator=g).to(dt).requires_grad_(True)
    dec = torch.randn(B, U + 1, Hd, device=dev, generator=g).to(dt).requires_grad_(True)
    labels = torch.randint(1, V, (B, U), device=dev, dtype=torch.int32, generator=g)
    al_host = torch.full((B,), T, dtype=torch.int32)
    ll_host = torch.full((B,), U, dtype=torch.int32)
    if args.ragged is not None:
        hg = torch.Generator().manual_seed(args.ragged)
        al_host = torch.randint(max(1, T // 8), T + 1, (B,), generator=hg, dtype=torch.int32)
        ll_host = torch.randint(U // 6, U + 1, (B,), generator=hg, dtype=torch.int32)
        al_host[0], ll_host[0] = T, U
    al, ll = al_host.to(dev), ll_host.to(dev)
    nodes = rnnt_amd.count_lattice_nodes(al_host, ll_host)
    crit = rnnt_amd.RNNTLoss(check_lengths=False)

    import contextlib
    amp = {"bf16": torch.bfloat16, "f16": torch.float16, None: None}[args.autocast]

    def cast():
        return torch.autocast("cuda", dtype=amp) if amp is not None else contextlib.nullcontext()

    def clear():
        enc.grad = dec.grad = None
        net.zero_grad(set_to_none=True)

    def reference_formulation():  # model.py:46-59 in stock torch, then the loss
        clear()
        with cast():
            e = enc.unsqueeze(2).repeat(1, 1, U + 1, 1)
            d = dec.unsqueeze(1).repeat(1, T, 1, 1)
            logits = net.mlp(torch.cat((e, d), dim=-1))
            loss = crit(logits, labels, al, ll)
        loss.backward()
        return loss

    def two_calls():
        clear()
        with cast():
            loss = crit(net(enc, dec), labels, al, ll)
        loss.backward()
        return loss

    def one_call():
        clear()
        with cast():
            loss = net.transducer_loss(enc, dec, labels, al_host, ll_host)  # (lengths on the host: model.py:104-111)
        loss.backward()
        return loss

    if args.only_one_call:
        print(json.dumps({"lengths": "uniform" if args.ragged is None else f"ragged (seed {args.ragged})", "nodes": nodes,
                          "JointNet.transducer_loss_ms": round(median_ms(one_call, args.reps), 3)}))
        return
    res = {"shape": dict(B=B, T=T, U=U, V=V, He=He, Hd=Hd, H=H), "dtype": args.dtype, "autocast": args.autocast,
           "lengths": "uniform" if args.ragged is None else
                      f"ragged (seed {args.ragged}): {nodes} of {B * T * (U + 1)} rows are lattice nodes",
           "loss": {"reference formulation": float(reference_formulation()), "JointNet + RNNTLoss": float(two_calls()),
                    "JointNet.transducer_loss": float(one_call())},
           "forward_backward_ms": {
               "reference formulation in stock torch (repeat, cat, Linear, Tanh, Linear) + rnnt_amd.RNNTLoss":
                   round(median_ms(reference_formulation, args.reps), 3),
               "rnnt_amd.JointNet + rnnt_amd.RNNTLoss": round(median_ms(two_calls, args.reps), 3),
               "rnnt_amd.JointNet.transducer_loss": round(median_ms(one_call, args.reps), 3)},
           "peak_memory_GB": {}}
    for name, fn in (("reference formulation", reference_formulation), ("JointNet + RNNTLoss", two_calls),
                     ("JointNet.transducer_loss", one_call)):
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        fn()
        torch.cuda.synchronize()
        res["peak_memory_GB"][name] = round(torch.cuda.max_memory_allocated() / 1e9, 2)
    line = json.dumps(res)
    print(line)
    if args.out:
        with open(args.out, "w") as f:
            f.write(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
