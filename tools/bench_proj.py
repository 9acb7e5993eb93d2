#!/usr/bin/env python3
"""Time the joint's output projection Linear(H -> V) (/root/reference/src/rnnt/model.py:14-18,59) as plain
library GEMMs at a configuration's shape, beside the loss kernels that consume its result:

    forward   logits[L, V]  = hidden[L, H] @ W[V, H]^T + b          L = B*T*(U+1)
    dH        g_hidden[L,H] = g_logits[L, V] @ W[V, H]
    dW        g_W[V, H]     = g_logits[L, V]^T @ hidden[L, H]

in fp32 and bf16 (torch.matmul -> hipBLASLt / rocBLAS), then K1 / K2 / K3 of the loss on the same logits.
Writes one JSON object (profiles/r02_proj_gemm.json is a run of this at config 2).
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def time_ms(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="32,400,40,4232,512", help="B,T,U,V,H")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    B, T, U, V, H = (int(x) for x in args.shape.split(","))
    L = B * T * (U + 1)
    dev = torch.device("cuda:0")
    flop = 2.0 * L * H * V
    res = {"shape": dict(B=B, T=T, U=U, V=V, H=H, L=L), "flop_per_gemm": flop, "device": torch.cuda.get_device_name(0),
           "gemm": {}, "loss": {}}
    for name, dt in (("bf16", torch.bfloat16), ("f32", torch.float32)):
        torch.cuda.empty_cache()
        hidden = torch.randn(L, H, device=dev, dtype=dt)
        W = torch.randn(V, H, device=dev, dtype=dt) * 0.05
        bias = torch.zeros(V, device=dev, dtype=dt)
        logits = torch.empty(L, V, device=dev, dtype=dt)
        gh = torch.empty(L, H, device=dev, dtype=dt)
        gw = torch.empty(V, H, device=dev, dtype=dt)
        r = {}
        r["fwd_ms"] = time_ms(lambda: torch.addmm(bias, hidden, W.t(), out=logits), args.reps)
        g_logits = logits  # any [L, V] operand of the same dtype times the same
        r["dH_ms"] = time_ms(lambda: torch.mm(g_logits, W, out=gh), args.reps)
        r["dW_ms"] = time_ms(lambda: torch.mm(g_logits.t(), hidden, out=gw), args.reps)
        for k in ("fwd", "dH", "dW"):
            r[k + "_TFLOPs"] = round(flop / (r[k + "_ms"] * 1e-3) / 1e12, 1)
        r["three_gemms_ms"] = r["fwd_ms"] + r["dH_ms"] + r["dW_ms"]
        res["gemm"][name] = {k: round(v, 4) if isinstance(v, float) else v for k, v in r.items()}

        # the loss on the same logits
        import numpy as np

        import rnnt_amd
        from rnnt_amd import _lib, loss as loss_mod

        del hidden, gh, gw
        acts = logits.view(B, T, U + 1, V).requires_grad_(True)
        labels = torch.randint(1, V, (B, U), device=dev, dtype=torch.int32)
        al = torch.full((B,), T, device=dev, dtype=torch.int32)
        ll = torch.full((B,), U, device=dev, dtype=torch.int32)
        crit = rnnt_amd.RNNTLoss(check_lengths=False)
        k1, k2, k3 = [], [], []
        for i in range(args.reps + 2):
            f, b = _lib.StageEvents(3), _lib.StageEvents(2)
            loss_mod.set_stage_events(f, b)
            acts.grad = None
            crit(acts, labels, al, ll).backward()
            torch.cuda.synchronize()
            if i >= 2:
                k1.append(f.elapsed_ms(0, 1))
                k2.append(f.elapsed_ms(1, 2))
                k3.append(b.elapsed_ms(0, 1))
        loss_mod.set_stage_events(None, None)
        res["loss"][name] = {"k1_ms": round(float(np.median(k1)), 4), "k2_ms": round(float(np.median(k2)), 4),
                             "k3_ms": round(float(np.median(k3)), 4)}
        del acts, logits
    out = json.dumps(res, indent=1)
    print(out)
    if args.out:
        with open(args.out, "w") as f:
            f.write(out + "\n")


if __name__ == "__main__":
    main()
