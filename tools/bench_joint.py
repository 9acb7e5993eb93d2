#!/usr/bin/env python3
"""Times the joint hidden layer at BASELINE config 2 (B32 T400 U40, He=Hd=320, H=512) on one MI355X:
rnnt_amd.joint_hidden (two small GEMMs + one HIP kernel; backward one kernel + ordered reduction)
against the reference's formulation (repeat, cat, Linear(640->512), Tanh; model.py:46-59) run by
stock torch on the same device.  Prints one JSON line.  HBM algorithmic bytes: forward writes
B*T*(U+1)*H elements; backward reads hidden and grad_hidden once."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rnnt_amd  # noqa: E402


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--ragged", type=int, default=None, metavar="SEED",
                    help="lengths as BASELINE config 3 scaled to the shape; the kernels then skip the rows outside the lattices")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.float32 if args.dtype == "f32" else torch.bfloat16
    esz = 4 if args.dtype == "f32" else 2
    B, T, U1, He, Hd, H = 32, 400, 41, 320, 320, 512
    g = torch.Generator(device=dev).manual_seed(0)
    enc = torch.randn(B, T, He, device=dev, generator=g).to(dt).requires_grad_(True)
    dec = torch.randn(B, U1, Hd, device=dev, generator=g).to(dt).requires_grad_(True)
    lin = torch.nn.Linear(He + Hd, H).to(dev).to(dt)
    gh = torch.randn(B, T, U1, H, device=dev, generator=g).to(dt)
    n = B * T * U1 * H

    ep = torch.nn.functional.linear(enc, lin.weight[:, :He], lin.bias).detach().requires_grad_(True)
    dp = torch.nn.functional.linear(dec, lin.weight[:, He:]).detach().requires_grad_(True)
    state = {}
    lens = ()
    if args.ragged is not None:
        hg = torch.Generator().manual_seed(args.ragged)
        al = torch.randint(max(1, T // 8), T + 1, (B,), generator=hg, dtype=torch.int32)
        ll = torch.randint((U1 - 1) // 6, U1, (B,), generator=hg, dtype=torch.int32)
        al[0], ll[0] = T, U1 - 1
        n = rnnt_amd.count_lattice_nodes(al, ll) * H
        lens = (al.to(dev), ll.to(dev))

    def k_fwd():
        state["h"] = rnnt_amd.joint_hidden(ep, dp, *lens)

    def k_bwd():
        ep.grad = dp.grad = None
        state["h"].backward(gh, retain_graph=True)

    def ours_all():
        enc.grad = dec.grad = None
        lin.zero_grad(set_to_none=True)
        e = torch.nn.functional.linear(enc, lin.weight[:, :He], lin.bias)
        d = torch.nn.functional.linear(dec, lin.weight[:, He:])
        rnnt_amd.joint_hidden(e, d, *lens).backward(gh)

    def ref_all():
        enc.grad = dec.grad = None
        lin.zero_grad(set_to_none=True)
        e = enc.unsqueeze(2).repeat(1, 1, U1, 1)
        d = dec.unsqueeze(1).repeat(1, T, 1, 1)
        torch.tanh(lin(torch.cat((e, d), dim=-1))).backward(gh)

    t_fwd = timed(k_fwd, args.steps, args.warmup)
    t_bwd = timed(k_bwd, args.steps, args.warmup)
    t_ours = timed(ours_all, args.steps, args.warmup)
    t_ref = timed(ref_all, args.steps, args.warmup)
    out = {
        "workload": f"joint hidden layer B{B} T{T} U{U1 - 1} He{He} Hd{Hd} H{H} {args.dtype}"
                    + ("" if args.ragged is None else f", ragged (seed {args.ragged}): {n // H} of {B * T * U1} rows are nodes"),
        "joint_hidden_fwd_ms": round(t_fwd, 4),
        "fwd_GBs": round(n * esz / t_fwd / 1e6, 1),
        "joint_hidden_bwd_ms": round(t_bwd, 4),
        "bwd_GBs": round(2 * n * esz / t_bwd / 1e6, 1),
        "fwd_bwd_with_gemms_ms": round(t_ours, 4),
        "reference_formulation_torch_fwd_bwd_ms": round(t_ref, 4),
        "speedup": round(t_ref / t_ours, 2),
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
