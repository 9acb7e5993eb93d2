#!/usr/bin/env python3
"""Host cost of one RNN-T loss forward+backward through each binding.

    python tools/bench_host_overhead.py [--steps N] [--out file.json]

For B=1 (one utterance of BASELINE config 1's shape: T200 U20 V4232 fp32) and BASELINE config 1 (B8), for the
compiled binding (torch.ops.rnnt_amd.rnnt_loss, the default) and the ctypes binding, with and without the host-side
length check, in a loop of N back-to-back steps with no synchronisation inside:

  host_us   host time to enqueue one forward+backward (perf_counter around the two calls)
  wall_us   wall time per step of the whole loop, final synchronize included
  device_us sum of the library's three stage intervals K1 | K2+reduce | K3 (hipEvents recorded by the library on
            the launch stream around its own launches), median over the steps
  gap_us    wall_us - device_us: what the host adds to a step

When host_us < device_us the loop is bound by the device and the gap is what the launches leave between kernels.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(binding, check_lengths, B, T, U, V, steps):
    import rnnt_amd
    from rnnt_amd import _lib, loss as loss_mod

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(7)
    acts = torch.randn(B, T, U + 1, V, device=dev, generator=g).requires_grad_(True)
    labels = torch.randint(1, V, (B, U), device=dev, dtype=torch.int32, generator=g)
    al = torch.full((B,), T, device=dev, dtype=torch.int32)
    ll = torch.full((B,), U, device=dev, dtype=torch.int32)
    crit = rnnt_amd.RNNTLoss(check_lengths=check_lengths, binding=binding)

    def step():
        acts.grad = None
        crit(acts, labels, al, ll).backward()

    for _ in range(20):
        step()
    torch.cuda.synchronize()
    # pass 1: host and wall time, nothing else enqueued
    host = []
    t_begin = time.perf_counter()
    for _ in range(steps):
        t0 = time.perf_counter()
        step()
        host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t_begin) / steps
    # pass 2: device time per step from the library's own stage events (one step in flight at a time)
    fwd, bwd = _lib.StageEvents(3), _lib.StageEvents(2)
    loss_mod.set_stage_events(fwd, bwd)
    devs = []
    try:
        for _ in range(min(steps, 100)):
            step()
            torch.cuda.synchronize()
            devs.append(fwd.elapsed_ms(0, 1) + fwd.elapsed_ms(1, 2) + bwd.elapsed_ms(0, 1))
    finally:
        loss_mod.set_stage_events(None, None)
        fwd.close()
        bwd.close()
    device_us = statistics.median(devs) * 1e3
    return {"binding": binding, "check_lengths": check_lengths, "shape": dict(B=B, T=T, U=U, V=V),
            "host_us": round(statistics.median(host) * 1e6, 1), "wall_us": round(wall * 1e6, 1),
            "device_us": round(device_us, 1), "gap_us": round(wall * 1e6 - device_us, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    rows = []
    for B in (1, 8):
        for check in (False, True):
            for binding in ("ctypes", "torch"):
                rows.append(measure(binding, check, B, 200, 20, 4232, args.steps))
                print(json.dumps(rows[-1]), flush=True)
    res = {"device": torch.cuda.get_device_name(0), "steps": args.steps, "rows": rows}
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(res, fh, indent=1)
            fh.write("\n")


if __name__ == "__main__":
    main()
