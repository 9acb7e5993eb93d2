#!/usr/bin/env python3
"""Joint projection + RNN-T loss, fused against unfused, in one run.

    python tools/bench_fused.py [--shape B,T,U,V,H] [--dtype bf16|f16] [--reps N] [--out file.json]
                                [--ragged SEED] [--no-host-count]

--ragged: utterance lengths as in BASELINE config 3, scaled to the shape (T_b uniform in [T/8, T], U_b in [U/6, U],
one utterance at (T, U)); the fused path then walks the COMPACTED node order (only the nodes inside the lattices) and
is also timed over every row of the tensor ("fused, dense").  --no-host-count: the fused path is not told the number of
nodes (logits buffer and grids cut for the whole tensor; the surplus workgroups leave at once).

forward   unfused: library GEMM (torch.addmm -> hipBLASLt) writes the logits, rnnt_forward reads them (K1) + K2
          fused:   rnnt_joint_loss_forward (MFMA GEMM with K1 as its epilogue) + K2
fwd+bwd   both paths then run K3 and the two library GEMMs of the Linear's backward (dH, dW) + the bias sum.

Device time from hipEvents on the launch stream, median of --reps after warm-up; one JSON object.
profiles/r02_bench_fused.json is a run of this at BASELINE config 2's lattice (B32 T400 U40 V4232, H=512, bf16).
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def median_ms(fn, reps, warm=3):
    for _ in range(warm):
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
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="32,400,40,4232,512")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--reps", type=int, default=15)
    ap.add_argument("--out", default=None)
    ap.add_argument("--only-backward", action="store_true", help="stage times of rnnt_joint_loss_backward only")
    ap.add_argument("--only-forward", action="store_true", help="the fused forward only (tuning A/B runs)")
    ap.add_argument("--ragged", type=int, default=None, metavar="SEED", help="ragged lengths (config 3's distribution)")
    ap.add_argument("--no-host-count", action="store_true", help="do not pass the number of lattice nodes to the fused path")
    ap.add_argument("--dense", action="store_true", help="fused path over every row of the tensor (no compacted order)")
    args = ap.parse_args()
    import rnnt_amd

    B, T, U, V, H = (int(x) for x in args.shape.split(","))
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1234)
    hidden = torch.tanh(torch.randn(B, T, U + 1, H, device=dev, generator=g)).to(dt).requires_grad_(True)
    proj = torch.nn.Linear(H, V).to(dev).to(dt)
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
    valid_nodes = None if args.no_host_count else nodes
    crit = rnnt_amd.RNNTLoss(check_lengths=False)
    fused_mod = rnnt_amd.FusedJointLoss(min_nodes=0, compact_order=not args.dense)  # time the kernel itself, whatever the size
    fused = lambda *a: fused_mod(*a, valid_nodes=valid_nodes)
    fused_dense = rnnt_amd.FusedJointLoss(min_nodes=0, compact_order=False)
    fused_r2 = rnnt_amd.FusedJointLoss(min_nodes=0, fused_backward=False)  # round 2: K3 + two library GEMMs
    go = torch.full((1,), 0.5, device=dev)

    L = B * T * (U + 1)
    flop = 2.0 * L * H * V

    def clear():
        hidden.grad = None
        proj.weight.grad = None
        proj.bias.grad = None

    def unfused_fwd():
        with torch.no_grad():
            return crit(torch.nn.functional.linear(hidden, proj.weight, proj.bias), labels, al, ll)

    def fused_fwd():
        with torch.no_grad():
            return fused(hidden, proj, labels, al, ll)

    def unfused_step():
        clear()
        crit(torch.nn.functional.linear(hidden, proj.weight, proj.bias), labels, al, ll).backward(go)

    def fused_step():
        clear()
        fused(hidden, proj, labels, al, ll).backward(go)

    def fused_r2_step():
        clear()
        fused_r2(hidden, proj, labels, al, ll).backward(go)

    def fused_dense_step():
        clear()
        fused_dense(hidden, proj, labels, al, ll).backward(go)

    def backward_stages():
        """rnnt_joint_loss_backward alone, with the library's stage events: node coefficients, dH, dW + reduce."""
        import ctypes
        from rnnt_amd import _lib
        from rnnt_amd.loss import _DTYPES
        lib = _lib.load()
        U1 = U + 1
        with torch.no_grad():
            ld = (V + 127) // 128 * 128      # the row stride rnnt_amd.FusedJointLoss keeps the logits at
            cap = B * T * U1 if (args.dense or valid_nodes is None) else valid_nodes
            logits = torch.empty(cap, ld, dtype=dt, device=dev)
            ws = torch.empty(_lib.workspace_bytes(T, U1, B), dtype=torch.uint8, device=dev)
            costs = torch.empty(B, device=dev)
            b32 = proj.bias.float().contiguous()
            st = torch.cuda.current_stream().cuda_stream
            opts = _lib.RnntOptions(st, 0, T, U1, _DTYPES[dt], None)
            n = ctypes.c_size_t(0)
            if args.dense:
                _lib.check(lib.rnnt_joint_loss_forward_strided(hidden.data_ptr(), proj.weight.data_ptr(), b32.data_ptr(),
                                                               logits.data_ptr(), ld, labels.data_ptr(), al.data_ptr(),
                                                               ll.data_ptr(), H, V, B, costs.data_ptr(), None, 0,
                                                               ws.data_ptr(), opts), "fwd")
                _lib.check(lib.rnnt_joint_loss_backward_workspace_size(H, V, B, T, U1, ctypes.byref(n)), "size")
            else:
                _lib.check(lib.rnnt_joint_loss_compact_workspace_size(B, T, U1, ctypes.byref(n)), "size")
                cws = torch.empty(int(n.value), dtype=torch.uint8, device=dev)
                _lib.check(lib.rnnt_joint_loss_forward_compact(hidden.data_ptr(), proj.weight.data_ptr(), b32.data_ptr(),
                                                               logits.data_ptr(), ld, cap, labels.data_ptr(), al.data_ptr(),
                                                               ll.data_ptr(), H, V, B, costs.data_ptr(), None, 0,
                                                               ws.data_ptr(), cws.data_ptr(), opts), "fwd")
                _lib.check(lib.rnnt_joint_loss_backward_compact_workspace_size(H, V, cap, ctypes.byref(n)), "size")
            bws = torch.empty(int(n.value), dtype=torch.uint8, device=dev)
            dh, dw = torch.empty_like(hidden), torch.empty_like(proj.weight)
            db = torch.empty(V, device=dev)
            ev = _lib.StageEvents(4)
            opts = _lib.RnntOptions(st, 0, T, U1, _DTYPES[dt], ev.ptr())
            rows = []
            for _ in range(args.reps + 3):
                if args.dense:
                    _lib.check(lib.rnnt_joint_loss_backward_strided(
                        hidden.data_ptr(), proj.weight.data_ptr(), logits.data_ptr(), ld, labels.data_ptr(), al.data_ptr(),
                        ll.data_ptr(), H, V, B, go.data_ptr(), 0, ctypes.c_float(1.0 / B), dh.data_ptr(), dw.data_ptr(),
                        db.data_ptr(), ws.data_ptr(), bws.data_ptr(), opts), "bwd")
                else:
                    _lib.check(lib.rnnt_joint_loss_backward_compact(
                        hidden.data_ptr(), proj.weight.data_ptr(), logits.data_ptr(), ld, cap, labels.data_ptr(),
                        al.data_ptr(), ll.data_ptr(), H, V, B, go.data_ptr(), 0, ctypes.c_float(1.0 / B), dh.data_ptr(), 1,
                        dw.data_ptr(), db.data_ptr(), ws.data_ptr(), cws.data_ptr(), bws.data_ptr(), opts), "bwd")
                torch.cuda.synchronize()
                rows.append([ev.elapsed_ms(i, i + 1) for i in range(3)])
            rows = sorted(rows[3:], key=sum)
            med = rows[len(rows) // 2]
            ev.close()
        vflop = flop * nodes / L   # the products over the nodes inside the lattices
        return {"node_coefs": round(med[0], 4), "dH": round(med[1], 4), "dW+reduce": round(med[2], 4),
                "dH_TFLOPs": round(vflop / (med[1] * 1e-3) / 1e12, 1), "dW_TFLOPs": round(vflop / (med[2] * 1e-3) / 1e12, 1),
                "workspace_MB": round(int(n.value) / 1e6, 1)}

    from rnnt_amd.fused import fused_backward_supported
    if args.only_forward:
        ms = median_ms(fused_fwd, args.reps)
        print(json.dumps({"shape": [B, T, U, V, H], "dtype": args.dtype, "fused_forward_ms": round(ms, 4),
                          "TFLOPs": round(flop / (ms * 1e-3) / 1e12, 1)}))
        return
    if args.only_backward:
        print(json.dumps({"shape": [B, T, U, V, H], "dtype": args.dtype, "fused_backward_stage_ms": backward_stages()}))
        return
    res = {
        "shape": dict(B=B, T=T, U=U, V=V, H=H, nodes=L), "dtype": args.dtype, "device": torch.cuda.get_device_name(0),
        "lengths": "uniform" if args.ragged is None else f"ragged (seed {args.ragged}): {nodes} of {L} rows are lattice nodes",
        "fused_path": ("every row of the tensor" if args.dense else "compacted node order, " +
                       ("node count from the host" if valid_nodes is not None else "node count on the device only")),
        "gemm_flop": flop,
        "loss_fused": float(fused_fwd()), "loss_unfused": float(unfused_fwd()),
        "forward_ms": {"unfused (library GEMM + K1 + K2)": round(median_ms(unfused_fwd, args.reps), 4),
                       "fused (MFMA GEMM with K1 epilogue + K2)": round(median_ms(fused_fwd, args.reps), 4)},
        "forward_backward_ms": {"unfused": round(median_ms(unfused_step, args.reps), 4),
                                "fused forward, K3 + two library GEMMs (round 2)": round(median_ms(fused_r2_step, args.reps), 4),
                                "fused": round(median_ms(fused_step, args.reps), 4)},
    }
    if args.ragged is not None and not args.dense:
        res["forward_backward_ms"]["fused, every row of the tensor (no compacted order)"] = round(median_ms(fused_dense_step, args.reps), 4)
    if fused_backward_supported(H, V):
        res["fused_backward_stage_ms"] = backward_stages()
    f = res["forward_ms"]
    res["forward_speedup"] = round(f["unfused (library GEMM + K1 + K2)"] / f["fused (MFMA GEMM with K1 epilogue + K2)"], 3)
    res["forward_backward_speedup"] = round(res["forward_backward_ms"]["unfused"] / res["forward_backward_ms"]["fused"], 3)
    # (the fused path multiplies the nodes inside the lattices only)
    res["fused_forward_TFLOPs"] = round(flop * (1.0 if args.dense else nodes / L) /
                                        (f["fused (MFMA GEMM with K1 epilogue + K2)"] * 1e-3) / 1e12, 1)
    out = json.dumps(res, indent=1)
    print(out)
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(out + "\n")


if __name__ == "__main__":
    main()
