#!/usr/bin/env python3
"""bench.py — RNN-T loss fwd+bwd throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config2|config3|config5|config1]

One "step" = RNNTLoss forward + backward over one synthetic batch already resident in HBM.
N=1 runs BASELINE.json configs[1] (B=32, T=400, U=40, V=4232, fp32).  For N>1 the driver starts
one rank per GPU with torch.distributed.run; every rank owns B=32 utterances (weak scaling, so
N=8 is configs[3], B=256) and the ranks exchange only the per-utterance costs (one RCCL
all-gather of 4·B bytes, rnnt_amd/parallel.py).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (K3, the fused gradient
pass): algorithmic bytes per launch (read N_valid + write N_full elements) over its mean
duration measured with hipEvents that the library records on the launch stream.  `cpu_baseline`
is the CPU oracle (oracle/rnnt_oracle.c, a restatement of warp-transducer's CPU semantics — NOT
warp-transducer itself) timed on this host on the first utterances of the very batch the GPU ran,
as one native call (`value`) and in upstream's shape (`upstream_shaped`: torch.log_softmax -> C -> autograd).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate

WORKLOADS = {
    # name: (B per GPU, T, U, V, dtype, ragged)
    "config1": dict(B=8, T=200, U=20, V=4232, dtype="f32", ragged=False),
    "config2": dict(B=32, T=400, U=40, V=4232, dtype="f32", ragged=False),
    "config3": dict(B=64, T=800, U=60, V=4232, dtype="f32", ragged=True),
    "config5": dict(B=16, T=1500, U=150, V=16384, dtype="bf16", ragged=False),
}
TORCH_DTYPES = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}


def make_device_batch(w, dev, first_utt, seed=1234):
    """Per-utterance seeded N(0,1) logits generated on the device (shard-invariant)."""
    B, T, U, V = w["B"], w["T"], w["U"], w["V"]
    dt = TORCH_DTYPES[w["dtype"]]
    acts = torch.empty((B, T, U + 1, V), dtype=dt, device=dev)
    g = torch.Generator(device=dev)
    for b in range(B):
        g.manual_seed(seed + first_utt + b)
        if dt is torch.float32:
            acts[b].normal_(0.0, 1.0, generator=g)
        else:
            for t0 in range(0, T, 100):  # bounded fp32 temporaries
                t1 = min(T, t0 + 100)
                acts[b, t0:t1] = torch.randn((t1 - t0, U + 1, V), device=dev, generator=g).to(dt)
    rng = np.random.default_rng(4321 + first_utt)
    labels = rng.integers(1, V, size=(B, U)).astype(np.int32)
    if w["ragged"]:
        act_lens = rng.integers(100, T + 1, size=B).astype(np.int32)
        label_lens = rng.integers(10, U + 1, size=B).astype(np.int32)
        act_lens[0], label_lens[0] = T, U
        for b in range(B):
            labels[b, label_lens[b]:] = 0
    else:
        act_lens = np.full(B, T, np.int32)
        label_lens = np.full(B, U, np.int32)
    return (acts, torch.from_numpy(labels).to(dev), torch.from_numpy(act_lens).to(dev),
            torch.from_numpy(label_lens).to(dev), act_lens, label_lens)


def cpu_baseline(w, acts_dev, labels_dev, al, ll, sample_B, reps):
    """The loss on this host's cores, on the FIRST sample_B utterances of the very batch the GPU was timed on
    (copied back from the device: same per-utterance-seeded logits, labels and lengths), two ways:

    * ``value``: the all-C port (oracle/rnnt_oracle.c: log-softmax, alpha/beta, gradient and log-softmax backward
      in one OpenMP native call, one utterance per thread);
    * ``upstream_shaped``: the way upstream's CPU path is shaped: ``torch.log_softmax`` in
      Python -> native alpha/beta + sparse gradient -> autograd back through the log-softmax.
    Neither is warp-transducer itself (its sources are not available here): kind = "port".
    """
    import oracle  # test infrastructure: used here only as the reported CPU baseline

    oracle.build()
    per_utt = w["T"] * (w["U"] + 1) * w["V"]
    # bounded sample: at most ~2.5e9 logits (10 GB of fp32 logits + as much for the gradient) on the host
    B = max(1, min(sample_B, w["B"], int(2.5e9 // per_utt)))
    T, U, V = w["T"], w["U"], w["V"]
    acts = acts_dev[:B].detach().float().cpu().numpy()
    labels = labels_dev[:B].cpu().numpy()
    al, ll = np.ascontiguousarray(al[:B]), np.ascontiguousarray(ll[:B])
    threads = oracle.max_threads()
    if B > 1:  # warm-up (page in, thread pool); a one-utterance sample (config 5) is its own warm-up
        oracle.rnnt_loss_c(acts[:1], labels[:1], al[:1], ll[:1])
    ts = []
    for i in range(reps):  # bounded: stop early once ~20 s of CPU work are on the clock (large lattices: config 5)
        t0 = time.perf_counter()
        c, gr, _ = oracle.rnnt_loss_c(acts, labels, al, ll, want_grads=True)
        oracle.reduce_like_upstream(c, None, "mean")
        ts.append(time.perf_counter() - t0)
        del gr
        print(f"[bench] cpu baseline run {i + 1}: {ts[-1]:.2f} s", file=sys.stderr, flush=True)
        if sum(ts) > 20.0:
            break
    reps = len(ts)
    t = float(np.median(ts))
    # the upstream-shaped path holds the logits, the log-probs and two gradients at once: half the sample
    Bu = max(1, min(B, int(1.25e9 // per_utt)))
    tu = []
    for i in range(max(2, reps // 2)):
        t0 = time.perf_counter()
        lu, gu = oracle.rnnt_loss_upstream_shaped(acts[:Bu], labels[:Bu], al[:Bu], ll[:Bu], reduction="mean",
                                                  grad_output=0.5)
        tu.append(time.perf_counter() - t0)
        del gu
        print(f"[bench] cpu baseline (upstream-shaped) run {i + 1}: {tu[-1]:.2f} s", file=sys.stderr, flush=True)
        if sum(tu) > 10.0:
            break
    tu = float(np.median(tu))
    lens = "uniform lengths" if not w["ragged"] else "the batch's own ragged lengths"
    return {
        "value": B / t, "unit": "utt/s", "cores": int(min(threads, B)), "kind": "port",
        "sample": f"value = ALL-C port (oracle/rnnt_oracle.c, OpenMP, one utterance per thread; a restatement of "
                  f"warp-transducer's CPU semantics, not warp-transducer itself), fwd+bwd on the first B={B} "
                  f"utterances of the GPU run's own batch (per-utterance seeds, copied back from the device), "
                  f"T={T} U={U} V={V} fp32, {lens}, median of {reps} runs, {t * 1e3:.0f} ms per run; "
                  f"upstream_shaped = torch.log_softmax -> C alpha/beta + sparse grad -> autograd on the first "
                  f"B={Bu} of the same utterances, {tu * 1e3:.0f} ms per run; omp_max_threads={threads}, "
                  f"torch threads={torch.get_num_threads()}, os.cpu_count={os.cpu_count()}",
        "upstream_shaped": {"value": Bu / tu, "unit": "utt/s", "cores": int(torch.get_num_threads()),
                            "kind": "port", "loss": float(lu[0])},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS) + ["custom"])
    ap.add_argument("--shape", default=None,
                    help="with --workload custom: B,T,U,V,dtype[,ragged]  e.g. 64,400,40,128,f32 (tuning sweeps; "
                         "not a BASELINE configuration)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-b", type=int, default=32)
    ap.add_argument("--cpu-reps", type=int, default=6)
    ap.add_argument("--force-sharded", action="store_true",
                    help="use the sharded harness (RCCL all-gather of costs) even with one rank")
    ap.add_argument("--no-check-lengths", action="store_true",
                    help="skip the host-side length check (fully asynchronous forward)")
    args = ap.parse_args()

    # Libraries (RCCL prints a version banner) write to fd 1; the contract is ONE JSON line on stdout.
    # Point fd 1 at stderr until the result is ready.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("for --gpus N>1 launch with: python -m torch.distributed.run --nproc-per-node N "
                             "--master-addr 127.0.0.1 bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")

    import torch.distributed as dist

    import rnnt_amd
    from rnnt_amd import _lib, loss as loss_mod

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; rnnt_amd has no CPU path")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    sharded = world > 1 or args.force_sharded
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # RCCL over xGMI

    # which device each rank really drives: the scaling numbers only mean something if the N ranks sit on N
    # distinct GPUs, so every rank reports (rank, device index, uuid, PCI address) and rank 0 prints the table
    props = torch.cuda.get_device_properties(dev)
    me = {
        "rank": rank,
        "local_rank": local_rank,
        "device": torch.cuda.current_device(),
        "uuid": str(getattr(props, "uuid", "")),
        "pci": "{:04x}:{:02x}:{:02x}".format(getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0),
                                             getattr(props, "pci_device_id", 0)),
        "name": props.name,
        "visible": os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES", "")),
    }
    if sharded:
        rccl_ranks = [None] * world
        dist.all_gather_object(rccl_ranks, me)
    else:
        rccl_ranks = [me]
    distinct_devices = len({(r["uuid"], r["pci"]) for r in rccl_ranks})
    if sharded and distinct_devices != world:
        raise SystemExit(f"{world} ranks share {distinct_devices} device(s): {rccl_ranks}")

    if args.workload == "custom":
        f = (args.shape or "").split(",")
        if len(f) < 5:
            raise SystemExit("--workload custom needs --shape B,T,U,V,dtype[,ragged]")
        w = dict(B=int(f[0]), T=int(f[1]), U=int(f[2]), V=int(f[3]), dtype=f[4], ragged=len(f) > 5 and f[5] == "ragged")
    else:
        w = dict(WORKLOADS[args.workload])
    B = w["B"]
    acts, labels, al_d, ll_d, al, ll = make_device_batch(w, dev, first_utt=rank * B)
    acts.requires_grad_(True)
    go = torch.full((1,), 0.5, dtype=torch.float32, device=dev)  # grad_output != 1 exercises the fused scale

    if sharded:
        crit = rnnt_amd.ShardedRNNTLoss(blank=0, reduction="mean", check_lengths=not args.no_check_lengths)

        def step():
            acts.grad = None
            loss = crit(acts, labels, al_d, ll_d, global_batch=B * world)
            loss.backward(go)
            return loss
    else:
        crit = rnnt_amd.RNNTLoss(blank=0, reduction="mean", check_lengths=not args.no_check_lengths)

        def step():
            acts.grad = None
            loss = crit(acts, labels, al_d, ll_d)
            loss.backward(go)
            return loss

    def fence():
        torch.cuda.synchronize(dev)
        if sharded:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        loss = step()
    fence()

    fwd_ev = [_lib.StageEvents(3) for _ in range(args.steps)]
    bwd_ev = [_lib.StageEvents(2) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss_mod.set_stage_events(fwd_ev[i], bwd_ev[i])
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    loss_mod.set_stage_events(None, None)

    k1 = float(np.mean([e.elapsed_ms(0, 1) for e in fwd_ev]))
    k2 = float(np.mean([e.elapsed_ms(1, 2) for e in fwd_ev]))
    k3 = float(np.mean([e.elapsed_ms(0, 1) for e in bwd_ev]))

    def span_ms(a, b):  # start of a step's K1 -> end of its K3, on the launch stream
        import ctypes
        ms = ctypes.c_float(0.0)
        _lib.check(_lib.load().rnnt_event_elapsed_ms(a, b, ctypes.byref(ms)), "rnnt_event_elapsed_ms")
        return float(ms.value)

    per_step = [span_ms(f.handles[0], b.handles[1]) for f, b in zip(fwd_ev, bwd_ev)]
    p10, p50, p90 = (float(x) for x in np.percentile(per_step, [10, 50, 90]))
    loss_val = float(loss.detach().cpu()[0])

    t_all = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if sharded:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
    elapsed = float(t_all.item())

    esz = 4 if w["dtype"] == "f32" else 2
    V = w["V"]
    n_valid = int(sum(int(t) * (int(u) + 1) for t, u in zip(al, ll))) * V
    n_full = B * w["T"] * (w["U"] + 1) * V
    k3_bytes = (n_valid + n_full) * esz           # K3: read valid logits once, write every grad once
    k1_bytes = n_valid * esz                      # K1: read valid logits once
    step_bytes = (2 * n_valid + n_full) * esz     # SURVEY.md §8d algorithmic bytes per batch
    ms_per_step = elapsed / args.steps * 1e3
    value = B * world / (elapsed / args.steps)

    out = {
        "metric": "RNNT loss fwd+bwd utts/sec at B32 T400 U40 V4232; % of HBM BW roofline",
        "value": round(value, 2),
        "unit": "utt/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": w["dtype"],
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: RNN-T loss fwd+bwd, B={B} per GPU x {world} GPU(s), T={w['T']}, "
                        f"U={w['U']}, V={V}, {w['dtype']} logits+grads, fp32 lattice, "
                        f"{'ragged' if w['ragged'] else 'uniform'} lengths, N(0,1) logits, grad_output=0.5, "
                        f"reduction=mean",
            "global_batch": B * world,
            "parallelism": f"batch-sharded x{world}, all-gather of per-utterance costs" if sharded else "single GPU",
            "check_lengths": not args.no_check_lengths,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "rnnt_k3_grad (fused gradient: read logits, write grads)",
            "achieved": round(k3_bytes / (k3 * 1e-3) / 1e9, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(k3_bytes / (k3 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "traffic": None,
            "traffic_source": None,
            "algorithmic_bytes_per_launch": k3_bytes,
            "avg_launch_ms": round(k3, 4),
        },
        "stages_ms": {"k1_lse": round(k1, 4), "k2_lattice_and_reduce": round(k2, 4), "k3_grad": round(k3, 4)},
        "step_device_ms": {"p10": round(p10, 4), "median": round(p50, 4), "p90": round(p90, 4)},
        "k1_achieved_GBs": round(k1_bytes / (k1 * 1e-3) / 1e9, 1),
        "whole_step": {
            "algorithmic_bytes": step_bytes,
            "achieved_GBs": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
            "frac_of_hbm_peak": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        },
        "loss": loss_val,
        "rccl_ranks": rccl_ranks,
        "distinct_devices": distinct_devices,
    }
    # HBM bytes per K3 launch from the PMC counters cannot be collected inside this process: they come from the
    # committed rocprofv3 passes over this same command (tools/profile.sh), and only for the workload they ran.
    traffic_file = os.path.join(ROOT, "profiles", "k3_traffic.json")
    if os.path.exists(traffic_file) and args.workload == "config2" and world == 1:
        try:
            with open(traffic_file) as f:
                tj = json.load(f)
            out["roofline"]["traffic"] = tj.get("hbm_bytes_per_launch")
            out["roofline"]["traffic_source"] = (
                f"profiles/k3_traffic.json ({tj.get('collected', 'round 1')}: rocprofv3 --pmc FETCH_SIZE / "
                f"WRITE_SIZE passes of `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline`, FETCH_SIZE "
                f"doubled as MI355X_MICROARCH.md prescribes; NOT measured in this run)")
        except Exception:
            pass

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # free the device tensors' host shadows first; the CPU sample needs a few GB of RAM
        out["cpu_baseline"] = cpu_baseline(w, acts, labels, al, ll, args.cpu_sample_b, args.cpu_reps)
    elif rank == 0:
        out["cpu_baseline"] = None
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
