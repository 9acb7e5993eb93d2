#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/profile.sh (gpurun_out/prof) into the files under profiles/.

    python tools/summarize_profile.py [gpurun_out/prof] [rNN]

Writes profiles/<rNN>_kernel_stats.csv (the --kernel-trace --stats summary, rnnt_* kernels first),
profiles/<rNN>_pmc_fetch_size.csv / _pmc_write_size.csv (one line per rnnt_* dispatch) and
profiles/k3_traffic.json (HBM bytes per launch; FETCH_SIZE is in KB and is doubled on gfx950 as
MI355X_MICROARCH.md prescribes for 16-byte-per-lane streaming reads; WRITE_SIZE is in KB).
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof")
TAG = sys.argv[2] if len(sys.argv) > 2 else "r01"
OUT = os.path.join(ROOT, "profiles")
KERNELS = ("rnnt_node_spans", "rnnt_k1_lse", "rnnt_k2_lattice", "rnnt_reduce_costs", "rnnt_k3_grad")
# config 2 (bench.py default): B32 T400 U40 V4232 fp32, uniform lengths
N = 32 * 400 * 41 * 4232
ALGO = {"rnnt_k3_grad": 2 * N * 4, "rnnt_k1_lse": N * 4}


def short(name):
    for k in KERNELS:
        if k in name:
            return k
    return None


def kernel_stats():
    path = max(glob.glob(os.path.join(SRC, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)  # newest run
    rows = list(csv.reader(open(path)))
    head, body = rows[0], rows[1:]
    ours = [r for r in body if short(r[0])]
    rest = [r for r in body if not short(r[0])]
    with open(os.path.join(OUT, f"{TAG}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(head)
        w.writerows(ours + rest)
    return {short(r[0]): {"calls": int(r[1]), "avg_us": float(r[3]) / 1e3} for r in ours}


def counter(which, name):
    per = {k: [] for k in KERNELS}
    lines = []
    for path in [max(glob.glob(os.path.join(SRC, which, "*", "*_counter_collection.csv")), key=os.path.getmtime)]:
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if k and r["Counter_Name"] == name:
                per[k].append(float(r["Counter_Value"]))
                lines.append((r["Dispatch_Id"], k, r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"],
                              r["Counter_Name"], r["Counter_Value"]))
    with open(os.path.join(OUT, f"{TAG}_pmc_{name.lower()}.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel", "Grid_Size", "Workgroup_Size", "VGPR_Count", "Counter_Name",
                    "Counter_Value_KB"])
        w.writerows(lines)
    return {k: {"launches": len(v), "mean_raw_KB": sum(v) / len(v)} for k, v in per.items() if v}


def main():
    stats = kernel_stats()
    fetch = counter("fetch", "FETCH_SIZE")
    write = counter("write", "WRITE_SIZE")
    out = {"workload": "config2 B32 T400 U40 V4232 fp32", "raw": {"FETCH_SIZE": fetch, "WRITE_SIZE": write},
           "kernel_trace_avg_us": stats}
    for key, k in (("k3", "rnnt_k3_grad"), ("k1", "rnnt_k1_lse")):
        rd = 2.0 * fetch[k]["mean_raw_KB"] * 1024.0
        wr = write[k]["mean_raw_KB"] * 1024.0
        out[key] = {"read_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
                    "algorithmic_bytes": ALGO[k], "ratio": (rd + wr) / ALGO[k]}
    out["hbm_bytes_per_launch"] = out["k3"]["hbm_bytes_per_launch"]
    out["collected"] = f"round {int(TAG[1:])}" if TAG[1:].isdigit() else TAG
    out["method"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of `python3 bench.py "
                     "--steps 3 --warmup 1 --no-cpu-baseline` (tools/profile.sh), summarised by "
                     "tools/summarize_profile.py; KB units; FETCH_SIZE doubled per MI355X_MICROARCH.md")
    with open(os.path.join(OUT, "k3_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ("k3", "k1", "kernel_trace_avg_us")}, indent=1))


if __name__ == "__main__":
    main()
