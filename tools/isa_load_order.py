#!/usr/bin/env python3
"""Per-kernel order of vector-memory loads, stores and `s_waitcnt vmcnt(N)` in the gfx950 ISA of the
library's kernels: the view that showed loads being waited for one at a time ("Address
space and load issue").  No GPU needed.

    python tools/isa_load_order.py [regex over mangled kernel names] [--src joint_mi355.hip]

Prints, per kernel: instruction count, number of flat memory instructions (must be 0), number of
`load; s_waitcnt vmcnt(0); load` patterns, and the sequence (L load, S store, Wn wait)."""
import argparse
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rnnt_amd", "csrc")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pattern", nargs="?", default=".")
    ap.add_argument("--src", default="rnnt_mi355.hip")
    ap.add_argument("--width", type=int, default=400)
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                        "--offload-device-only", "-S", "-o", asm, os.path.join(CSRC, args.src)],
                       check=True, stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    vmcnt = re.compile(r"vmcnt\((\d+)\)")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines)
              if l.startswith("_Z") and l.split(";")[0].rstrip().endswith(":")]
    for start, name in starts:
        if not re.search(args.pattern, name):
            continue
        seq, n, flat = [], 0, 0
        for l in lines[start + 1:]:
            t = l.strip()
            if t.startswith("s_endpgm"):
                break
            if not t or t.startswith((";", ".")) or t.endswith(":"):
                continue
            n += 1
            op = t.split()[0]
            if op.startswith("flat_"):
                flat += 1
            if op.startswith(("global_load", "buffer_load", "flat_load")):
                seq.append("L")
            elif op.startswith(("global_store", "buffer_store", "flat_store")):
                seq.append("S")
            elif op.startswith("s_waitcnt") and "vmcnt" in t:
                seq.append("W" + vmcnt.search(t).group(1))
        s = " ".join(seq)
        print(f"{name}\n  instructions {n}  flat {flat}  load-wait0-load {len(re.findall(r'L W0(?= L)', s))}\n  {s[:args.width]}")


if __name__ == "__main__":
    main()
