#!/usr/bin/env python3
"""Developer runner: a script of this repository against a differently-built kernels library.

    python tools/with_lib.py gpurun_tmp/libbwd_MMA.so tools/bench_fused.py --only-backward

Sets ``rnnt_amd._lib.LIB_PATH`` before anything is loaded and forces the ctypes binding (the compiled torch binding
is linked against the in-tree library: rnnt_amd.loss.set_binding).  The product itself has no such switch (tests/test_abi.py).
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    lib, script = os.path.abspath(sys.argv[1]), sys.argv[2]
    from rnnt_amd import _lib

    _lib.LIB_PATH = lib
    from rnnt_amd import loss

    loss.set_binding("ctypes")
    sys.argv = sys.argv[2:]
    runpy.run_path(script, run_name="__main__")
