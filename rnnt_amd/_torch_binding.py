"""Loader of the compiled PyTorch binding ``librnnt_amd_torch.so`` (rnnt_amd/csrc/torch_binding.cpp).

``torch.ops.rnnt_amd.rnnt_loss`` is what ``RNNTLoss`` calls by default: stream from torch's current HIP stream,
buffers from its caching allocator, the backward pass a C++ autograd node, the C ABI of ``librnnt_mi355.so``
underneath.  Built in-tree by ``make -C rnnt_amd/csrc`` (``__graft_entry__.build()``).  No fallback: a missing or
stale library raises.
"""
from __future__ import annotations

import os
import threading

import torch

from . import _lib

TORCH_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "librnnt_amd_torch.so")

_lock = threading.Lock()
_ops = None


def _mapped_kernel_libraries():
    try:
        with open("/proc/self/maps") as fh:
            return {line.split()[-1] for line in fh if "librnnt_mi355" in line}
    except OSError:
        return set()


def load():
    """``torch.ops.rnnt_amd`` with the library loaded (once per process)."""
    global _ops
    if _ops is not None:
        return _ops
    with _lock:
        if _ops is None:
            if not os.path.exists(TORCH_LIB_PATH):
                raise _lib.RnntLibraryError(
                    f"{TORCH_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(or `make -C rnnt_amd/csrc`). rnnt_amd has no CPU or PyTorch fallback.")
            _lib.load()  # librnnt_mi355.so first: the binding's DT_NEEDED entry then resolves to this very copy
            torch.ops.load_library(TORCH_LIB_PATH)
            ops = torch.ops.rnnt_amd
            mapped = _mapped_kernel_libraries()
            if len(mapped) > 1:  # e.g. a tools/with_lib.py library built without -Wl,-soname,librnnt_mi355.so
                raise _lib.RnntLibraryError(
                    "two copies of the kernels library are mapped (" + ", ".join(sorted(mapped)) + "): the compiled "
                    "binding would run one and the ctypes entry points the other; build the override with "
                    "`-Wl,-soname,librnnt_mi355.so` or select binding='ctypes'")
            if ops.abi_version() != _lib.ABI_VERSION:
                raise _lib.RnntLibraryError(f"{TORCH_LIB_PATH}: ABI version {ops.abi_version()} != {_lib.ABI_VERSION}; rebuild")
            _ops = ops
    return _ops
