"""ctypes binding of ``librnnt_mi355.so`` — the C ABI declared in ``include/rnnt_mi355.h``.

The library is built in-tree by ``__graft_entry__.build()`` (``make -C rnnt_amd/csrc``).
There is NO fallback: if the shared object is missing or does not export the
ABI, importing the loss raises.  Nothing here touches ``oracle/``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import sys
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# (No environment override: the product loads its own in-tree library and nothing else.  Developer A/B runs of
# differently-built libraries go through tools/with_lib.py, which sets LIB_PATH before anything is loaded.)
LIB_PATH = os.path.join(_HERE, "lib", "librnnt_mi355.so")
CSRC_DIR = os.path.join(_HERE, "csrc")
ABI_VERSION = 1

# status codes (include/rnnt_mi355.h: rnntStatus_t)
RNNT_STATUS_SUCCESS = 0
RNNT_STATUS_STATIC_WORKSPACE_EXCEEDED = 5
STATIC_WORKSPACE_BYTES = 32 << 20  # RNNT_STATIC_WORKSPACE_BYTES
# dtype / reduction enums
DTYPE_F32, DTYPE_BF16, DTYPE_F16 = 0, 1, 2
REDUCTION_NONE, REDUCTION_MEAN, REDUCTION_SUM = 0, 1, 2

EXPORTED_SYMBOLS = (
    "rnntGetStatusString",
    "rnnt_mi355_abi_version",
    "rnnt_get_workspace_size",
    "get_workspace_size",
    "rnnt_forward",
    "rnnt_backward",
    "compute_rnnt_loss",
    "compute_rnnt_loss_static",
    "rnnt_reduce_costs",
    "rnnt_joint_loss_forward",
    "rnnt_joint_loss_forward_strided",
    "rnnt_joint_loss_backward_workspace_size",
    "rnnt_joint_loss_backward",
    "rnnt_joint_loss_backward_strided",
    "rnnt_joint_loss_compact_workspace_size",
    "rnnt_joint_loss_forward_compact",
    "rnnt_joint_loss_backward_compact_workspace_size",
    "rnnt_joint_loss_backward_compact",
    "rnnt_compact_rows",
    "rnnt_forward_compact",
    "rnnt_backward_compact",
    "rnnt_joint_hidden_forward",
    "rnnt_joint_hidden_workspace_size",
    "rnnt_joint_hidden_backward",
    "rnnt_joint_hidden_forward_ragged",
    "rnnt_joint_hidden_backward_ragged",
    "rnnt_joint_hidden_forward_compact",
    "rnnt_joint_hidden_backward_compact",
    "rnnt_row_argmax",
    "rnnt_row_max_logp",
    "rnnt_event_create",
    "rnnt_event_destroy",
    "rnnt_event_record",
    "rnnt_event_elapsed_ms",
)


class RnntOptions(ctypes.Structure):
    """Mirror of ``rnntOptions`` (include/rnnt_mi355.h)."""

    _fields_ = [
        ("stream", ctypes.c_void_p),
        ("blank_label", ctypes.c_int),
        ("maxT", ctypes.c_int),
        ("maxU", ctypes.c_int),
        ("dtype", ctypes.c_int),
        ("stage_events", ctypes.POINTER(ctypes.c_void_p)),
    ]


class RnntLibraryError(RuntimeError):
    pass


_lock = threading.Lock()
_lib = None


def build(verbose: bool = False) -> str:
    """Compile the HIP library for gfx950 (cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    # PYTHON: the compiled binding must be built against the torch of the interpreter that will load it
    subprocess.check_call(["make", "-C", CSRC_DIR, "PYTHON=" + sys.executable], stdout=out)
    return LIB_PATH


def load() -> ctypes.CDLL:
    """Load the library once per process and type its entry points."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RnntLibraryError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C rnnt_amd/csrc`). rnnt_amd has no CPU or PyTorch fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name in EXPORTED_SYMBOLS:
            if not hasattr(lib, name):
                raise RnntLibraryError(f"{LIB_PATH} does not export {name}")
        vp, ip, fp = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p  # raw device addresses
        lib.rnntGetStatusString.argtypes = [ctypes.c_int]
        lib.rnntGetStatusString.restype = ctypes.c_char_p
        lib.rnnt_mi355_abi_version.argtypes = []
        lib.rnnt_mi355_abi_version.restype = ctypes.c_int
        lib.rnnt_get_workspace_size.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.POINTER(ctypes.c_size_t)]
        lib.rnnt_get_workspace_size.restype = ctypes.c_int
        lib.get_workspace_size.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.POINTER(ctypes.c_size_t)]
        lib.get_workspace_size.restype = ctypes.c_int
        lib.rnnt_forward.argtypes = [vp, ip, ip, ip, ctypes.c_int, ctypes.c_int, fp, fp, ctypes.c_int,
                                     vp, RnntOptions]
        lib.rnnt_forward.restype = ctypes.c_int
        lib.rnnt_backward.argtypes = [vp, vp, ip, ip, ip, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int,
                                      ctypes.c_float, vp, RnntOptions]
        lib.rnnt_backward.restype = ctypes.c_int
        lib.compute_rnnt_loss.argtypes = [vp, vp, ip, ip, ip, ctypes.c_int, ctypes.c_int, fp, vp,
                                          RnntOptions]
        lib.compute_rnnt_loss.restype = ctypes.c_int
        lib.compute_rnnt_loss_static.argtypes = [vp, vp, ip, ip, ip, ctypes.c_int, ctypes.c_int, fp, RnntOptions]
        lib.compute_rnnt_loss_static.restype = ctypes.c_int
        lib.rnnt_reduce_costs.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        lib.rnnt_reduce_costs.restype = ctypes.c_int
        lib.rnnt_joint_loss_forward.argtypes = [vp, vp, fp, vp, ip, ip, ip, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                fp, fp, ctypes.c_int, vp, RnntOptions]
        lib.rnnt_joint_loss_forward.restype = ctypes.c_int
        lib.rnnt_joint_loss_forward_strided.argtypes = [vp, vp, fp, vp, ctypes.c_int, ip, ip, ip, ctypes.c_int, ctypes.c_int,
                                                        ctypes.c_int, fp, fp, ctypes.c_int, vp, RnntOptions]
        lib.rnnt_joint_loss_forward_strided.restype = ctypes.c_int
        lib.rnnt_joint_loss_backward_workspace_size.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_size_t)]
        lib.rnnt_joint_loss_backward_workspace_size.restype = ctypes.c_int
        lib.rnnt_joint_loss_backward.argtypes = [vp, vp, vp, ip, ip, ip, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 fp, ctypes.c_int, ctypes.c_float, vp, vp, fp, vp, vp, RnntOptions]
        lib.rnnt_joint_loss_backward.restype = ctypes.c_int
        lib.rnnt_joint_loss_backward_strided.argtypes = [vp, vp, vp, ctypes.c_int, ip, ip, ip, ctypes.c_int, ctypes.c_int,
                                                         ctypes.c_int, fp, ctypes.c_int, ctypes.c_float, vp, vp, fp, vp, vp,
                                                         RnntOptions]
        lib.rnnt_joint_loss_backward_strided.restype = ctypes.c_int
        lib.rnnt_joint_loss_compact_workspace_size.argtypes = [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_size_t)]
        lib.rnnt_joint_loss_compact_workspace_size.restype = ctypes.c_int
        lib.rnnt_joint_loss_forward_compact.argtypes = [vp, vp, fp, vp, ctypes.c_int, ctypes.c_longlong, ip, ip, ip,
                                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, fp, ctypes.c_int,
                                                        vp, vp, RnntOptions]
        lib.rnnt_joint_loss_forward_compact.restype = ctypes.c_int
        lib.rnnt_joint_loss_backward_compact_workspace_size.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong,
                                                                        ctypes.POINTER(ctypes.c_size_t)]
        lib.rnnt_joint_loss_backward_compact_workspace_size.restype = ctypes.c_int
        lib.rnnt_joint_loss_backward_compact.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_longlong, ip, ip, ip,
                                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int,
                                                         ctypes.c_float, vp, ctypes.c_int, vp, fp, vp, vp, vp, RnntOptions]
        lib.rnnt_joint_loss_backward_compact.restype = ctypes.c_int
        lib.rnnt_compact_rows.argtypes = [ip, ip, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, vp, vp,
                                          ctypes.c_void_p]
        lib.rnnt_compact_rows.restype = ctypes.c_int
        lib.rnnt_forward_compact.argtypes = [vp, ctypes.c_longlong, ip, ip, ip, ctypes.c_int, ctypes.c_int, fp, fp,
                                             ctypes.c_int, vp, RnntOptions]
        lib.rnnt_forward_compact.restype = ctypes.c_int
        lib.rnnt_backward_compact.argtypes = [vp, vp, ctypes.c_longlong, ip, ip, ip, ctypes.c_int, ctypes.c_int, fp,
                                              ctypes.c_int, ctypes.c_float, vp, RnntOptions]
        lib.rnnt_backward_compact.restype = ctypes.c_int
        lib.rnnt_joint_hidden_forward.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        lib.rnnt_joint_hidden_forward.restype = ctypes.c_int
        lib.rnnt_joint_hidden_workspace_size.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                         ctypes.POINTER(ctypes.c_size_t)]
        lib.rnnt_joint_hidden_workspace_size.restype = ctypes.c_int
        lib.rnnt_joint_hidden_backward.argtypes = [vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_int, ctypes.c_int, vp, ctypes.c_void_p]
        lib.rnnt_joint_hidden_backward.restype = ctypes.c_int
        lib.rnnt_joint_hidden_forward_ragged.argtypes = [vp, vp, vp, ip, ip, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                         ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        lib.rnnt_joint_hidden_forward_ragged.restype = ctypes.c_int
        lib.rnnt_joint_hidden_backward_ragged.argtypes = [vp, vp, vp, vp, ip, ip, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                          ctypes.c_int, ctypes.c_int, vp, ctypes.c_void_p]
        lib.rnnt_joint_hidden_backward_ragged.restype = ctypes.c_int
        lib.rnnt_joint_hidden_forward_compact.argtypes = [vp, vp, vp, ctypes.c_longlong, ip, ip, ctypes.c_int, ctypes.c_int,
                                                          ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        lib.rnnt_joint_hidden_forward_compact.restype = ctypes.c_int
        lib.rnnt_joint_hidden_backward_compact.argtypes = [vp, vp, ctypes.c_longlong, vp, vp, ip, ip, ctypes.c_int,
                                                           ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp,
                                                           ctypes.c_void_p]
        lib.rnnt_joint_hidden_backward_compact.restype = ctypes.c_int
        lib.rnnt_row_argmax.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_void_p]
        lib.rnnt_row_argmax.restype = ctypes.c_int
        lib.rnnt_row_max_logp.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_void_p]
        lib.rnnt_row_max_logp.restype = ctypes.c_int
        lib.rnnt_event_create.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        lib.rnnt_event_create.restype = ctypes.c_int
        lib.rnnt_event_destroy.argtypes = [ctypes.c_void_p]
        lib.rnnt_event_destroy.restype = ctypes.c_int
        lib.rnnt_event_record.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.rnnt_event_record.restype = ctypes.c_int
        lib.rnnt_event_elapsed_ms.argtypes = [ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.POINTER(ctypes.c_float)]
        lib.rnnt_event_elapsed_ms.restype = ctypes.c_int
        if lib.rnnt_mi355_abi_version() != ABI_VERSION:
            raise RnntLibraryError(
                f"{LIB_PATH}: ABI version {lib.rnnt_mi355_abi_version()} != expected {ABI_VERSION}; rebuild")
        _lib = lib
    return _lib


def check(status: int, what: str) -> None:
    """Turn a non-zero ``rnntStatus_t`` into ``RuntimeError`` (no exceptions cross the C ABI)."""
    if status != RNNT_STATUS_SUCCESS:
        msg = load().rnntGetStatusString(status).decode()
        raise RuntimeError(f"{what}: {msg} (rnntStatus_t={status})")


def workspace_bytes(maxT: int, maxU: int, minibatch: int) -> int:
    n = ctypes.c_size_t(0)
    check(load().rnnt_get_workspace_size(maxT, maxU, minibatch, ctypes.byref(n)), "rnnt_get_workspace_size")
    return int(n.value)


class StageEvents:
    """Three hipEvents the library records at its stage boundaries (bench.py only)."""

    def __init__(self, n: int = 3):
        lib = load()
        self.n = n
        self.handles = (ctypes.c_void_p * n)()
        for i in range(n):
            h = ctypes.c_void_p()
            check(lib.rnnt_event_create(ctypes.byref(h)), "rnnt_event_create")
            self.handles[i] = h.value

    def ptr(self):
        return ctypes.cast(self.handles, ctypes.POINTER(ctypes.c_void_p))

    def address(self) -> int:
        """Address of the handle array, for the compiled binding (an ``int`` argument of ``torch.ops``)."""
        return ctypes.addressof(self.handles)

    def elapsed_ms(self, i: int, j: int) -> float:
        ms = ctypes.c_float(0.0)
        check(load().rnnt_event_elapsed_ms(self.handles[i], self.handles[j], ctypes.byref(ms)),
              "rnnt_event_elapsed_ms")
        return float(ms.value)

    def close(self):
        lib = load()
        for i in range(self.n):
            if self.handles[i]:
                lib.rnnt_event_destroy(self.handles[i])
                self.handles[i] = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass
