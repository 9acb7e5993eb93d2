"""CPU ORACLE for the RNN-T loss — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  ``rnnt_amd`` never does.

Three independent statements of the loss the reference obtains from
``warprnnt_pytorch.RNNTLoss`` (/root/reference/src/rnnt/model.py:5,87,116):

* :func:`rnnt_loss_c` — ``rnnt_oracle.c``: C/OpenMP restatement of the upstream
  CPU path (log_softmax -> alpha/beta -> sparse grad -> log_softmax backward),
  fp32 or fp64.  This is the parity oracle and the CPU baseline.
* :func:`rnnt_loss_dp_f64` — float64 torch dynamic programme; gradients come
  from autograd, i.e. independently of the closed-form gradient.
* :func:`rnnt_loss_bruteforce` — explicit enumeration of every alignment of a
  tiny lattice.

Restatements of the rows next to the loss live beside it, under the same rule:
``joint_oracle`` (src/rnnt/model.py:39-59), ``decode_oracle`` (model.py:145-165 and
beam_searcher.py:65-127) and ``train_step_oracle`` (bin/train.py:34-51).

PARITY UNPINNED: the reference has no tests or golden vectors for this path and
the upstream package cannot be built or imported here.  The
three statements are checked against each other and against two known-answer
vectors of the upstream package's own unit tests (B1·T2·U2·V5 and the
two-utterance B2·T4·U2·V3 case, both recalled; ``tests/cases.py``) in
``tests/test_oracle.py``.
"""
from __future__ import annotations

import ctypes
import itertools
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librnnt_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile ``librnnt_oracle.so`` with gcc (a few seconds)."""
    src = [os.path.join(_HERE, f) for f in ("rnnt_oracle.c", "rnnt_oracle_impl.h")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "librnnt_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        lib = ctypes.CDLL(_LIB_PATH)
        for name, real in (("rnnt_oracle_f32", ctypes.c_float), ("rnnt_oracle_f64", ctypes.c_double)):
            fn = getattr(lib, name)
            p = ctypes.POINTER(real)
            ip = ctypes.POINTER(ctypes.c_int)
            fn.argtypes = [p, ip, ip, ip, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                           ctypes.c_int, p, p, p, ctypes.c_int]
            fn.restype = ctypes.c_int
        for name, real in (("rnnt_oracle_logprobs_f32", ctypes.c_float), ("rnnt_oracle_logprobs_f64", ctypes.c_double)):
            fn = getattr(lib, name)
            p = ctypes.POINTER(real)
            ip = ctypes.POINTER(ctypes.c_int)
            fn.argtypes = [p, ip, ip, ip, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                           ctypes.c_int, p, p, ctypes.c_int]
            fn.restype = ctypes.c_int
        lib.rnnt_oracle_max_threads.restype = ctypes.c_int
        _lib = lib
    return _lib


def max_threads() -> int:
    return int(_load().rnnt_oracle_max_threads())


def rnnt_loss_c(acts, labels, act_lens, label_lens, blank=0, want_grads=True, dtype=np.float32,
                num_threads=0):
    """Per-utterance costs ``[B]`` and ``d costs[b] / d acts[b]`` ``[B,T,U+1,V]`` (or None).

    Restates the call at /root/reference/src/rnnt/model.py:116 with the upstream
    CPU-path semantics; reductions ('mean' = sum/B) are left to the caller.
    """
    lib = _load()
    dtype = np.dtype(dtype)
    acts = np.ascontiguousarray(acts, dtype=dtype)
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    act_lens = np.ascontiguousarray(act_lens, dtype=np.int32)
    label_lens = np.ascontiguousarray(label_lens, dtype=np.int32)
    B, T, U1, V = acts.shape
    if labels.ndim != 2 or labels.shape[0] != B:
        raise ValueError("labels must be [B, U]")
    if labels.shape[1] != U1 - 1:
        # the C side indexes labels with row stride maxU-1
        lab = np.zeros((B, U1 - 1), dtype=np.int32)
        w = min(U1 - 1, labels.shape[1])
        lab[:, :w] = labels[:, :w]
        labels = lab
    costs = np.empty(B, dtype=dtype)
    grads = np.empty_like(acts) if want_grads else None
    llb = np.empty(B, dtype=dtype)
    real = ctypes.c_float if dtype == np.float32 else ctypes.c_double
    fn = lib.rnnt_oracle_f32 if dtype == np.float32 else lib.rnnt_oracle_f64
    P = ctypes.POINTER(real)
    IP = ctypes.POINTER(ctypes.c_int)
    rc = fn(acts.ctypes.data_as(P), labels.ctypes.data_as(IP), act_lens.ctypes.data_as(IP),
            label_lens.ctypes.data_as(IP), B, T, U1, V, int(blank), costs.ctypes.data_as(P),
            grads.ctypes.data_as(P) if want_grads else None, llb.ctypes.data_as(P), int(num_threads))
    if rc != 0:
        raise ValueError("rnnt_oracle: invalid arguments (lengths/labels out of range?)")
    return costs, grads, llb


def rnnt_loss_upstream_shaped(acts, labels, act_lens, label_lens, blank=0, reduction="mean", grad_output=1.0,
                              num_threads=0):
    """The loss the way upstream's CPU path is SHAPED: ``torch.log_softmax`` in Python,
    the native call for alpha/beta and the sparse gradient w.r.t. the log-probs, autograd back through the
    log_softmax, the reduction in Python.  Returns ``(loss ndarray, grads ndarray)``; same numbers as
    :func:`rnnt_loss_c` + :func:`reduce_like_upstream` up to rounding (tests/test_oracle.py)."""
    import torch

    lib = _load()
    x = torch.from_numpy(np.ascontiguousarray(acts)).requires_grad_(True)
    dtype = np.dtype(np.float64 if x.dtype is torch.float64 else np.float32)
    x32 = x if x.dtype in (torch.float32, torch.float64) else x.float()
    lp = torch.nn.functional.log_softmax(x32, dim=-1)
    lp_np = lp.detach().numpy()
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    act_lens = np.ascontiguousarray(act_lens, dtype=np.int32)
    label_lens = np.ascontiguousarray(label_lens, dtype=np.int32)
    B, T, U1, V = lp_np.shape
    costs = np.empty(B, dtype=dtype)
    g_lp = np.empty_like(lp_np)
    real = ctypes.c_float if dtype == np.float32 else ctypes.c_double
    fn = lib.rnnt_oracle_logprobs_f32 if dtype == np.float32 else lib.rnnt_oracle_logprobs_f64
    P, IP = ctypes.POINTER(real), ctypes.POINTER(ctypes.c_int)
    rc = fn(lp_np.ctypes.data_as(P), labels.ctypes.data_as(IP), act_lens.ctypes.data_as(IP),
            label_lens.ctypes.data_as(IP), B, T, U1, V, int(blank), costs.ctypes.data_as(P), g_lp.ctypes.data_as(P),
            int(num_threads))
    if rc != 0:
        raise ValueError("rnnt_oracle_logprobs: invalid arguments (lengths/labels out of range?)")
    scale = {"mean": 1.0 / B, "sum": 1.0, "none": 1.0}[reduction]
    go = np.broadcast_to(np.asarray(grad_output, dtype=dtype), (B,)).reshape(B, 1, 1, 1) * scale
    lp.backward(torch.from_numpy(g_lp * go))
    loss = costs if reduction == "none" else np.array([costs.sum(dtype=dtype) * scale], dtype=dtype)
    return loss, x.grad.numpy()


def reduce_like_upstream(costs, grads, reduction="mean", grad_output=1.0):
    """A6: upstream's Python-side reduction. 'mean' divides loss and grads by B."""
    B = costs.shape[0]
    if reduction == "mean":
        loss = np.array([costs.sum(dtype=costs.dtype) / B], dtype=costs.dtype)
        g = None if grads is None else grads * (np.asarray(grad_output, dtype=grads.dtype) / B)
    elif reduction == "sum":
        loss = np.array([costs.sum(dtype=costs.dtype)], dtype=costs.dtype)
        g = None if grads is None else grads * np.asarray(grad_output, dtype=grads.dtype)
    elif reduction == "none":
        loss = costs
        go = np.broadcast_to(np.asarray(grad_output, dtype=costs.dtype), (B,))
        g = None if grads is None else grads * go.reshape(B, 1, 1, 1)
    else:
        raise ValueError(reduction)
    return loss, g


# ----------------------------------------------------------------------------
# independent statement 2: float64 DP in torch, gradients from autograd
# ----------------------------------------------------------------------------
def rnnt_loss_dp_f64(acts, labels, act_lens, label_lens, blank=0, want_grads=True):
    """Small cases only (Python loops over the lattice)."""
    import torch

    x = torch.as_tensor(np.asarray(acts, dtype=np.float64)).clone().requires_grad_(want_grads)
    B = x.shape[0]
    lp = torch.log_softmax(x, dim=-1)
    costs = []
    for b in range(B):
        T, U = int(act_lens[b]), int(label_lens[b]) + 1
        y = [int(v) for v in labels[b][: U - 1]]
        alpha = [[None] * U for _ in range(T)]
        for t in range(T):
            for u in range(U):
                if t == 0 and u == 0:
                    alpha[0][0] = lp.new_zeros(())
                    continue
                terms = []
                if t > 0:
                    terms.append(alpha[t - 1][u] + lp[b, t - 1, u, blank])
                if u > 0:
                    terms.append(alpha[t][u - 1] + lp[b, t, u - 1, y[u - 1]])
                alpha[t][u] = terms[0] if len(terms) == 1 else torch.logaddexp(terms[0], terms[1])
        costs.append(-(alpha[T - 1][U - 1] + lp[b, T - 1, U - 1, blank]))
    costs = torch.stack(costs)
    grads = None
    if want_grads:
        # d costs[b] / d acts[b]: utterances are independent, so one backward of the sum suffices
        costs.sum().backward()
        grads = x.grad.numpy()
    return costs.detach().numpy(), grads


# ----------------------------------------------------------------------------
# independent statement 3: brute-force enumeration of alignments
# ----------------------------------------------------------------------------
def rnnt_loss_bruteforce(acts, labels, act_lens, label_lens, blank=0):
    """-log of the sum over every monotone alignment; tiny lattices only."""
    acts = np.asarray(acts, dtype=np.float64)
    B = acts.shape[0]
    mx = acts.max(axis=-1, keepdims=True)
    lp = acts - mx - np.log(np.exp(acts - mx).sum(axis=-1, keepdims=True))
    costs = np.zeros(B)
    for b in range(B):
        T, U = int(act_lens[b]), int(label_lens[b])
        y = [int(v) for v in labels[b][:U]]
        # an alignment = where the U label emissions sit among the T-1 blank moves that
        # precede the final blank: choose positions of U emissions in a string of T-1+U moves
        total = -math.inf
        n_moves = T - 1 + U
        for emit_pos in itertools.combinations(range(n_moves), U):
            emit = set(emit_pos)
            t = u = 0
            s = 0.0
            for m in range(n_moves):
                if m in emit:
                    s += lp[b, t, u, y[u]]
                    u += 1
                else:
                    s += lp[b, t, u, blank]
                    t += 1
            s += lp[b, T - 1, U, blank]
            total = np.logaddexp(total, s)
        costs[b] = -total
    return costs
