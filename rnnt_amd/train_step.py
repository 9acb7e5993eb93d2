"""The loss's consumer without host round trips.

The reference's ``iter_one_batch`` (``bin/train.py:22-51``) reads the loss back three times per
micro-batch: ``torch.isnan(loss)`` and ``torch.isinf(loss)`` in ``if`` statements (``:36-41``) and
``loss.item()`` for the log (``:51``).  Each of them drains the stream, so the asynchronous loss
(K1 -> K2 -> K3 queued back to back) is serialised by its caller.  :class:`GuardedStep` keeps the
same semantics with every decision taken on the device:

* a NaN or infinite loss contributes nothing: its gradients are replaced by zeros before they are
  added to ``.grad`` and the clipping of that micro-batch is skipped, exactly what returning before
  ``backward()`` does upstream (``:36-41``);
* the loss is divided by ``num_gpu`` (if > 1) and by ``accumulation_steps`` (``:34-35,42``);
* ``clip_grad_norm_(parameters, max_grad_norm)`` after every micro-batch that was not skipped
  (``:45-49``), with the norm kept on the device;
* the values the reference returns (``loss.item(), grad_norm``; ``0, 0`` for a skipped batch) come
  back as 0-d device tensors, and running sums are kept for the log line (``:98``), to be read
  whenever the caller chooses (e.g. once per ``log_interval``), not every step.

Nothing here needs the HIP library: it is plain tensor code and runs wherever the model runs.
"""
from typing import Iterable, Optional, Tuple

import torch

__all__ = ["GuardedStep"]


class GuardedStep:
    """Device-side equivalent of the reference's ``iter_one_batch`` tail (``bin/train.py:34-51``).

    Args:
        parameters: the parameters ``loss`` is differentiated with respect to.
        accumulation_steps: ``config.training.accumulation_steps``.
        max_grad_norm: ``config.training.max_grad_norm`` (falsy: no clipping, grad_norm is 0).
        num_gpu: ``config.training.num_gpu``; the loss is divided by it when > 1 (``:34-35``).
            Leave it at 1 with :class:`rnnt_amd.ShardedRNNTLoss`, whose 'mean' already divides by the
            global batch.
        group: one-process-per-GPU training (``ShardedRNNTLoss``): pass the process group (or ``True``
            for the default one) and every micro-batch's parameter gradients are SUM all-reduced across
            the ranks before they reach ``.grad`` — what ``DataParallel`` did for the reference.  This
            class calls ``torch.autograd.grad``, which fires no ``backward()`` hooks, so wrapping the
            model in DistributedDataParallel would NOT synchronise anything here; use ``group=``.
            The sharded loss has the same value on every rank, so all ranks skip the same batches.
    """

    def __init__(self, parameters: Iterable[torch.nn.Parameter], accumulation_steps: int = 1,
                 max_grad_norm: Optional[float] = None, num_gpu: int = 1, group=None):
        self.params = [p for p in parameters if p.requires_grad]
        if not self.params:
            raise ValueError("no parameter requires a gradient")
        if accumulation_steps < 1:
            raise ValueError("accumulation_steps must be at least 1")
        self.accumulation_steps = int(accumulation_steps)
        self.max_grad_norm = float(max_grad_norm) if max_grad_norm else None
        self.num_gpu = int(num_gpu)
        self.group = group
        dev = self.params[0].device
        self.total_loss = torch.zeros((), dtype=torch.float32, device=dev)  # as `total_loss += loss_val`
        self.skipped = torch.zeros((), dtype=torch.int64, device=dev)       # micro-batches dropped
        self.steps = 0

    def __call__(self, loss: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Backward + guard + clip for one micro-batch; returns ``(loss_value, grad_norm)`` as 0-d
        device tensors (both 0 when the batch was skipped).  No host synchronisation."""
        loss = loss.reshape(())
        if self.num_gpu > 1:
            loss = loss / self.num_gpu
        finite = torch.isfinite(loss.detach())
        loss = loss / self.accumulation_steps
        grads = torch.autograd.grad(loss, self.params, allow_unused=True)
        if self.group is not None:
            grads = self._sum_over_ranks(grads)
        for p, g in zip(self.params, grads):
            if g is None:
                continue
            g = torch.where(finite, g, torch.zeros((), dtype=g.dtype, device=g.device))
            if p.grad is None:
                p.grad = g.contiguous()
            else:
                p.grad.add_(g)
        zero = torch.zeros((), dtype=torch.float32, device=loss.device)
        grad_norm = zero
        if self.max_grad_norm:
            have = [p.grad for p in self.params if p.grad is not None]
            norms = torch._foreach_norm(have, 2.0)
            total = torch.linalg.vector_norm(torch.stack([n.to(torch.float32) for n in norms]), 2.0)
            # clip_grad_norm_'s coefficient; 1 for a skipped batch (upstream returns before clipping)
            coef = torch.clamp(self.max_grad_norm / (total + 1e-6), max=1.0)
            coef = torch.where(finite, coef, torch.ones_like(coef))
            torch._foreach_mul_(have, [coef.to(g.dtype) for g in have])
            grad_norm = torch.where(finite, total, zero)
        value = torch.where(finite, loss.detach().to(torch.float32), zero)
        self.total_loss += value
        self.skipped += (~finite).to(torch.int64)
        self.steps += 1
        return value, grad_norm

    def _sum_over_ranks(self, grads):
        """One flat SUM all-reduce per dtype of this micro-batch's gradients (absent ones as zeros, so the
        ranks issue identical collectives)."""
        import torch.distributed as dist

        group = None if self.group is True else self.group
        full = [g if g is not None else torch.zeros_like(p) for p, g in zip(self.params, grads)]
        out = list(full)
        # (first-occurrence order of the parameters: a set of dtypes iterates in an order that may differ between
        # the rank processes, and the ranks must issue their collectives in the same order)
        for dtype in dict.fromkeys(g.dtype for g in full):
            idx = [i for i, g in enumerate(full) if g.dtype == dtype]
            flat = torch.cat([full[i].reshape(-1) for i in idx])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            off = 0
            for i in idx:
                n = full[i].numel()
                out[i] = flat[off:off + n].view_as(full[i])
                off += n
        return out

    def average_loss(self) -> torch.Tensor:
        """``total_loss / (step + 1)`` of the reference's log line (``bin/train.py:98``), on the device."""
        return self.total_loss / max(self.steps, 1)
