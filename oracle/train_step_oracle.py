"""Restatement of the reference's loss consumer — TEST INFRASTRUCTURE ONLY.

What ``iter_one_batch`` does with the loss once the model has produced it
(/root/reference/bin/train.py:34-51), with its host-side decisions: divide by the number of GPUs
when there are several (:34-35); a NaN or infinite loss ends the micro-batch with ``(0, 0)`` and no
backward pass (:36-41); otherwise divide by the accumulation steps (:42), back-propagate (:43), clip
the gradient norm when a limit is configured (:45-49) and hand back the loss as a Python float with
the norm (:51).  PARITY UNPINNED: the reference holds no fixtures for its training loop; the torch
calls are the ones it makes.
"""
from __future__ import annotations

import torch


def consume_loss_reference(loss, parameters, num_gpu=1, accumulation_steps=1, max_grad_norm=None):
    value = loss / num_gpu if num_gpu > 1 else loss
    if not bool(torch.isfinite(value).all()):  # isnan (:36) or isinf (:39): both leave before backward()
        return 0, 0
    value = value / accumulation_steps
    value.backward()
    norm = torch.nn.utils.clip_grad_norm_(parameters, max_grad_norm) if max_grad_norm else 0
    return value.item(), norm
