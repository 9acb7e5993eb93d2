"""CPU restatement of the reference's joint network — TEST INFRASTRUCTURE ONLY.

Follows /root/reference/src/rnnt/model.py:39-59 (``JointNet.forward``, training path) step by
step with stock torch CPU ops: unsqueeze (:40-41), repeat over U+1 and T (:47-50), concatenate
(:54) or add (:44), then ``Linear -> Tanh -> Linear`` (:14-18, :59).  Only ``tests/`` may import
this; the product (``rnnt_amd/joint.py``) never does.

PINNED to the reference's own code: tests/golden/make_ref_fixtures.py imports the reference's
``JointNet`` in the build container (its two absent, unrelated imports stubbed), runs it in float64 for
``concat`` and ``sum`` and writes inputs, parameters, logits and autograd gradients to
tests/golden/ref_joint_*.npz; tests/test_joint_oracle.py holds this restatement to those within 1e-12.
(The reference itself has no tests or fixtures for its joint network.)
"""
from __future__ import annotations

import torch


def joint_logits(enc_state: torch.Tensor, dec_state: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor,
                 w2: torch.Tensor, b2: torch.Tensor, joint: str = "concat", return_hidden: bool = False):
    """enc_state [B,T,He], dec_state [B,U+1,Hd] -> logits [B,T,U+1,V] (CPU tensors, any float dtype)."""
    e = enc_state.unsqueeze(2)
    d = dec_state.unsqueeze(1)
    if joint == "sum":
        x = e + d
    elif joint == "concat":
        x = torch.cat((e.repeat(1, 1, d.size(2), 1), d.repeat(1, e.size(1), 1, 1)), dim=-1)
    else:
        raise NotImplementedError
    hidden = torch.tanh(torch.nn.functional.linear(x, w1, b1))
    logits = torch.nn.functional.linear(hidden, w2, b2)
    return (logits, hidden) if return_hidden else logits
