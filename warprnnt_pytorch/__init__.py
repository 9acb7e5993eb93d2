"""Import shim: lets the reference's ``from warprnnt_pytorch import RNNTLoss``
(/root/reference/src/rnnt/model.py:5) resolve to the MI355X implementation unchanged."""
from rnnt_amd.loss import RNNTLoss, rnnt_loss  # noqa: F401

__all__ = ["RNNTLoss", "rnnt_loss"]
