"""rnnt_amd — MI355X-native RNN-Transducer loss (drop-in for ``warprnnt_pytorch.RNNTLoss``).

Scope: the one hot path the reference reaches at /root/reference/src/rnnt/model.py:116 plus the
batch-sharding harness that replaces its ``DataParallel`` use (bin/train.py:241-244).
``JointNet`` / ``joint_hidden`` are the first widening next to that path,
``greedy_decode`` the decode inner step (rank 4), ``GuardedStep`` the loss's consumer (rank 3),
``FusedJointLoss`` the joint's output projection fused with the loss's first pass (rank 1).
See DESIGN.md.  Importing this package does not load the HIP library; the first loss call does,
and raises if ``rnnt_amd/lib/librnnt_mi355.so`` or the compiled binding ``librnnt_amd_torch.so`` is missing
(no fallback exists).
"""
from .loss import RNNTLoss, rnnt_loss, certify_inputs, set_binding, get_binding  # noqa: F401
from .decode import greedy_decode, greedy_search_batch, row_argmax, row_max_logp  # noqa: F401
from .joint import JointNet, joint_hidden  # noqa: F401
from .parallel import (ShardedRNNTLoss, all_reduce_gradients, balanced_assignment, shard_bounds,  # noqa: F401
                       shard_batch)
from .train_step import GuardedStep  # noqa: F401
from .fused import FusedJointLoss, count_lattice_nodes, fused_joint_loss  # noqa: F401
from .compact import compact_rowmap, rnnt_loss_compact  # noqa: F401

__all__ = ["RNNTLoss", "rnnt_loss", "certify_inputs", "ShardedRNNTLoss", "shard_bounds", "shard_batch",
           "balanced_assignment", "all_reduce_gradients", "JointNet", "joint_hidden", "greedy_decode", "greedy_search_batch", "row_argmax",
           "row_max_logp", "GuardedStep", "FusedJointLoss", "fused_joint_loss", "count_lattice_nodes", "compact_rowmap", "rnnt_loss_compact", "set_binding", "get_binding"]
__version__ = "0.2.0"
