"""tf_euler.python.euler_ops.util_ops (module path kept for ported code): inflate_idx
(tf_euler/kernels/inflate_idx_op.cc) and sparse_gather (tf_euler/kernels/sparse_gather_op.cc);
the ops live in euler_amd.ops."""
from ..ops import inflate_idx, sparse_gather  # noqa: F401
