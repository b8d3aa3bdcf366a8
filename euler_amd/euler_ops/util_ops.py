"""tf_euler.python.euler_ops.util_ops (module path kept for ported code): sparse_gather
(tf_euler/kernels/sparse_gather_op.cc); the op lives in euler_amd.ops."""
from ..ops import sparse_gather  # noqa: F401
