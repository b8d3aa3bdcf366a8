"""tf_euler.python.euler_ops.mp_ops (module path kept for ported code); the ops live in euler_amd.ops."""
from ..ops import (gather, scatter_add, scatter_max, scatter_mean,  # noqa: F401
                   scatter_softmax, scatter_)
