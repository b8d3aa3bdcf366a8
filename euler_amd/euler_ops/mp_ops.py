"""tf_euler/python/euler_ops/mp_ops.py: message-passing operators with the
gradients the reference registers (mp_ops.py:39-62)."""
from ..ops import (gather, scatter_add, scatter_max, scatter_mean,  # noqa: F401
                   scatter_softmax, scatter_)

__all__ = ["gather", "scatter_add", "scatter_max", "scatter_mean",
           "scatter_softmax", "scatter_"]
