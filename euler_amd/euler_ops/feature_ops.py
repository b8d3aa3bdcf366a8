"""tf_euler/python/euler_ops/feature_ops.py: dense node features."""
from . import base

__all__ = ["get_dense_feature"]


def get_dense_feature(nodes, feature_names, dimensions, thread_num=1):
    """Fetch dense (float) features of nodes: a list of [n, dim] float32
    tensors, one per feature id in `feature_names` (ints, or their string
    forms as tf_euler passes them).  thread_num is accepted for signature
    compatibility; the fetch is one kernel per feature."""
    fids = [int(str(f)) for f in feature_names]
    return base.get_default_graph().get_dense_feature(nodes, fids, list(dimensions))
