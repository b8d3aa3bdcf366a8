"""tf_euler/python/euler_ops/feature_ops.py: dense and sparse node features."""
from . import base

__all__ = ["get_dense_feature", "get_sparse_feature"]


def get_dense_feature(nodes, feature_names, dimensions, thread_num=1):
    """Fetch dense (float) features of nodes: a list of [n, dim] float32
    tensors, one per feature id in `feature_names` (ints, or their string
    forms as tf_euler passes them).  thread_num is accepted for signature
    compatibility; the fetch is one kernel per feature."""
    fids = [int(str(f)) for f in feature_names]
    return base.get_default_graph().get_dense_feature(nodes, fids, list(dimensions))


def get_sparse_feature(nodes, feature_names, default_values=None, thread_num=1):
    """Fetch sparse (uint64) features of nodes (feature_ops.py:57-73): one
    SparseTensor triple (indices [nnz, 2], values [nnz], dense_shape) per
    feature id; nodes that store nothing get the single entry (row, 0) =
    default value (0).  thread_num is accepted for signature compatibility."""
    fids = [int(str(f)) for f in feature_names]
    return base.get_default_graph().get_sparse_feature(nodes, fids, default_values)
