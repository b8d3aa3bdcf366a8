"""tf_euler.python.euler_ops.feature_ops (module path kept for ported code); the functions live in node_ops.
Edge and binary features are not loaded by this backend (out of scope)."""
from .node_ops import get_dense_feature, get_sparse_feature  # noqa: F401
