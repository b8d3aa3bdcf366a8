"""tf_euler.python.euler_ops.type_ops (module path kept for ported code); the functions live in node_ops."""
from .node_ops import get_node_type_id, get_edge_type_id  # noqa: F401
