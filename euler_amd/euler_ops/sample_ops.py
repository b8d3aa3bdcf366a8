"""tf_euler.python.euler_ops.sample_ops (module path kept for ported code); the functions live in node_ops.
sample_edge / get_graph_by_label need Edge records / graph labels, which this backend does not load."""
from .node_ops import sample_node, sample_node_with_src, get_node_type  # noqa: F401
