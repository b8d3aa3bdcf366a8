"""tf_euler.python.euler_ops.walk_ops (module path kept for ported code); the functions live in node_ops."""
from .node_ops import random_walk, gen_pair  # noqa: F401
