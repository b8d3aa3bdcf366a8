"""Operator surface of tf_euler.python.euler_ops for the hot path."""
from .base import *          # noqa: F401,F403
from .node_ops import *      # noqa: F401,F403
from .neighbor_ops import *  # noqa: F401,F403
