"""tf_euler/python/euler_ops/sample_ops.py (hot-path subset)."""
from . import base, type_ops

__all__ = ["sample_node"]


def sample_node(count, node_type, condition=''):
    """[count] int64 node ids sampled by node weight (sample_ops.py:42-59);
    node_type '-1' / -1 = all types."""
    if condition:
        raise NotImplementedError("index conditions are out of scope (SURVEY §2)")
    if node_type == '-1' or node_type == -1:
        types = -1
    else:
        types = type_ops.get_node_type_id(node_type)[0]
    return base.get_default_graph().sample_node(int(count), types)
