"""tf_euler/python/euler_ops/sample_ops.py (hot-path subset)."""
from . import base, type_ops

__all__ = ["sample_node", "sample_node_with_src", "get_node_type"]


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


def get_node_type(nodes):
    """int32 node types, INT32_MIN for unknown ids (base._LIB_OP.get_node_type,
    tf_euler/kernels/get_node_type_op.cc:33-57)."""
    return base.get_default_graph().get_node_type(nodes)


def sample_node_with_src(src_nodes, count):
    """For every src node, `count` nodes of the same node type:
    [len(src_nodes), count] int64 (sample_ops.py:75-87)."""
    g = base.get_default_graph()
    return g.sample_n_with_types(int(count), g.get_node_type(src_nodes))
