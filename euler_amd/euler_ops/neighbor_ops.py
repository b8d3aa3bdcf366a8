"""tf_euler/python/euler_ops/neighbor_ops.py (hot-path subset)."""
from . import base, type_ops

__all__ = ["sample_neighbor", "sample_fanout", "get_full_neighbor"]


def sample_neighbor(nodes, edge_types, count, default_node=-1, condition=''):
    """(neighbors [n,count] int64, weights f32, types int32)
    (neighbor_ops.py:39-41)."""
    if condition:
        raise NotImplementedError("index conditions are out of scope (SURVEY §2)")
    edge_types = type_ops.get_edge_type_id(edge_types)
    return base.get_default_graph().sample_neighbor(nodes, edge_types, count,
                                                    default_node)


def sample_fanout(nodes, edge_types, counts, default_node=-1):
    """(neighbors_list, weights_list, types_list) with shapes [n], [n*c1],
    [n*c1*c2] ... (neighbor_ops.py:122-158)."""
    edge_types = [type_ops.get_edge_type_id(et) for et in edge_types]
    return base.get_default_graph().sample_fanout(nodes, edge_types, counts,
                                                  default_node)


def get_full_neighbor(nodes, edge_types, condition=''):
    """GQL-layout full neighbours: (idx [n,2] int32, ids, weights, types)."""
    if condition:
        raise NotImplementedError("index conditions are out of scope (SURVEY §2)")
    edge_types = type_ops.get_edge_type_id(edge_types)
    return base.get_default_graph().get_full_neighbor(nodes, edge_types)
