"""tf_euler/python/euler_ops/walk_ops.py."""
from .. import ops
from . import base, type_ops

__all__ = ["random_walk", "gen_pair"]

gen_pair = ops.gen_pair


def random_walk(nodes, edge_types, p=1.0, q=1.0, default_node=-1):
    """nodes [n] -> paths [n, len(edge_types)+1] (walk_ops.py:29-43)."""
    edge_types = [type_ops.get_edge_type_id(et) for et in edge_types]
    return base.get_default_graph().random_walk(nodes, edge_types, p, q,
                                                default_node)
