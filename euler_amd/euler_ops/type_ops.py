"""tf_euler/python/euler_ops/type_ops.py: type names -> ids.  Names are the
strings of euler.meta; integers pass through (type_ops.py:31-35)."""
from . import base

__all__ = ["ALL_NODE_TYPE", "get_node_type_id", "get_edge_type_id"]

ALL_NODE_TYPE = -1


def _get_type_id(table, type_id_or_names):
    if isinstance(type_id_or_names, (str, bytes, int)):
        type_id_or_names = [type_id_or_names]
    out = []
    for t in type_id_or_names:
        if isinstance(t, bytes):
            t = t.decode()
        if isinstance(t, str):
            if t in table:
                t = table[t]
            elif t.lstrip('-').isdigit():
                t = int(t)
            else:
                raise KeyError("unknown type name %r" % t)
        out.append(int(t))
    return out


def get_node_type_id(type_id_or_names):
    return _get_type_id(base.get_default_graph().node_type_names, type_id_or_names)


def get_edge_type_id(type_id_or_names):
    return _get_type_id(base.get_default_graph().edge_type_names, type_id_or_names)
