"""The rest of tf_euler.python.euler_ops' surface for the hot path, in one module (the
reference spreads it over type_ops / sample_ops / walk_ops / feature_ops / mp_ops.py; the
neighbour operators are in neighbor_ops.py): type names -> ids, node sampling, walks and
pair generation, dense / sparse features, and the message-passing operators with the
gradients the reference registers (mp_ops.py:39-62) re-exported from euler_amd.ops."""
from .. import ops
from ..ops import (gather, scatter_add, scatter_max, scatter_mean,  # noqa: F401
                   scatter_softmax, scatter_)
from . import base

__all__ = ["ALL_NODE_TYPE", "get_node_type_id", "get_edge_type_id",
           "sample_node", "sample_node_with_src", "get_node_type",
           "random_walk", "gen_pair", "get_dense_feature", "get_sparse_feature",
           "gather", "scatter_add", "scatter_max", "scatter_mean", "scatter_softmax", "scatter_"]

# ---- type names -> ids (type_ops.py:31-35): names are the strings of euler.meta, integers
# pass through
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


# ---- node sampling (sample_ops.py, hot-path subset)
def sample_node(count, node_type, condition=''):
    """[count] int64 node ids sampled by node weight (sample_ops.py:42-59);
    node_type '-1' / -1 = all types."""
    if condition:
        raise NotImplementedError("index conditions are out of scope (SURVEY §2)")
    if node_type == '-1' or node_type == -1:
        types = -1
    else:
        types = get_node_type_id(node_type)[0]
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


# ---- walks (walk_ops.py)
gen_pair = ops.gen_pair


def random_walk(nodes, edge_types, p=1.0, q=1.0, default_node=-1):
    """nodes [n] -> paths [n, len(edge_types)+1] (walk_ops.py:29-43)."""
    edge_types = [get_edge_type_id(et) for et in edge_types]
    return base.get_default_graph().random_walk(nodes, edge_types, p, q,
                                                default_node)


# ---- features (feature_ops.py)
def get_dense_feature(nodes, feature_names, dimensions, thread_num=1):
    """Fetch dense (float) features of nodes: a list of [n, dim] float32
    tensors, one per feature id in `feature_names` (ints, or their string
    forms as tf_euler passes them).  thread_num is accepted for signature
    compatibility; the fetch is one kernel per feature."""
    fids = [int(str(f)) for f in feature_names]
    return base.get_default_graph().get_dense_feature(nodes, fids, list(dimensions))


def get_sparse_feature(nodes, feature_names, default_values=None, thread_num=1):
    """Fetch sparse (uint64) features of nodes (feature_ops.py:57-73): one
    SparseTensor triple (indices [nnz, 2], values [nnz], dense_shape) per
    feature id; nodes that store nothing get the single entry (row, 0) =
    default value (0).  thread_num is accepted for signature compatibility."""
    fids = [int(str(f)) for f in feature_names]
    return base.get_default_graph().get_sparse_feature(nodes, fids, default_values)
