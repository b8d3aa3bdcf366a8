"""tf_euler/python/euler_ops/neighbor_ops.py (hot-path subset)."""
from . import base, type_ops

__all__ = ["sample_neighbor", "sample_fanout", "get_full_neighbor",
           "get_sorted_full_neighbor", "get_top_k_neighbor", "to_sparse",
           "sample_fanout_with_feature"]


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


def get_sorted_full_neighbor(nodes, edge_types, condition=''):
    """Full neighbours ordered by id (neighbor_ops.py:96-114), GQL layout
    (idx [n,2] int32, ids, weights, types); see to_sparse() for the
    SparseTensor triple the TF op returns."""
    if condition:
        raise NotImplementedError("index conditions are out of scope (SURVEY §2)")
    edge_types = type_ops.get_edge_type_id(edge_types)
    return base.get_default_graph().get_sorted_full_neighbor(nodes, edge_types)


def get_top_k_neighbor(nodes, edge_types, k, default_node=-1, condition=''):
    """(neighbors [n,k] int64, weights f32, types int32): the k heaviest
    neighbours, default_node / 0.0 / -1 padded (neighbor_ops.py:44-46)."""
    if condition:
        raise NotImplementedError("index conditions are out of scope (SURVEY §2)")
    edge_types = type_ops.get_edge_type_id(edge_types)
    return base.get_default_graph().get_top_k_neighbor(nodes, edge_types, k,
                                                       default_node)


def to_sparse(idx, values):
    """(indices [nnz,2] int64, values, dense_shape [2]) of the SparseTensor the
    TF full-neighbour ops build from the GQL layout
    (tf_euler/kernels/get_full_neighbor_op.cc:95-110): entry j of row i sits at
    (i, j - start_i); dense_shape = [n, longest row]."""
    import torch
    idx = idx.to(torch.int64)
    lens = idx[:, 1] - idx[:, 0]
    n = idx.shape[0]
    rows = torch.repeat_interleave(torch.arange(n, device=idx.device), lens)
    cols = torch.arange(values.shape[0], device=idx.device) - \
        torch.repeat_interleave(idx[:, 0], lens)
    width = int(lens.max().item()) if n else 0
    return (torch.stack([rows, cols], dim=1), values,
            torch.tensor([n, width], dtype=torch.int64))


def sample_fanout_with_feature(nodes, edge_types, count, default_node,
                               dense_feature_names, dense_dimensions,
                               sparse_feature_names=(), sparse_default_values=()):
    """sample_fanout + the dense features of every layer's nodes in one call
    (neighbor_ops.py:49-70 over tf_euler/kernels/sample_fanout_with_feature_op.cc):
    returns (neighbors, weights, types, dense_features, sparse_features) with
    dense_features[layer * len(names) + j] = feature j of layer `layer`'s nodes
    (layer 0 = the roots), as the op lays its outputs out (:135-178,233).
    Sparse (uint64) features are out of scope: the list must be empty."""
    if len(sparse_feature_names):
        raise NotImplementedError("sparse features are out of scope (SURVEY §8f)")
    g = base.get_default_graph()
    ets = [type_ops.get_edge_type_id(et) for et in edge_types]
    neighbors, weights, types = g.sample_fanout(nodes, ets, count, default_node)
    fids = [int(str(f)) for f in dense_feature_names]
    dense = []
    for layer_nodes in neighbors:
        dense.extend(g.get_dense_feature(layer_nodes, fids, list(dense_dimensions)))
    return neighbors, weights, types, dense, []
