"""tf_euler/python/euler_ops/neighbor_ops.py (hot-path subset)."""
from . import base
from . import node_ops as type_ops

__all__ = ["sample_neighbor", "sample_fanout", "sample_fanout_multi", "get_full_neighbor",
           "get_sorted_full_neighbor", "get_top_k_neighbor", "to_sparse",
           "sample_fanout_with_feature", "sparse_get_adj",
           "sample_neighbor_layerwise", "sample_fanout_layerwise_each_node",
           "sample_fanout_layerwise", "get_multi_hop_neighbor"]


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


def sample_fanout_multi(batches, edge_types, counts, default_node=-1):
    """sample_fanout of M minibatches ([M, B] roots, or a list of M equally long root
    tensors) in ONE enqueue: a list of M (neighbors_list, weights_list, types_list), each
    equal bit for bit to what M consecutive sample_fanout calls return (draws are keyed by
    (call id, node id)).  Not in the reference: its callers issue one op per minibatch of a
    few hundred roots (dataflow/sage_dataflow.py:35-50) - this is that loop as one launch."""
    edge_types = [type_ops.get_edge_type_id(et) for et in edge_types]
    return base.get_default_graph().sample_fanout_multi(batches, edge_types, counts,
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
    sparse_features[layer * len(sparse names) + j] likewise, each a
    SparseTensor triple (indices, values, dense_shape) with the per-feature
    default values (tf_euler/kernels/sample_fanout_with_feature_op.cc:180-232)."""
    g = base.get_default_graph()
    ets = [type_ops.get_edge_type_id(et) for et in edge_types]
    fids = [int(str(f)) for f in dense_feature_names]
    sfids = [int(str(f)) for f in sparse_feature_names]
    sdef = list(sparse_default_values) if len(sparse_default_values) else [0] * len(sfids)
    # the fanout and every layer's dense features: one enqueue, no host round trip
    # (euler_gpu_sample_fanout_with_feature); the sparse features need their sizes on the
    # host, one query per layer
    neighbors, weights, types, dense = g.sample_fanout_with_feature(
        nodes, ets, count, default_node, fids, list(dense_dimensions))
    sparse = []
    for layer_nodes in neighbors:
        if sfids:
            sparse.extend(g.get_sparse_feature(layer_nodes, sfids, sdef))
    return neighbors, weights, types, dense, sparse


def sparse_get_adj(nodes, nb_nodes, edge_types, n=-1, m=-1):
    """The SparseTensor triple (indices [nnz,3] int64, values int64,
    dense_shape) of the [batch, n, m] 0/1 adjacency between `nodes` and
    `nb_nodes` over the listed edge types (neighbor_ops.py:33-36 over
    tf_euler/kernels/sparse_get_adj_op.cc; n / m = -1: one batch row)."""
    edge_types = type_ops.get_edge_type_id(edge_types)
    return base.get_default_graph().sparse_get_adj(nodes, nb_nodes, edge_types, n, m)


def sample_neighbor_layerwise(nodes, edge_types, count, default_node=-1,
                              weight_func=''):
    """nodes [batch, n] -> (neighbors [batch, count] int64, (indices, values,
    dense_shape) of the [batch, n, count] adjacency) (neighbor_ops.py:72-77).
    Only weight_func == '' (the sampleLNB form without a weight function)."""
    edge_types = type_ops.get_edge_type_id(edge_types)
    return base.get_default_graph().sample_neighbor_layerwise(
        nodes, edge_types, count, default_node, weight_func)


def sample_fanout_layerwise_each_node(nodes, edge_types, counts, default_node=-1):
    """neighbor_ops.py:161-186: hop 1 = sample_neighbor, later hops =
    sample_neighbor_layerwise over each root's own previous layer."""
    import torch
    g = base.get_default_graph()
    ets = [type_ops.get_edge_type_id(et) for et in edge_types]
    neighbors_list = [torch.as_tensor(nodes).reshape(-1)]
    adj_list = []
    last_count = None
    for hop_edge_types, count in zip(ets, counts):
        if len(neighbors_list) == 1:
            neighbors, _, _ = g.sample_neighbor(neighbors_list[-1], hop_edge_types,
                                                count, default_node)
        else:
            neighbors, adj = g.sample_neighbor_layerwise(
                neighbors_list[-1].reshape(-1, last_count), hop_edge_types, count,
                default_node)
            adj_list.append(adj)
        neighbors_list.append(neighbors.reshape(-1))
        last_count = count
    return neighbors_list, adj_list


def sample_fanout_layerwise(nodes, edge_types, counts, default_node=-1,
                            weight_func=''):
    """neighbor_ops.py:189-206: every hop samples `count` nodes for the whole
    previous layer (one batch row)."""
    import torch
    g = base.get_default_graph()
    ets = [type_ops.get_edge_type_id(et) for et in edge_types]
    neighbors_list = [torch.as_tensor(nodes).reshape(-1)]
    adj_list = []
    last_count = neighbors_list[0].numel()
    for hop_edge_types, count in zip(ets, counts):
        neighbors, adj = g.sample_neighbor_layerwise(
            neighbors_list[-1].reshape(-1, last_count), hop_edge_types, count,
            default_node, weight_func)
        neighbors_list.append(neighbors.reshape(-1))
        adj_list.append(adj)
        last_count = count
    return neighbors_list, adj_list


def get_multi_hop_neighbor(nodes, edge_types):
    """neighbor_ops.py:209-243: per hop the distinct full neighbours of the
    previous node set (first-occurrence order, tf.unique) and the weighted
    adjacency between the two sets as (indices [nnz,2] int64 sorted row-major,
    values f32, dense_shape) - tf.sparse_reorder'ed like the reference."""
    import torch
    from .. import ops
    g = base.get_default_graph()
    ets = [type_ops.get_edge_type_id(et) for et in edge_types]
    nodes = torch.as_tensor(nodes).reshape(-1).to(g.device)
    nodes_list, adj_list = [nodes], []
    for hop_edge_types in ets:
        idx, ids, w, _t = g.get_full_neighbor(nodes, hop_edge_types)
        lens = (idx[:, 1] - idx[:, 0]).to(torch.int64)
        rows = torch.repeat_interleave(
            torch.arange(nodes.numel(), device=nodes.device), lens)
        next_nodes, next_idx = ops.id_unique(ids)
        next_idx = next_idx.to(torch.int64)
        # tf.sparse_reorder: canonical row-major order (stable for equal cells)
        order = torch.argsort(rows * max(int(next_nodes.numel()), 1) + next_idx,
                              stable=True)
        indices = torch.stack([rows[order], next_idx[order]], 1)
        adj_list.append((indices, w[order],
                         [int(nodes.numel()), int(next_nodes.numel())]))
        nodes_list.append(next_nodes)
        nodes = next_nodes
    return nodes_list, adj_list
