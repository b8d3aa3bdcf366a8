"""Minibatch block construction on the GPU: tf_euler/python/dataflow
(`base_dataflow.py:22-51`, `neighbor_dataflow.py:22-110`, `sage_dataflow.py:
20-50`, `gcn_dataflow.py:24-47`, `relation_dataflow.py:24-75`) with the sampler,
the full-neighbour query and `tf.unique` replaced by the HIP kernels of this
package (sample_neighbor, get_full_neighbor, ID_UNIQUE in first-occurrence order).  Everything
stays in HBM; the only host round trip per hop is the unique count (the shape of
the next hop), exactly where TensorFlow has a dynamic shape.

    flow = SageDataFlow(graph, fanouts=[10, 5], metapath=[[0], [0]], max_id=N)
    df = flow(roots)            # DataFlow: blocks from the outermost hop inwards
    for block in df: block.n_id, block.res_n_id, block.edge_index, block.size
"""
import torch

from . import ops
from ._lib import EINVAL, EulerGpuError


class Block(object):
    """base_dataflow.py:22-28."""

    def __init__(self, n_id, res_n_id, e_id, edge_index, size):
        self.n_id = n_id
        self.res_n_id = res_n_id
        self.e_id = e_id
        self.edge_index = edge_index
        self.size = size


class DataFlow(object):
    """base_dataflow.py:31-51."""

    def __init__(self, n_id):
        self.n_id = n_id
        self.__last_n_id__ = n_id
        self.blocks = []

    def append(self, n_id, res_n_id, e_id, edge_index):
        size = [int(self.__last_n_id__.shape[0]), int(n_id.shape[0])]
        self.blocks.append(Block(n_id, res_n_id, e_id, edge_index, size))
        self.__last_n_id__ = n_id

    def __len__(self):
        return len(self.blocks)

    def __getitem__(self, idx):
        return self.blocks[::-1][idx]

    def __iter__(self):
        for block in self.blocks[::-1]:
            yield block


def _unique(ids):
    """tf.unique: (values in first-occurrence order, index of every input)."""
    uq, inv = ops.id_unique(ids)
    return uq, inv.to(torch.int64)


class UniqueDataFlow(object):
    """neighbor_dataflow.py:78-110 (UniqueDataFlow.produce_subgraph)."""

    def __init__(self, num_hops, add_self_loops=True):
        self.num_hops = num_hops
        self.add_self_loops = add_self_loops

    def get_neighbors(self, n_id):
        raise NotImplementedError()

    def produce_subgraph(self, n_id):
        n_id = n_id.reshape(-1)
        dev = n_id.device
        last_idx = torch.arange(n_id.numel(), device=dev)
        data_flow = DataFlow(n_id)
        n_neighbors, n_edge_src = self.get_neighbors(n_id)
        for i in range(self.num_hops):
            edge_src = n_edge_src[i]
            n_prev = n_id.numel()
            new_n_id, new_inv = _unique(torch.cat([n_neighbors[i], n_id]))
            res_n_id = new_inv[new_inv.numel() - n_prev:]
            if self.add_self_loops:
                edge_src = torch.cat([edge_src, last_idx])
                last_idx = torch.arange(new_n_id.numel(), device=dev)
            else:
                new_inv = new_inv[:new_inv.numel() - n_prev]
                last_idx = new_inv
            n_id = new_n_id
            edge_index = torch.stack([edge_src, new_inv], 0)
            data_flow.append(new_n_id, res_n_id, None, edge_index)
        return data_flow

    def __call__(self, n_id):
        return self.produce_subgraph(n_id)


class SageDataFlow(UniqueDataFlow):
    """sage_dataflow.py:20-50: every hop samples `count` neighbours of the
    nodes seen so far (a unique set), default_node = max_id + 1."""

    def __init__(self, graph, fanouts, metapath, add_self_loops=True, max_id=-1):
        super(SageDataFlow, self).__init__(len(metapath), add_self_loops)
        self.graph = graph
        self.fanouts = fanouts
        self.metapath = metapath
        self.max_id = max_id

    def produce_subgraph(self, n_id):
        """One enqueue for all hops (Graph.sage_blocks -> euler_gpu_sage_blocks):
        sampler, first-occurrence unique, res_n_id and edge_index of every hop run
        back to back on the stream; the host reads the layer sizes once at the end.
        `fused = False` (or hops whose edge-type lists differ in length) takes the
        op-by-op composition of the base class, one host round trip per hop."""
        lens = {len(m) for m in self.metapath}
        if not getattr(self, "fused", True) or len(lens) != 1:
            return super(SageDataFlow, self).produce_subgraph(n_id)
        n_id = n_id.reshape(-1)
        try:
            blocks, _cnt = self.graph.sage_blocks(n_id, self.metapath, self.fanouts,
                                                  default_node=self.max_id + 1,
                                                  add_self_loops=self.add_self_loops)
        except EulerGpuError as e:
            # the one-enqueue form has preconditions (worst-case layer sizes below 2^31, a
            # sampler the counted launch supports): EINVAL there means "not this way"
            if e.code != EINVAL:
                raise
            return super(SageDataFlow, self).produce_subgraph(n_id)
        data_flow = DataFlow(n_id)
        for new_n_id, res_n_id, _edge_src, _edge_dst, edge_index in blocks:
            data_flow.append(new_n_id, res_n_id, None, edge_index)
        return data_flow

    def produce_subgraphs(self, n_ids):
        """The list form: one DataFlow per minibatch of `n_ids` ([M, n] tensor or a list of M
        equally long id tensors), ALL of them built by one enqueue (Graph.sage_blocks_multi ->
        euler_gpu_sage_blocks_multi) - what a loop over produce_subgraph returns, bit for
        bit, for the price of one flow's launches.  Minibatches of different lengths, hops
        whose edge-type lists differ in length and `fused = False` take that loop."""
        if not torch.is_tensor(n_ids):
            n_ids = list(n_ids)
            same = len({int(x.numel()) for x in n_ids}) == 1
            if not same or not n_ids:
                return [self.produce_subgraph(x) for x in n_ids]
            n_ids = torch.stack([x.reshape(-1) for x in n_ids], 0)
        lens = {len(m) for m in self.metapath}
        if not getattr(self, "fused", True) or len(lens) != 1:
            return [self.produce_subgraph(x) for x in n_ids]
        try:
            per_mb = self.graph.sage_blocks_multi(n_ids, self.metapath, self.fanouts,
                                                  default_node=self.max_id + 1,
                                                  add_self_loops=self.add_self_loops)
        except EulerGpuError as e:
            if e.code != EINVAL:
                raise
            return [self.produce_subgraph(x) for x in n_ids]
        flows = []
        for b, (blocks, _cnt) in enumerate(per_mb):
            data_flow = DataFlow(n_ids[b].reshape(-1))
            for new_n_id, res_n_id, _edge_src, _edge_dst, edge_index in blocks:
                data_flow.append(new_n_id, res_n_id, None, edge_index)
            flows.append(data_flow)
        return flows

    def get_neighbors(self, n_id):
        neighbors, neighbor_src = [], []
        for hop_edge_types, count in zip(self.metapath, self.fanouts):
            n_id = n_id.reshape(-1)
            one_neighbor, _w, _t = self.graph.sample_neighbor(
                n_id, hop_edge_types, count, default_node=self.max_id + 1)
            new_n_id = one_neighbor.reshape(-1)
            neighbors.append(new_n_id)
            neighbor_src.append(torch.arange(n_id.numel(), device=n_id.device)
                                .repeat_interleave(count))
            n_id, _ = _unique(torch.cat([new_n_id, n_id]))
        return neighbors, neighbor_src


def _full_neighbor_coo(graph, n_id, edge_types):
    """tf_euler.get_full_neighbor as the dataflows read it: the neighbour ids in
    row order (`SparseTensor.values`), their row index (`indices[:, 0]`) and
    their edge types."""
    idx, ids, _w, t = graph.get_full_neighbor(n_id, edge_types)
    lens = (idx[:, 1] - idx[:, 0]).to(torch.int64)
    src = torch.repeat_interleave(torch.arange(n_id.numel(), device=n_id.device), lens)
    return ids.reshape(-1), src, t.reshape(-1)


class _FullBlocksMixin(object):
    """One enqueue for all hops of a full-neighbour flow (Graph.full_blocks ->
    euler_gpu_full_blocks): the row lengths are data, so every hop's edge list gets a
    capacity - an estimate from the graph's mean degree at first, then what the last
    minibatches needed with headroom; a minibatch that overflows it is redone op by op
    (one host round trip per hop, as TensorFlow's dynamic shapes) and raises the estimate."""

    _growth = None

    def _edge_caps(self, n):
        g = self.graph
        if self._growth is None:
            mean_deg = max(1.0, g.num_edges / max(1, g.num_nodes))
            self._growth = [4.0 * mean_deg] * len(self.metapath)
        caps, nodes = [], float(n)
        for gr in self._growth:
            e = int(nodes * gr) + 1024
            caps.append(e)
            # a layer cannot hold more distinct nodes than the graph has (+ the unknown ids
            # of the batch): estimates must not compound past that
            nodes = min(nodes + e, float(g.num_nodes + n))
        return caps

    def _full_blocks(self, n_id, self_loops, with_types):
        if not getattr(self, "fused", True) or len({len(m) for m in self.metapath}) != 1:
            return None
        n_id = n_id.reshape(-1)
        caps = self._edge_caps(n_id.numel())
        try:
            blocks, cnt = self.graph.full_blocks(n_id, self.metapath, caps, self_loops, with_types)
        except EulerGpuError as e:
            if e.code != EINVAL:
                raise
            return None
        L = len(self.metapath)
        # edges per node of the layer, as seen: the next estimate (with headroom)
        over = cnt[2 * L + 1] - 1 if blocks is None else -1     # the overflow word names the hop (h + 1)
        for h in range(L):
            if h == over:
                self._growth[h] = 4.0 * self._growth[h]           # only the hop that overflowed
            elif over < 0 or h < over:                            # later hops saw no input
                seen = cnt[L + 1 + h] / max(1, cnt[h])
                self._growth[h] = max(1.0, 0.5 * self._growth[h], 2.0 * seen)
        return blocks

    def _learn_from(self, data_flow, self_loops):
        """After an overflow the minibatch was redone op by op: its blocks hold the true sizes
        of every hop - the next minibatch's capacities come from them, not from a guess."""
        if self._growth is None:
            return
        for h, blk in enumerate(data_flow.blocks):
            n_prev = int(blk.size[0])
            edges = int(blk.edge_index.shape[1]) - (n_prev if self_loops else 0)
            self._growth[h] = max(self._growth[h], 1.0, 2.0 * edges / max(1, n_prev))


class GCNDataFlow(_FullBlocksMixin, UniqueDataFlow):
    """gcn_dataflow.py:24-47: every hop takes ALL neighbours (of the listed edge
    types) of the nodes seen so far."""

    def __init__(self, graph, metapath, add_self_loops=True):
        super(GCNDataFlow, self).__init__(len(metapath), add_self_loops)
        self.graph = graph
        self.metapath = metapath

    def produce_subgraph(self, n_id):
        blocks = self._full_blocks(n_id, self.add_self_loops, False)
        if blocks is None:
            data_flow = super(GCNDataFlow, self).produce_subgraph(n_id)
            self._learn_from(data_flow, self.add_self_loops)
            return data_flow
        data_flow = DataFlow(n_id.reshape(-1))
        for new_n_id, res_n_id, edge_src, edge_dst, _t in blocks:
            data_flow.append(new_n_id, res_n_id, None, torch.stack([edge_src, edge_dst], 0))
        return data_flow

    def get_neighbors(self, n_id):
        neighbors, neighbor_src = [], []
        for hop_edge_types in self.metapath:
            n_id = n_id.reshape(-1)
            new_n_id, src, _t = _full_neighbor_coo(self.graph, n_id, hop_edge_types)
            neighbors.append(new_n_id)
            neighbor_src.append(src)
            n_id, _ = _unique(torch.cat([new_n_id, n_id]))
        return neighbors, neighbor_src


class RelationDataFlow(_FullBlocksMixin):
    """relation_dataflow.py:24-75 (RGCN): full neighbours per hop, the edge TYPE
    of every edge as the block's e_id, no self loops."""

    def __init__(self, graph, metapath):
        self.graph = graph
        self.metapath = metapath

    def get_neighbors(self, n_id):
        neighbors, types, neighbor_src = [], [], []
        for hop_edge_types in self.metapath:
            n_id = n_id.reshape(-1)
            new_n_id, src, t = _full_neighbor_coo(self.graph, n_id, hop_edge_types)
            neighbors.append(new_n_id)
            types.append(t)
            neighbor_src.append(src)
            n_id, _ = _unique(torch.cat([new_n_id, n_id]))
        return neighbors, types, neighbor_src

    def produce_subgraph(self, n_id):
        n_id = n_id.reshape(-1)
        blocks = self._full_blocks(n_id, False, True)
        if blocks is not None:
            data_flow = DataFlow(n_id)
            for new_n_id, res_n_id, edge_src, edge_dst, e_type in blocks:
                data_flow.append(new_n_id, res_n_id, e_type, torch.stack([edge_src, edge_dst], 0))
            return data_flow
        data_flow = DataFlow(n_id)
        n_neighbors, types, n_edge_src = self.get_neighbors(n_id)
        for i in range(len(self.metapath)):
            n_prev = n_id.numel()
            new_n_id, new_inv = _unique(torch.cat([n_neighbors[i], n_id]))
            res_n_id = new_inv[new_inv.numel() - n_prev:]
            edge_dst = new_inv[:new_inv.numel() - n_prev]
            n_id = new_n_id
            edge_index = torch.stack([n_edge_src[i], edge_dst], 0)
            data_flow.append(new_n_id, res_n_id, types[i], edge_index)
        self._learn_from(data_flow, False)
        return data_flow

    def __call__(self, n_id):
        return self.produce_subgraph(n_id)


class NeighborDataFlow(object):
    """neighbor_dataflow.py:24-75: blocks over the hop results as they are (no
    dedup between hops)."""

    def __init__(self, num_hops, add_self_loops=True):
        self.num_hops = num_hops
        self.add_self_loops = add_self_loops

    def get_neighbors(self, n_id):
        raise NotImplementedError()

    def produce_subgraph(self, n_id):
        n_id = n_id.reshape(-1)
        dev = n_id.device
        last_idx = torch.arange(n_id.numel(), device=dev)
        data_flow = DataFlow(n_id)
        n_neighbors, n_edge_src = self.get_neighbors(n_id)
        for i in range(self.num_hops):
            edge_src = n_edge_src[i].to(torch.int64)
            n_prev = n_id.numel()
            new_n_id = torch.cat([n_neighbors[i], n_id])
            new_inv = torch.arange(new_n_id.numel(), device=dev)
            res_n_id = new_inv[new_inv.numel() - n_prev:]
            if self.add_self_loops:
                edge_src = torch.cat([edge_src, last_idx])
                last_idx = new_inv
            else:
                new_inv = new_inv[:new_inv.numel() - n_prev]
                last_idx = new_inv
            n_id = new_n_id
            edge_index = torch.stack([edge_src, new_inv], 0)
            data_flow.append(new_n_id, res_n_id, None, edge_index)
        return data_flow

    def __call__(self, n_id):
        return self.produce_subgraph(n_id)


class LayerwiseDataFlow(UniqueDataFlow):
    """layerwise_dataflow.py:26-62: every hop but the last draws
    sum(fanouts[:i+1]) nodes for the whole current node set with
    sample_neighbor_layerwise (one batch row) and keeps the (source, sampled
    node) pairs that are edges; the last hop takes full neighbours."""

    def __init__(self, graph, fanouts, metapath, add_self_loops=True):
        super(LayerwiseDataFlow, self).__init__(len(metapath), add_self_loops)
        self.graph = graph
        self.fanouts = fanouts
        self.metapath = metapath

    def get_neighbors(self, n_id):
        neighbors, neighbor_src = [], []
        total_fanout = 0
        for i in range(len(self.metapath)):
            if i == len(self.metapath) - 1:
                one_neighbor, one_indices, _t = _full_neighbor_coo(
                    self.graph, n_id.reshape(-1), self.metapath[i])
            else:
                total_fanout += self.fanouts[i]
                unique_neighbor, (ind, _val, _shape) = \
                    self.graph.sample_neighbor_layerwise(
                        n_id.reshape(1, -1), self.metapath[i], total_fanout)
                # the reference gathers through EVERY stored entry, the explicit
                # zero at (0, n-1, m-1) included (layerwise_dataflow.py:48-51)
                one_neighbor = unique_neighbor.reshape(-1)[ind[:, 2]]
                one_indices = ind[:, 1]
            neighbors.append(one_neighbor.reshape(-1))
            neighbor_src.append(one_indices)
            n_id, _ = _unique(torch.cat([one_neighbor.reshape(-1), n_id.reshape(-1)]))
        return neighbors, neighbor_src


class WholeDataFlow(NeighborDataFlow):
    """whole_dataflow.py:25-60: one adjacency among the batch's own nodes
    (sparse_get_adj(n_id, n_id)), reused by every hop.  Like the reference it
    reads columns 0 and 1 of the [1, n, n] sparse indices as (src, dst)."""

    def __init__(self, graph, metapath, add_self_loops=True):
        super(WholeDataFlow, self).__init__(len(metapath), add_self_loops)
        self.graph = graph
        self.neighbor_type = metapath[0]
        for n_type in metapath:
            if not n_type == self.neighbor_type:
                raise ValueError('Metapath should be the same in whole graph sampler.')

    def get_self_neighbors(self, n_id):
        n_id = n_id.reshape(-1)
        ind, _val, _shape = self.graph.sparse_get_adj(n_id, n_id, self.neighbor_type,
                                                      -1, -1)
        return ind[:, 0], ind[:, 1]

    def produce_subgraph(self, n_id):
        n_id = n_id.reshape(-1)
        inv = torch.arange(n_id.numel(), device=n_id.device)
        data_flow = DataFlow(n_id)
        edge_src, edge_dst = self.get_self_neighbors(n_id)
        if self.add_self_loops:
            edge_dst = torch.cat([edge_dst, inv])
            edge_src = torch.cat([edge_src, inv])
        edge_index = torch.stack([edge_src, edge_dst], 0)
        for _ in range(self.num_hops):
            data_flow.append(n_id, inv, None, edge_index)
        return data_flow
