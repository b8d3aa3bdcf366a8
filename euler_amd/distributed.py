"""Multi-GPU sampling: one process per GPU, graph hash-partitioned by
owner(id) = (id % partitions) % shards (core/kernels/id_split_op.cc:46-49),
one exchange step per hop.

What this replaces in the reference's distributed mode:
  ID_SPLIT -> REMOTE (gRPC Execute on the shard server) -> IDX_MERGE/DATA_MERGE
becomes
  bucket_by_owner kernel -> all-to-all(ids) over RCCL/xGMI -> local HIP
  sample_neighbor on the owned rows -> all-to-all(results) -> merge_rows kernel.
Because the RNG is addressed by (seed, call_id, node id, draw) the sharded
result is bit-identical to the single-GPU result.

The local sampler is injected (`local_sample`), so the host logic - bucketing,
split sizes, the two exchanges, the inverse permutation - is exercised on CPU
with the gloo backend and a test double; on GPUs it is the HIP kernel path.
"""
import os

import torch
import torch.distributed as dist


def owner_of(ids, partitions, shards):
    """IDSplit::GetShardId on an int64 tensor holding uint64 bit patterns."""
    if ids.dtype != torch.int64:
        ids = ids.to(torch.int64)
    # unsigned modulo of the 64-bit pattern: fix up negative (>= 2^63) values
    m = torch.remainder(ids, partitions)
    neg = ids < 0
    if bool(neg.any()):
        # (ids + 2^64) % p = (ids % p + 2^64 % p) % p
        m = torch.where(neg, torch.remainder(m + (2 ** 64) % partitions, partitions), m)
    return torch.remainder(m, shards)


def sparse_from_core(idx, vals, default_value):
    """The SparseTensorBuilder rules of the TF GetSparseFeature kernel
    (tf_euler/kernels/get_sparse_feature_op.cc:96-107) over a GQL `values()`
    result (idx [n, 2] offsets, packed values): node j with no value gives the
    single entry (j, 0) = default, else (j, k) = value k.  Returns (indices
    [nnz, 2] int64, values [nnz] int64, dense_shape)."""
    idx = idx.reshape(-1, 2).to(torch.int64)
    dev = idx.device
    n = idx.shape[0]
    if n == 0:
        return (torch.zeros((0, 2), dtype=torch.int64, device=dev),
                torch.zeros(0, dtype=torch.int64, device=dev), [0, 0])
    lens = idx[:, 1] - idx[:, 0]
    emit = torch.clamp(lens, min=1)
    start = torch.cumsum(emit, 0) - emit
    rows = torch.repeat_interleave(torch.arange(n, device=dev), emit)
    col = torch.arange(int(emit.sum()), device=dev) - start[rows]
    has = lens[rows] > 0
    values = torch.full((rows.numel(),), int(default_value), dtype=torch.int64, device=dev)
    if vals.numel():
        src = (idx[rows, 0] + col).clamp(max=vals.numel() - 1)
        values = torch.where(has, vals.to(torch.int64)[src], values)
    return torch.stack([rows, col], 1), values, [n, int(emit.max())]


def adj_from_rows(idx, ids, nb_nodes, batch, n, m):
    """The TF SparseGetAdj triple (tf_euler/kernels/sparse_get_adj_op.cc:92-124)
    from the full-neighbour rows of the batch*n sources (idx [batch*n, 2], ids) and
    the candidates nb_nodes [batch*m]: (b, j, c) = 1 where candidate c of batch row
    b is in the row of source (b, j), plus the explicit 0 at (b, n-1, m-1).  The
    membership test is an exact join: ids and candidates are renumbered together
    (torch.unique), a pair is the key source * U + compact id."""
    dev = idx.device
    R = batch * n
    if R == 0 or m == 0:
        return (torch.zeros((0, 3), dtype=torch.int64, device=dev),
                torch.zeros(0, dtype=torch.int64, device=dev), [0, 0, 0])
    idx = idx.reshape(-1, 2).to(torch.int64)
    ids = ids.reshape(-1).to(torch.int64)
    lens = idx[:, 1] - idx[:, 0]
    src = torch.repeat_interleave(torch.arange(R, device=dev), lens)
    cand = nb_nodes.reshape(-1).to(torch.int64)[:batch * m]
    uq, inv = torch.unique(torch.cat([ids, cand]), return_inverse=True)
    U = int(uq.numel())
    ent_key = src * U + inv[:ids.numel()]
    cand_c = inv[ids.numel():].reshape(batch, m)
    r = torch.arange(R, device=dev)
    pair_key = r[:, None] * U + cand_c[r // n]
    mask = torch.isin(pair_key, ent_key).reshape(batch, n, m)
    emit = mask.clone()
    emit[:, n - 1, m - 1] = True
    return torch.nonzero(emit), mask[emit].to(torch.int64), [batch, n, m]


class ShardedSampler:
    """Neighbor sampling over a graph sharded across the ranks of `group`.

    local_sample(roots, edge_types, count, default_node, call_id)
        -> (ids [m,count] int64, weights f32, types i32, row_mask [m] uint8)
        TF-layout rows for roots this rank owns;
    split_fn(ids, partitions, shards) -> (shard_off list, shard_ids, merge_idx)
        stable bucket by owner (ID_SPLIT);
    merge_fn(rows, merge_idx) -> out with out[merge_idx[j]] = rows[j]
        (IDX_MERGE / DATA_MERGE for fixed-size rows, int32 rows);
    unique_fn(ids) -> (unique ids, gather_idx) (ID_UNIQUE);
    gather_fn(rows, gather_idx) -> rows[gather_idx] (DATA_GATHER, int32 rows).
    On GPUs all five are HIP kernels (gpu_sharded_sampler); the CPU tests inject
    oracle-backed doubles.
    """

    def __init__(self, local_sample, split_fn, merge_fn, partitions=None,
                 group=None, unique_fn=None, gather_fn=None, dedup_split_fn=None,
                 expand_fn=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        # a lone rank's ids never leave it, so its hops skip the exchanges; True makes it send to
        # itself what N ranks send to one another (EULER_GPU_SELF_EXCHANGE=1: the tests' way to
        # execute the RCCL all-to-all on a one-GPU box - never a production setting)
        self.force_exchange = os.environ.get("EULER_GPU_SELF_EXCHANGE") == "1"
        self.partitions = partitions or self.world
        self.local_sample = local_sample
        self.split_fn = split_fn
        self.merge_fn = merge_fn
        self.unique_fn = unique_fn
        self.gather_fn = gather_fn
        # fused front end / back end (one call each): dedup_split_fn(ids,
        # partitions, shards) -> (shard_off, distinct ids bucketed by owner,
        # pos [n]); expand_fn(pos, ids, w, t, mask, count) -> rows per position
        self.dedup_split_fn = dedup_split_fn
        self.expand_fn = expand_fn
        # wire format hooks of the fused path (HIP kernels on GPUs): pack_fn(ids,
        # w, t, mask, count, single_type) -> int32 rows; expand_fn(pos, rows,
        # count, single_type) -> outputs
        self.pack_fn = None
        # local_sample_packed(owned, edge_types, count, default_node, call_id) ->
        # wire rows (sampling and packing in one kernel)
        self.local_sample_packed = None
        # local_sample_sets_packed(owned ids, type_sets, count, default_node, call_id) -> wire rows per set
        self.local_sample_sets_packed = None
        # two-phase front end: front_begin_fn(ids, partitions, shards, root_mask,
        # root_group) enqueues it and returns a token, front_end_fn(token) waits
        # for the bucket sizes -> (shard_off, shard_ids, pos)
        self.front_begin_fn = None
        self.front_end_fn = None
        # counts_fn(send_counts list[world]) -> recv_counts list[world]
        self.counts_fn = None
        # optional hooks for get_dense_feature / sample_node (see those methods)
        self.local_feature = None
        self.row_gather_fn = None
        self.local_sample_node = None
        self.node_weight_sum = None
        self.node_split_fn = None
        self.local_full_neighbor = None
        self.local_sparse_feature = None
        # layerwise sampling (see sample_neighbor_layerwise)
        self.local_edge_sum_weight = None
        self.sample_root_fn = None
        self.local_sample_layer = None
        self.local_layer_fn = None
        self.idx_gather_fn = None
        self.data_gather_fn = None
        self.device = torch.device("cpu")
        # transport: RCCL moves device tensors directly; a gloo group (ranks that
        # share one GPU in the world-2-on-one-GPU tests, or CPU test doubles)
        # carries device tensors staged through the host
        self.host_staged = dist.get_backend(group) == "gloo"
        # wire statistics (rows this rank handed to / took from its peers, self
        # exchanges excluded): bench.py reports bytes exchanged per step
        self.bytes_sent = 0
        self.bytes_received = 0
        self.exchanges = 0
        # not None: every collective this sampler issues is appended as (kind, peers | row bytes) -
        # all ranks must produce the SAME sequence whatever their batches hold (tests)
        self.collective_log = None
        # call ids: every sampling call without an explicit call_id takes fresh
        # ones from this counter (by hops for a fanout, by steps for a walk), like
        # Graph._take_call_ids - identically on all ranks, because every rank makes
        # the same sequence of collective calls
        self._call_id = 0

    def set_call_id(self, call_id):
        """Restart the implicit call-id sequence (reproducible runs); call it with
        the same value on every rank."""
        self._call_id = int(call_id) & 0xFFFFFFFF

    def _take_call_ids(self, n, call_id=None):
        if call_id is not None:
            return int(call_id) & 0xFFFFFFFF
        c = self._call_id
        self._call_id = (self._call_id + int(n)) & 0xFFFFFFFF
        return c

    # -------------------------------------------------------------- helpers
    def _exchange_counts(self, send_counts, device):
        """Every peer learns how many rows it gets from this rank: counts_fn (a
        host-side mailbox between the ranks of one node, see ShmCounts) when the
        sampler has one, else an all-to-all of the counts on the GPU followed by
        a device sync."""
        if self.world == 1 and not self.force_exchange:
            return [int(c) for c in send_counts]
        if self.collective_log is not None:
            self.collective_log.append(("counts", len(send_counts)))
        if self.counts_fn is not None:
            return self.counts_fn(send_counts)
        sc = torch.tensor(send_counts, dtype=torch.int64,
                          device="cpu" if self.host_staged else device)
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc, group=self.group)
        return [int(x) for x in rc.tolist()]

    def _exchange(self, send, send_counts, recv_counts):
        """all-to-all(v) of rows; counts are in rows."""
        out_rows = int(sum(recv_counts))
        shape = (out_rows,) + tuple(send.shape[1:])
        row_bytes = send.element_size()
        for d in send.shape[1:]:
            row_bytes *= int(d)
        me = self.rank
        self.bytes_sent += row_bytes * (int(sum(send_counts)) - int(send_counts[me]))
        self.bytes_received += row_bytes * (out_rows - int(recv_counts[me]))
        self.exchanges += 1
        if self.world == 1 and not self.force_exchange:
            # one rank: every id is this rank's own, nothing crosses a link - the "exchange"
            # is the buffer itself (a self send / receive through RCCL costs a 46 MB copy per
            # fanout step and moves nothing)
            return send.contiguous()
        if self.collective_log is not None:
            self.collective_log.append(("alltoallv", row_bytes))
        staged = self.host_staged and send.is_cuda
        src = send.contiguous().cpu() if staged else send.contiguous()
        recv = torch.empty(shape, dtype=send.dtype, device=src.device)
        dist.all_to_all_single(recv, src,
                               output_split_sizes=[int(c) for c in recv_counts],
                               input_split_sizes=[int(c) for c in send_counts],
                               group=self.group)
        return recv.to(send.device) if staged else recv

    @staticmethod
    def _pack(ids, w, t, mask, count):
        """One int32 row per root: ids (2 words per id), weights, types, mask -
        one exchange and one merge instead of four."""
        m = ids.shape[0]
        # 4 * count + 2 words: an even width keeps the rows 8-byte aligned
        buf = torch.zeros((m, 4 * count + 2), dtype=torch.int32, device=ids.device)
        if m == 0:          # a shard nobody asked anything (found by the 8-rank rehearsal: an empty
            return buf      # tensor's stride cannot be viewed as another width)
        buf[:, :2 * count] = ids.reshape(m, count).contiguous().view(torch.int32)
        buf[:, 2 * count:3 * count] = w.reshape(m, count).contiguous().view(torch.int32)
        buf[:, 3 * count:4 * count] = t.reshape(m, count)
        buf[:, 4 * count] = mask.reshape(m).to(torch.int32)
        return buf

    @staticmethod
    def _unpack(buf, count):
        if buf.shape[0] == 0:
            e = lambda dt: torch.empty((0, count), dtype=dt, device=buf.device)
            return (e(torch.int64), e(torch.float32), e(torch.int32),
                    torch.empty(0, dtype=torch.uint8, device=buf.device))
        ids = buf[:, :2 * count].contiguous().view(torch.int64)
        w = buf[:, 2 * count:3 * count].contiguous().view(torch.float32)
        t = buf[:, 3 * count:4 * count].contiguous()
        mask = buf[:, 4 * count].to(torch.uint8)
        return ids, w, t, mask

    @staticmethod
    def _run(gen):
        """Drive a *_steps generator to its result."""
        try:
            while True:
                next(gen)
        except StopIteration as stop:
            return stop.value

    def sample_neighbor(self, roots, edge_types, count, default_node=-1,
                        call_id=None, root_mask=None, root_group=1):
        """One hop for this rank's `roots` ([n] int64).  root_mask ([n /
        root_group] uint8) marks roots that stand for a missing row of the
        previous hop: they sample as node id 0 (the reference chains hops on
        its core tensors, whose empty rows hold the sentinel 0).
        Returns (ids [n,count], weights, types, row_mask [n]).

        Duplicate roots are removed BEFORE the exchange (the reference's
        ID_UNIQUE precedes ID_SPLIT, parser/compiler.cc:76-90): rows depend only
        on the node id, and on a fanout's second hop >90 % of the roots repeat,
        so the wire carries the distinct ones only."""
        return self._run(self.sample_neighbor_steps(roots, edge_types, count, default_node,
                                                    call_id, root_mask, root_group))

    def sample_neighbor_steps(self, roots, edge_types, count, default_node=-1,
                              call_id=None, root_mask=None, root_group=1):
        """The hop as a generator that yields ONCE, at the only place the host
        has to wait (the bucket sizes of the front end) - after enqueueing the
        front end when the sampler has a two-phase one (front_begin_fn /
        front_end_fn).  A caller that interleaves several minibatches advances
        the other generators there.  The yield is unconditional (also for an
        empty batch or a sampler without a two-phase front end): every rank must
        issue its collectives in the same order."""
        call_id = self._take_call_ids(1, call_id)
        roots = roots.reshape(-1).to(torch.int64)
        n = roots.numel()
        gather_idx = None
        # one listed edge type: the wire rows of the fused path carry no type column
        et_list = [int(x) for x in (edge_types if hasattr(edge_types, "__len__") else [edge_types])]
        single_type = et_list[0] if len(et_list) == 1 else None
        # (not `and n > 0`: a rank whose own batch is empty still answers its peers,
        # and must do so in the wire format they expect)
        fused = self.dedup_split_fn is not None and self.expand_fn is not None
        token = None
        if fused and self.front_begin_fn is not None:
            token = self.front_begin_fn(roots, self.partitions, self.world, root_mask,
                                        root_group)
        yield
        if fused:
            # C1 (fused): distinct ids bucketed by owner + where each position's
            # row will sit among the answers (the root mask is applied inside)
            if token is not None:
                shard_off, shard_ids, pos = self.front_end_fn(token)
            else:
                shard_off, shard_ids, pos = self.dedup_split_fn(
                    roots, self.partitions, self.world, root_mask, root_group)
            merge_idx = None
        else:
            if root_mask is not None:
                expand = root_mask.to(torch.bool).repeat_interleave(root_group)[:n]
                roots = torch.where(expand, torch.zeros_like(roots), roots)
            if self.unique_fn is not None and n > 0:
                roots, gather_idx = self.unique_fn(roots)
            # C1: bucket by owner, tell every peer how many ids it gets
            shard_off, shard_ids, merge_idx = self.split_fn(roots, self.partitions,
                                                            self.world)
        send_counts = [int(shard_off[s + 1] - shard_off[s]) for s in range(self.world)]
        recv_counts = self._exchange_counts(send_counts, roots.device)
        owned = self._exchange(shard_ids, send_counts, recv_counts)
        if fused and self.local_sample_packed is not None:
            # local sampling straight into wire rows, back along the reversed
            # split; the shards answered in the order they were asked: row pos[i]
            # of `back` is position i's row
            rows = self.local_sample_packed(owned, edge_types, count, default_node, call_id)
            return self.expand_fn(pos, self._exchange(rows, recv_counts, send_counts), count,
                                  single_type)
        # local sampling on the rows this rank owns
        ids, w, t, mask = self.local_sample(owned, edge_types, count, default_node,
                                            call_id)
        # C2: results travel back along the reversed split, one packed row each
        if fused and self.pack_fn is not None:
            back = self._exchange(self.pack_fn(ids, w, t, mask, count, single_type),
                                  recv_counts, send_counts)
            # the shards answered in the order they were asked: row pos[i] of
            # `back` is position i's row (merge + gather + unpack in one pass)
            return self.expand_fn(pos, back, count, single_type)
        back = self._exchange(self._pack(ids, w, t, mask, count), recv_counts,
                              send_counts)
        if fused:
            b_ids, b_w, b_t, b_mask = self._unpack(back, count)
            return self.expand_fn(pos, b_ids, b_w, b_t, b_mask, count)
        # IDX_MERGE / DATA_MERGE: out[merge_idx[j]] = back[j]
        rows = self.merge_fn(back, merge_idx)
        if gather_idx is not None:                 # DATA_GATHER back to positions
            rows = self.gather_fn(rows, gather_idx)
        ids, w, t, mask = self._unpack(rows, count)
        return ids, w, t, mask

    def sample_neighbor_sets(self, roots, type_sets, count, default_node=-1, call_id=None):
        """The typed draws of a heterogeneous minibatch: `roots` sampled once per edge-type set
        of `type_sets` (set s draws with call_id + s - the results of len(type_sets)
        sample_neighbor calls, bit for bit).  ONE front end, one host wait and one id
        exchange serve all the sets (the roots are the same); every set is then one owners'
        pass + one row exchange + one expansion.  Returns a list of (ids, weights, types,
        row_mask) per set."""
        return self._run(self.sample_neighbor_sets_steps(roots, type_sets, count, default_node, call_id))

    def sample_neighbor_sets_steps(self, roots, type_sets, count, default_node=-1, call_id=None):
        """sample_neighbor_sets as a generator: one yield, where the host has to wait for the
        front end's bucket sizes (as sample_neighbor_steps; run_interleaved advances the other
        minibatches there).  A sampler without the fused path yields once per set."""
        type_sets = [list(et) for et in type_sets]
        call_id = self._take_call_ids(max(len(type_sets), 1), call_id)
        fused = (self.dedup_split_fn is not None and self.expand_fn is not None and
                 self.local_sample_packed is not None)
        if not fused:
            outs = []
            for s, et in enumerate(type_sets):
                outs.append((yield from self.sample_neighbor_steps(roots, et, count, default_node, call_id + s)))
            return outs
        roots = roots.reshape(-1).to(torch.int64)
        token = None
        if self.front_begin_fn is not None:
            token = self.front_begin_fn(roots, self.partitions, self.world, None, 1)
        yield
        if token is not None:
            shard_off, shard_ids, pos = self.front_end_fn(token)
        else:
            shard_off, shard_ids, pos = self.dedup_split_fn(roots, self.partitions, self.world, None, 1)
        send_counts = [int(shard_off[s + 1] - shard_off[s]) for s in range(self.world)]
        recv_counts = self._exchange_counts(send_counts, roots.device)
        owned = self._exchange(shard_ids, send_counts, recv_counts)
        outs = []
        # (the owners' passes of all the sets as ONE launch where the shard has it)
        all_rows = None
        if self.local_sample_sets_packed is not None and len(type_sets) > 1:
            all_rows = self.local_sample_sets_packed(owned, type_sets, count, default_node, call_id)
        for s, et in enumerate(type_sets):
            single_type = et[0] if len(et) == 1 else None
            rows = all_rows[s] if all_rows is not None else \
                self.local_sample_packed(owned, et, count, default_node, call_id + s)
            outs.append(self.expand_fn(pos, self._exchange(rows, recv_counts, send_counts), count,
                                       single_type))
        return outs

    def sample_fanout(self, roots, edge_types, counts, default_node=-1, call_id=None):
        """Multi-hop fanout (tf_euler sample_fanout): returns (neighbors_list,
        weights_list, types_list) flattened like euler_ops.sample_fanout."""
        return self._run(self.sample_fanout_steps(roots, edge_types, counts, default_node,
                                                  call_id))

    def sample_fanout_steps(self, roots, edge_types, counts, default_node=-1, call_id=None):
        """sample_fanout as a generator: one yield per hop (sample_neighbor_steps)."""
        call_id = self._take_call_ids(len(counts), call_id)
        roots = roots.reshape(-1).to(torch.int64)
        neighbors, weights, types = [roots], [], []
        mask, group = None, 1
        cur = roots
        for h, count in enumerate(counts):
            ids, w, t, m = yield from self.sample_neighbor_steps(
                cur, edge_types[h], count, default_node, call_id + h, mask, group)
            neighbors.append(ids.reshape(-1))
            weights.append(w.reshape(-1))
            types.append(t.reshape(-1))
            cur, mask, group = ids.reshape(-1), m, count
        return neighbors, weights, types

    # ------------------------------------------------------------ features
    def get_dense_feature(self, nodes, feature_ids, dimensions):
        """tf_euler get_dense_feature over the sharded graph: the distinct ids
        travel to their owners (same front end as a sampling hop), each owner
        gathers its rows, the rows come back and are expanded to positions.
        Needs local_feature(owned ids, feature_ids, dimensions) -> list of
        [m, dim] float32 tensors and row_gather_fn(rows f32 [m, D], pos) -> rows."""
        nodes = nodes.reshape(-1).to(torch.int64)
        n = nodes.numel()
        dims = [int(d) for d in dimensions]
        if self.dedup_split_fn is not None:
            shard_off, shard_ids, pos = self.dedup_split_fn(nodes, self.partitions,
                                                            self.world, None, 1)
        else:
            shard_off, shard_ids, merge_idx = self.split_fn(nodes, self.partitions,
                                                            self.world)
            pos = torch.empty_like(merge_idx)
            pos[merge_idx.long()] = torch.arange(n, dtype=merge_idx.dtype,
                                                 device=merge_idx.device)
        send_counts = [int(shard_off[s + 1] - shard_off[s]) for s in range(self.world)]
        recv_counts = self._exchange_counts(send_counts, nodes.device)
        owned = self._exchange(shard_ids, send_counts, recv_counts)
        feats = self.local_feature(owned, list(feature_ids), dims)
        rows = torch.cat([f.reshape(owned.numel(), d) for f, d in zip(feats, dims)], dim=1) \
            if dims else torch.empty((owned.numel(), 0), dtype=torch.float32,
                                     device=nodes.device)
        back = self._exchange(rows.contiguous(), recv_counts, send_counts)
        out = self.row_gather_fn(back, pos)
        return list(torch.split(out, dims, dim=1)) if dims else []

    # ------------------------------------------------------- full neighbours
    def get_full_neighbor(self, nodes, edge_types, packed_rows=False):
        """`v(nodes).outV(edge_types)` over the sharded graph, GQL layout (idx
        [n,2] int32, ids int64, weights f32, types int32).  Rows have different
        lengths, so the answers travel as (row lengths) + (packed values) and
        are put back with the variable-length IDX_MERGE / DATA_MERGE
        (core/kernels/idx_merge_op.cc:32-78, data_merge_op.cc:44-110), here
        IDX_GATHER + DATA_GATHER through the position map.  Needs
        local_full_neighbor(owned, edge_types) -> (idx, ids, w, t),
        idx_gather_fn(idx, gather_idx) -> (idx_out, total) and
        data_gather_fn(data, idx, gather_idx) -> data_out.
        packed_rows: return (idx [rows, 2], ids, w, t, row [n]) - the distinct rows and
        the row of every position - instead of one copy of the row per position."""
        nodes = nodes.reshape(-1).to(torch.int64)
        dev = nodes.device
        if self.dedup_split_fn is not None:
            shard_off, shard_ids, pos = self.dedup_split_fn(nodes, self.partitions,
                                                            self.world, None, 1)
        else:
            shard_off, shard_ids, merge_idx = self.split_fn(nodes, self.partitions,
                                                            self.world)
            pos = torch.empty_like(merge_idx)
            pos[merge_idx.long()] = torch.arange(nodes.numel(), dtype=merge_idx.dtype,
                                                 device=dev)
        send_counts = [int(shard_off[s + 1] - shard_off[s]) for s in range(self.world)]
        recv_counts = self._exchange_counts(send_counts, dev)
        owned = self._exchange(shard_ids, send_counts, recv_counts)
        idx, ids, w, t = self.local_full_neighbor(owned, edge_types)
        idx = idx.reshape(-1, 2).to(torch.int64)
        lens = (idx[:, 1] - idx[:, 0]).to(torch.int32)
        # values per requester: sum of the lengths of its rows
        bounds = [0]
        for c in recv_counts:
            bounds.append(bounds[-1] + c)
        csum = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev),
                          torch.cumsum(lens.to(torch.int64), 0)])
        val_send = [int(csum[bounds[s + 1]] - csum[bounds[s]]) for s in range(self.world)]
        val_recv = self._exchange_counts(val_send, dev)
        # (8-byte rows, like every other exchange of the sampler)
        lens_back = self._exchange(lens.to(torch.int64).reshape(-1, 1), recv_counts,
                                   send_counts).reshape(-1)
        vals = torch.empty((ids.numel(), 4), dtype=torch.int32, device=dev)
        vals[:, :2] = ids.reshape(-1, 1).contiguous().view(torch.int32).reshape(-1, 2)
        vals[:, 2] = w.contiguous().view(torch.int32)
        vals[:, 3] = t
        vals_back = self._exchange(vals, val_send, val_recv)
        end = torch.cumsum(lens_back.to(torch.int64), 0)
        idx_cat = torch.stack([end - lens_back, end], dim=1).to(torch.int32)
        if packed_rows:
            # the rows as they came back (one per distinct id asked, bucketed by owner) and
            # the row of every position: callers that can index rows do not need a copy of
            # a hub's list per position that named it
            return (idx_cat, vals_back[:, :2].contiguous().view(torch.int64).reshape(-1),
                    vals_back[:, 2].contiguous().view(torch.float32),
                    vals_back[:, 3].contiguous(), pos.to(torch.int32))
        out_idx, _total = self.idx_gather_fn(idx_cat, pos)
        out_ids = self.data_gather_fn(vals_back[:, :2].contiguous().view(torch.int64)
                                      .reshape(-1), idx_cat, pos)
        out_w = self.data_gather_fn(vals_back[:, 2].contiguous().view(torch.float32),
                                    idx_cat, pos)
        out_t = self.data_gather_fn(vals_back[:, 3].contiguous(), idx_cat, pos)
        return out_idx, out_ids, out_w, out_t

    # ------------------------------------------------------ sparse features
    def get_sparse_feature(self, nodes, feature_ids, default_values=None):
        """tf_euler get_sparse_feature over the sharded graph: per uint64 feature
        id the SparseTensor triple (indices [nnz, 2] int64, values [nnz] int64,
        dense_shape [n, max_len]).  The distinct ids travel to their owners, each
        owner answers in the GQL `values()` layout (row lengths + packed values,
        no default entries), the answers come back like the rows of
        get_full_neighbor (IDX_GATHER / DATA_GATHER through the position map) and
        the requester inserts the default entry of nodes without values, as the
        TF kernel does (tf_euler/kernels/get_sparse_feature_op.cc:96-107).  Needs
        local_sparse_feature(owned ids, fid) -> (idx [m, 2] int32, values int64)
        and the idx_gather_fn / data_gather_fn of get_full_neighbor."""
        nodes = nodes.reshape(-1).to(torch.int64)
        dev = nodes.device
        n = nodes.numel()
        if default_values is None:
            default_values = [0] * len(feature_ids)
        if self.dedup_split_fn is not None:
            shard_off, shard_ids, pos = self.dedup_split_fn(nodes, self.partitions,
                                                            self.world, None, 1)
        else:
            shard_off, shard_ids, merge_idx = self.split_fn(nodes, self.partitions,
                                                            self.world)
            pos = torch.empty_like(merge_idx)
            pos[merge_idx.long()] = torch.arange(n, dtype=merge_idx.dtype, device=dev)
        send_counts = [int(shard_off[s + 1] - shard_off[s]) for s in range(self.world)]
        recv_counts = self._exchange_counts(send_counts, dev)
        owned = self._exchange(shard_ids, send_counts, recv_counts)
        bounds = [0]
        for c in recv_counts:
            bounds.append(bounds[-1] + c)
        outs = []
        for fid, dv in zip(feature_ids, default_values):
            idx, vals = self.local_sparse_feature(owned, int(fid))
            idx = idx.reshape(-1, 2).to(torch.int64)
            lens = idx[:, 1] - idx[:, 0]
            csum = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev),
                              torch.cumsum(lens, 0)])
            val_send = [int(csum[bounds[s + 1]] - csum[bounds[s]]) for s in range(self.world)]
            val_recv = self._exchange_counts(val_send, dev)
            lens_back = self._exchange(lens.reshape(-1, 1).contiguous(), recv_counts,
                                       send_counts).reshape(-1)
            vals_back = self._exchange(vals.to(torch.int64).reshape(-1, 1).contiguous(),
                                       val_send, val_recv).reshape(-1)
            end = torch.cumsum(lens_back, 0)
            idx_cat = torch.stack([end - lens_back, end], dim=1).to(torch.int32)
            out_idx, _total = self.idx_gather_fn(idx_cat, pos)
            out_vals = self.data_gather_fn(vals_back.contiguous(), idx_cat, pos)
            outs.append(sparse_from_core(out_idx, out_vals, int(dv)))
        return outs

    # ---------------------------------------------------- layerwise sampling
    def get_edge_sum_weight(self, nodes, edge_types):
        """API_GET_EDGE_SUM_WEIGHT over the sharded graph: one f32 per node, by
        the id exchange of get_dense_feature.  Needs
        local_edge_sum_weight(owned ids, edge_types) -> f32 [m] and row_gather_fn."""
        nodes = nodes.reshape(-1).to(torch.int64)
        n = nodes.numel()
        if self.dedup_split_fn is not None:
            shard_off, shard_ids, pos = self.dedup_split_fn(nodes, self.partitions,
                                                            self.world, None, 1)
        else:
            shard_off, shard_ids, merge_idx = self.split_fn(nodes, self.partitions,
                                                            self.world)
            pos = torch.empty_like(merge_idx)
            pos[merge_idx.long()] = torch.arange(n, dtype=merge_idx.dtype,
                                                 device=merge_idx.device)
        send_counts = [int(shard_off[s + 1] - shard_off[s]) for s in range(self.world)]
        recv_counts = self._exchange_counts(send_counts, nodes.device)
        owned = self._exchange(shard_ids, send_counts, recv_counts)
        sums = self.local_edge_sum_weight(owned, edge_types).reshape(-1, 1)
        # (8-byte rows on the wire, like every other exchange of the sampler)
        rows = torch.zeros((owned.numel(), 2), dtype=torch.float32, device=nodes.device)
        rows[:, :1] = sums
        back = self._exchange(rows.contiguous(), recv_counts, send_counts)
        return self.row_gather_fn(back, pos)[:, 0].contiguous()

    def sample_layer(self, roots, edge_types, default_node=-1, call_id=None):
        """API_SAMPLE_L over the sharded graph.  The draw of a root depends on its
        POSITION in the list (a root listed twice is sampled twice), so nothing is
        deduplicated: ID_SPLIT buckets (id, position) by owner, the owner draws
        with the requester's positions as RNG streams
        (euler_gpu_sample_layer_at), IDX_MERGE / DATA_MERGE put the rows back.
        Needs local_sample_layer(ids, positions, edge_types, default_node,
        call_id) -> (ids, w, t)."""
        call_id = self._take_call_ids(1, call_id)
        roots = roots.reshape(-1).to(torch.int64)
        dev = roots.device
        shard_off, shard_ids, merge_idx = self.split_fn(roots, self.partitions, self.world)
        send_counts = [int(shard_off[s + 1] - shard_off[s]) for s in range(self.world)]
        recv_counts = self._exchange_counts(send_counts, dev)
        ask = torch.stack([shard_ids.to(torch.int64), merge_idx.to(torch.int64)], dim=1)
        got = self._exchange(ask.contiguous(), send_counts, recv_counts)
        ids, w, t = self.local_sample_layer(got[:, 0].contiguous(), got[:, 1].contiguous(),
                                            edge_types, default_node, call_id)
        rows = torch.empty((ids.numel(), 4), dtype=torch.int32, device=dev)
        rows[:, :2] = ids.to(torch.int64).reshape(-1, 1).contiguous().view(torch.int32).reshape(-1, 2)
        rows[:, 2] = w.to(torch.float32).contiguous().view(torch.int32)
        rows[:, 3] = t.to(torch.int32)
        back = self._exchange(rows, recv_counts, send_counts)
        out = self.merge_fn(back, merge_idx)
        return (out[:, :2].contiguous().view(torch.int64).reshape(-1),
                out[:, 2].contiguous().view(torch.float32), out[:, 3].contiguous())

    def sparse_get_adj(self, nodes, nb_nodes, edge_types, n=-1, m=-1):
        """tf_euler sparse_get_adj over the sharded graph: the SparseTensor
        triple of the [batch, n, m] adjacency.  The rows of `nodes` come to the
        requester through get_full_neighbor (EdgeExist is membership in the row,
        DESIGN.md), the membership test is a join on (source, compact id) keys
        and the TF kernel's explicit zero at (b, n-1, m-1) is added here."""
        nodes = nodes.reshape(-1).to(torch.int64)
        nb_nodes = nb_nodes.reshape(-1).to(torch.int64)
        dev = nodes.device
        n = nodes.numel() if n == -1 else int(n)
        m = nb_nodes.numel() if m == -1 else int(m)
        batch = nodes.numel() // n if n else 0
        if getattr(self, "local_adj_mask", None) is not None:
            return self._sparse_get_adj_owner_side(nodes, nb_nodes, edge_types, batch, n, m)
        idx, ids, _w, _t = self.get_full_neighbor(nodes, edge_types)
        return adj_from_rows(idx, ids, nb_nodes, batch, n, m)

    def _sparse_get_adj_owner_side(self, nodes, nb_nodes, edge_types, batch, n, m):
        """The adjacency answered on the OWNERS (core/kernels/sparse_get_adj_op.cc:35-92 is
        the per-shard op the reference runs remotely): every rank learns every rank's
        (sources, candidates) - a few KB -, computes the hit mask of the sources it owns
        (local_adj_mask(nodes, nb_nodes, batch, n, m, edge_types) -> int64 [batch * n,
        words]; a source it does not own gives zeros) and sends each requester its mask;
        the requester ORs the masks - an id has one owner - and builds the TF triple
        (adj_from_mask_fn(mask, batch, n, m)).  8 bytes per 64 candidates and source on the
        wire instead of the sources' rows."""
        dev = nodes.device
        wire = torch.device("cpu") if self.host_staged else dev
        W = self.world
        shp = torch.tensor([batch, n, m], dtype=torch.int64, device=wire)
        all_shp = [torch.empty_like(shp) for _ in range(W)]
        dist.all_gather(all_shp, shp, group=self.group)
        shapes = [[int(x) for x in t.tolist()] for t in all_shp]
        max_nodes = max(b * n_ for b, n_, _ in shapes)
        max_nb = max(b * m_ for b, _, m_ in shapes)
        pad = torch.zeros(max_nodes + max_nb, dtype=torch.int64, device=wire)
        pad[:batch * n] = nodes[:batch * n].to(wire)
        pad[max_nodes:max_nodes + batch * m] = nb_nodes[:batch * m].to(wire)
        allq = [torch.empty_like(pad) for _ in range(W)]
        dist.all_gather(allq, pad, group=self.group)
        parts, send_counts = [], []
        for r in range(W):
            b_, n_, m_ = shapes[r]
            q_nodes = allq[r][:b_ * n_].to(dev)
            q_nb = allq[r][max_nodes:max_nodes + b_ * m_].to(dev)
            mk = self.local_adj_mask(q_nodes, q_nb, b_, n_, m_, edge_types).reshape(-1)
            parts.append(mk)
            send_counts.append(int(mk.numel()))
        words = (m + 63) // 64
        mine = batch * n * words
        send = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64, device=dev)
        got = self._exchange(send.reshape(-1, 1), send_counts, [mine] * W).reshape(W, -1)
        mask = got[0].clone()
        for r in range(1, W):
            mask |= got[r]
        return self.adj_from_mask_fn(mask.reshape(batch * n, words), batch, n, m)

    def sample_neighbor_layerwise(self, nodes, edge_types, count, default_node=-1,
                                  call_id=None, weight_func=''):
        """With a weight function: the rows of `nodes` are fetched
        (get_full_neighbor) and API_LOCAL_SAMPLE_L - which needs no graph, only
        those lists - runs on the requester (local_layer_fn(idx, ids, w, t, batch,
        n, m, weight_func, default_node, call_id) -> (ids, w, t)).  Otherwise:
        tf_euler sample_neighbor_layerwise (weight_func == '') over the sharded
        graph: nodes [batch, n] -> (neighbors [batch, count], adjacency triple),
        equal to the single-GPU result: edge weight sums by id exchange, the root
        draw locally (it needs no graph; sample_root_fn(roots, weights, m,
        default_node, call_id)), the layer draw on the owners with the requester's
        positions, the adjacency from the rows fetched by get_full_neighbor."""
        call_id = self._take_call_ids(1, call_id)
        nodes = nodes.to(torch.int64)
        batch, n = nodes.shape
        if weight_func:
            idx, ids, w_, t_ = self.get_full_neighbor(nodes.reshape(-1), edge_types)
            l_nb, _lw, _lt = self.local_layer_fn(idx, ids, w_, t_, batch, n, int(count),
                                                 weight_func, default_node, call_id)
            l_nb = l_nb.reshape(batch, int(count))
            return l_nb, self.sparse_get_adj(nodes, l_nb, edge_types, n, int(count))
        w = self.get_edge_sum_weight(nodes.reshape(-1), edge_types).reshape(batch, n)
        l_root = self.sample_root_fn(nodes, w, int(count), default_node, call_id)
        l_nb, _lw, _lt = self.sample_layer(l_root.reshape(-1), edge_types, default_node, call_id)
        l_nb = l_nb.reshape(batch, int(count))
        return l_nb, self.sparse_get_adj(nodes, l_nb, edge_types, n, int(count))

    # ---------------------------------------------------------- sample_node
    def sample_node(self, count, node_type=-1, call_id=None):
        """SampleNode over the shards: SAMPLE_NODE_SPLIT divides
        `count` in proportion to the shards' weight sums of the type (remainder
        by RNG domain SPLIT, core/kernels/sample_node_split_op.cc:57-85), every
        shard draws its share from its own alias tables, APPEND_MERGE
        concatenates in shard order (append_merge_op.cc:67-93).  Every rank
        gets the same [count] tensor.  Needs local_sample_node(count, node_type,
        call_id), node_weight_sum(node_type) and node_split_fn(call_id, count,
        weights[world + 1]) -> counts[world]."""
        call_id = self._take_call_ids(1, call_id)
        dev = self.device
        wire = torch.device("cpu") if self.host_staged else dev
        mine = torch.tensor([float(self.node_weight_sum(node_type))], dtype=torch.float32,
                            device=wire)
        allw = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(allw, mine, group=self.group)
        w = torch.cat(allw).cpu().numpy().astype("float32")
        total = w[0].copy()
        for x in w[1:]:
            total = (total + x).astype("float32")          # f32 adds in shard order
        split = [int(c) for c in self.node_split_fn(call_id, int(count),
                                                   list(w) + [float(total)])]
        own = self.local_sample_node(split[self.rank], node_type, call_id) \
            if split[self.rank] > 0 else torch.empty(0, dtype=torch.int64, device=dev)
        width = max(max(split), 1)
        pad = torch.zeros(width, dtype=torch.int64, device=wire)
        pad[:own.numel()] = own.to(wire)
        parts = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(parts, pad, group=self.group)
        return torch.cat([parts[s][:split[s]] for s in range(self.world)]).to(dev)

    def random_walk(self, nodes, edge_types, p=1.0, q=1.0, default_node=-1, call_id=None):
        """tf_euler random_walk over the sharded graph; edge_types is a list (walk_len)
        of per-step edge type lists.  Returns [n, walk_len + 1] int64, identical to the
        single-GPU kernel (step s uses call_id + s).
        p = q = 1 (TraditionalRandomWalk, tf_euler/kernels/random_walk_op.cc:207-247):
        one id / result exchange per step, sample_neighbor with count 1.
        Otherwise node2vec as the reference's client runs it (random_walk_op.cc:83-168):
        every step fetches the neighbour lists of the walkers' current nodes
        (get_full_neighbor: `v(nodes).outV(edge_types)`, one row per DISTINCT node), keeps
        the previous step's lists as the parents' and draws on the requester -
        n2v_step_fn(call_id, c_row, c_idx, c_ids, c_w, p_row, p_idx, p_ids, parent_ids, p, q,
        default_node) -> next nodes (euler_gpu_node2vec_step).  The draw is keyed by the
        walker's index in `nodes`."""
        call_id = self._take_call_ids(max(len(edge_types), 1), call_id)
        nodes = nodes.reshape(-1).to(torch.int64)
        cols = [nodes]
        # random_walk_op.cc:281: fabs(p_ - 1.0) <= kEps && fabs(q_ - 1.0) <= kEps, in f32
        import numpy as _np
        k_eps = 1.0e-6
        if abs(float(_np.float32(p)) - 1.0) <= k_eps and abs(float(_np.float32(q)) - 1.0) <= k_eps:
            if getattr(self, "c_walk_fn", None) is not None:
                # the walk over levels of merged walkers, orchestrated inside libeuler_gpu.so
                # (euler_gpu_sharded_random_walk): one host wait per step, every step as large as
                # the distinct nodes its walkers stand on
                return self.c_walk_fn(nodes, edge_types, default_node, call_id)
            cur, mask = nodes, None
            for s, et in enumerate(edge_types):
                ids, _, _, mask = self.sample_neighbor(cur, et, 1, default_node,
                                                       call_id + s, mask, 1)
                cur = ids.reshape(-1)
                cols.append(cur)
            return torch.stack(cols, dim=1)
        if getattr(self, "c_n2v_fn", None) is not None:
            # the same loop inside libeuler_gpu.so (euler_gpu_sharded_node2vec_walk)
            return self.c_n2v_fn(nodes, edge_types, p, q, default_node, call_id)
        cur, parent = nodes, nodes            # parent_ids_ starts as the start nodes
        p_row = p_idx = p_ids = None          # parent_neighbors_ starts empty
        for s, et in enumerate(edge_types):
            c_idx, c_ids, c_w, _t, c_row = self.get_full_neighbor(cur, et, packed_rows=True)
            nxt = self.n2v_step_fn(call_id + s, c_row, c_idx, c_ids, c_w, p_row, p_idx, p_ids,
                                   parent, p, q, default_node)
            parent, cur = cur, nxt
            p_row, p_idx, p_ids = c_row, c_idx, c_ids
            cols.append(cur)
        return torch.stack(cols, dim=1)


def run_interleaved(make_steps, n_jobs, in_flight=2, enter=None, on_result=None):
    """Drive n_jobs *_steps generators from one host thread with `in_flight` of
    them active: job j runs in slot j % in_flight and is advanced one step at a
    time, slot by slot, so that while the host waits for one minibatch's bucket
    sizes the GPU has the other minibatches' kernels and exchanges queued.  The
    schedule depends on nothing but (n_jobs, in_flight) and the number of yields
    per job, which the *_steps generators keep independent of the data - every
    rank therefore issues its collectives in the same order.
      make_steps(j) -> generator for job j (called when its slot becomes free);
      enter(slot)   -> optional context manager entered around every advance of
                       that slot (e.g. its HIP stream).
      on_result(job, value) -> optional consumer; the results are then NOT kept
                       (a fanout's outputs are ~0.6 GB per minibatch of the metric:
                       holding every step's tensors makes the caching allocator go
                       back to the driver for each new step).
    Returns the results in job order (None entries when on_result consumes them)."""
    import contextlib
    results = [None] * n_jobs
    slots = [None] * in_flight           # (job, generator)
    next_job = 0
    live = 0
    while next_job < n_jobs or live:
        for k in range(in_flight):
            ctx = enter(k) if enter is not None else contextlib.nullcontext()
            with ctx:
                if slots[k] is None and next_job < n_jobs and next_job % in_flight == k:
                    slots[k] = (next_job, make_steps(next_job))
                    next_job += 1
                    live += 1
                if slots[k] is None:
                    continue
                job, gen = slots[k]
                try:
                    next(gen)
                except StopIteration as stop:
                    if on_result is not None:
                        on_result(job, stop.value)
                    else:
                        results[job] = stop.value
                    slots[k] = None
                    live -= 1
    return results


class ShmCounts:
    """All-to-all of the per-peer row counts of a hop through a shared-memory
    mailbox (euler_shm_* in include/euler_gpu.h) - the ranks are processes of one
    node, so the counts need neither a GPU collective nor a device sync.
    Construction is collective over `group`; `ok` is the same on every rank (any
    rank that cannot attach or fails the self-test makes all ranks fall back to
    the GPU exchange)."""

    def __init__(self, group=None, device=None):
        from . import _lib
        import ctypes as C
        import uuid
        self._lib, self._C = _lib, C
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._h = None
        on_gpu = dist.get_backend(group) == "nccl"
        dev = device if on_gpu else "cpu"
        name = ["/euler_amd_%d_%s" % (os.getpid(), uuid.uuid4().hex[:12])]
        dist.broadcast_object_list(name, src=dist.get_global_rank(group, 0) if group is not None else 0,
                                   group=group, **({"device": dev} if on_gpu else {}))
        self.name = name[0]
        ok = 1

        def agree(flag):
            v = torch.tensor([flag], dtype=torch.int32, device=dev)
            dist.all_reduce(v, op=dist.ReduceOp.MIN, group=group)
            return int(v.item())

        L = _lib.lib()
        h = C.c_void_p()
        if self.rank == 0:
            ok = 1 if L.euler_shm_open(self.name.encode(), 0, self.world, 1, C.byref(h)) == 0 else 0
        ok = agree(ok)                       # the region exists (or nobody goes on)
        if ok and self.rank != 0:
            ok = 1 if L.euler_shm_open(self.name.encode(), self.rank, self.world, 0, C.byref(h)) == 0 else 0
        if h.value:
            self._h = h
        ok = agree(ok)                       # everybody is attached
        if self.rank == 0 and self._h is not None:
            L.euler_shm_unlink(self._h)       # the mappings stay; the name cannot leak
        if ok:
            try:
                got = self(list(range(self.rank * 1000, self.rank * 1000 + self.world)))
                ok = 1 if got == [p * 1000 + self.rank for p in range(self.world)] else 0
            except Exception:
                ok = 0
        self.ok = bool(agree(ok))
        if not self.ok:
            self.close()

    def __call__(self, send_counts):
        C = self._C
        n = self.world
        send = (C.c_int64 * n)(*[int(c) for c in send_counts])
        recv = (C.c_int64 * n)()
        self._lib.check(self._lib.lib().euler_shm_alltoall_i64(self._h, send, recv, 1, 60000))
        return list(recv)

    def close(self):
        if self._h is not None:
            self._lib.lib().euler_shm_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


DENSE_ID_LIMIT = 1 << 30      # ids: a 4 GB table at most


def gpu_sharded_sampler(graph, partitions=None, group=None, dedup=True, dense_ids=None,
                        packed=True):
    """ShardedSampler over an euler_amd.Graph shard living on this rank's GPU.
    dedup: True / "fused" = one-call front end (euler_gpu_dedup_split) and back end
    (euler_gpu_expand_rows); "ops" = ID_UNIQUE / ID_SPLIT / merge / gather as separate
    kernels; False = no duplicate removal.  dense_ids: False = always find
    duplicates by hashing (None = by id-indexed table when the ids allow it).
    packed: the shard samples straight into wire rows (False = sample, then pack)."""
    from . import ops

    # a dataset's files are kept by file_idx % shards while ids are routed by
    # (id % partitions) % shards: both must use euler.meta's partitions_num
    meta_parts = getattr(graph, "partitions", 0)
    if meta_parts:
        if partitions is None:
            partitions = meta_parts
        elif int(partitions) != int(meta_parts):
            raise ValueError("gpu_sharded_sampler: partitions = %d but the dataset was written with "
                             "partitions_num = %d (ids would be routed to ranks that do not hold "
                             "their rows)" % (int(partitions), meta_parts))

    def local_sample(owned, edge_types, count, default_node, call_id):
        ids, w, t, mask = graph.sample_neighbor(owned, edge_types, count,
                                                default_node, layout="tf",
                                                call_id=call_id, return_mask=True,
                                                dedup=not bool(dedup))
        return ids, w, t, mask

    def gather_rows(rows, gather_idx):
        # MPGather kernel on the int32 rows viewed as f32 words (a bit copy)
        return ops.gather(rows.view(torch.float32), gather_idx).view(torch.int32)

    fused = dedup == "fused" or dedup is True
    # duplicate detection of the front end: when every shard's ids are base +
    # stride * row and the largest id of the whole graph is small enough for a
    # table indexed by the id itself (4 bytes per id, kept by this sampler),
    # one store + one load per id replaces two rounds of hashing.  All ranks
    # agree on the choice (two small all-reduces at construction).
    dense_table = None
    if fused and dense_ids is not False:
        max_id, identity = graph.id_range()
        if dist.is_available() and dist.is_initialized():
            on_gpu = dist.get_backend(group) == "nccl"
            v = torch.tensor([max_id, 0 if identity else 1], dtype=torch.int64,
                             device=graph.device if on_gpu else "cpu")
            dist.all_reduce(v, op=dist.ReduceOp.MAX, group=group)
            max_id, identity = int(v[0]), int(v[1]) == 0
        if identity and 0 < max_id + 2 <= DENSE_ID_LIMIT:
            dense_table = torch.empty(max_id + 2, dtype=torch.int32, device=graph.device)

    def front(ids, parts, shards, root_mask, root_group):
        return ops.dedup_split(ids, parts, shards, root_mask, root_group,
                               dense_table=dense_table)

    S = ShardedSampler(local_sample, ops.id_split, ops.merge_rows, partitions,
                       group, ops.id_unique if dedup else None, gather_rows,
                       front if fused else None,
                       ops.expand_packed if fused else None)
    S.dense_table = dense_table
    if fused:
        handle = ops.FrontHandle()          # one front end in flight per sampler
        S.front_begin_fn = lambda ids, parts, shards, root_mask, root_group: handle.begin(
            ids, parts, shards, root_mask, root_group, dense_table=dense_table)
        S.front_end_fn = lambda token: token.end()
    # peer counts through a shared-memory mailbox instead of a GPU collective
    # plus device sync per hop (all ranks agree on whether it came up)
    if S.world > 1 and os.environ.get("EULER_AMD_SHM_COUNTS", "1") != "0":
        shm = ShmCounts(group, graph.device)
        if shm.ok:
            S.counts_fn = shm
    if fused:
        S.pack_fn = ops.pack_rows
        if packed:
            S.local_sample_packed = graph.sample_neighbor_packed
            S.local_sample_sets_packed = graph.sample_neighbor_sets_packed
    S.device = graph.device
    S.local_feature = graph.get_dense_feature
    S.row_gather_fn = lambda rows, pos: ops.gather(rows, pos.to(torch.int32))
    S.local_sample_node = lambda count, node_type, call_id: graph.sample_node(
        count, node_type, call_id=call_id)

    def weight_sum(node_type):
        sums = graph.node_weight_sums()
        return float(sums.sum(dtype="float32")) if node_type == -1 else float(sums[node_type])

    S.local_full_neighbor = graph.get_full_neighbor
    S.local_sparse_feature = graph.get_sparse_feature_core
    S.local_edge_sum_weight = graph.get_edge_sum_weight
    S.sample_root_fn = lambda roots, w, m, default_node, call_id: graph.sample_root(
        roots, w, m, default_node, call_id=call_id)
    S.local_sample_layer = lambda ids, positions, et, default_node, call_id: graph.sample_layer(
        ids, et, default_node, call_id=call_id, positions=positions)
    S.local_layer_fn = lambda idx, ids, w, t, batch, n, m, wf, default_node, call_id: \
        graph.local_sample_layer(idx, ids, w, t, batch, n, m, wf, default_node, call_id=call_id)
    S.idx_gather_fn = ops.idx_gather
    S.data_gather_fn = ops.data_gather
    S.node_weight_sum = weight_sum
    S.node_split_fn = lambda call_id, count, weights: ops.sample_node_split(
        graph.seed, call_id, count, weights)
    S.n2v_step_fn = lambda call_id, *lists: ops.node2vec_step(graph.seed, call_id, *lists)
    if fused and graph.device.type == "cuda" and os.environ.get("EULER_AMD_C_WALK", "1") != "0":
        tr_c = CTransport(group, counts_fn=S.counts_fn, device=graph.device)
        S.c_transport = tr_c
        # (measured on one rank, 1M walkers x 40 steps: 1 cohort 2.55 ms, 2 cohorts 4.18, 4 cohorts
        # 6.80 - the cohorts' steps share one stream and each cohort merges only its own walkers;
        # more than one can only pay where an exchange's latency is there to hide)
        S.walk_cohorts = int(os.environ.get("EULER_AMD_WALK_COHORTS", "1"))
        S.c_walk_fn = lambda nodes, edge_types, default_node, call_id: c_sharded_random_walk(
            graph, tr_c, nodes, edge_types, default_node, call_id, S.partitions, S.walk_cohorts,
            dense_table)
        if os.environ.get("EULER_AMD_C_N2V", "1") != "0":
            S.c_n2v_fn = lambda nodes, edge_types, p, q, default_node, call_id: c_sharded_node2vec_walk(
                graph, tr_c, nodes, edge_types, p, q, default_node, call_id, S.partitions, dense_table)
    S.local_adj_mask = graph.sparse_adj_mask
    S.adj_from_mask_fn = type(graph).adj_from_mask
    return S


# --------------------------------------------------------------------------
# The C-level multi-GPU path (include/euler_gpu.h: euler_gpu_sharded_sample_fanout)
# driven from Python: the hop's orchestration runs inside libeuler_gpu.so, the
# exchange goes through an euler_gpu_transport.
# --------------------------------------------------------------------------
class _DevBytes(object):
    """A raw device pointer as a uint8 torch tensor (no copy): __cuda_array_interface__."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1",
                                         "data": (int(ptr), False), "version": 2}


class CTransport:
    """euler_gpu_transport over a torch.distributed group, for driving the C entry points
    (euler_gpu_sharded_sample_fanout / _random_walk) from Python.
    nccl backend: the callbacks hand the device buffers to all_to_all_single as they are
    (RCCL over xGMI; the C call must be made with `stream` = torch's current stream, which
    c_sharded_* do).  gloo backend - ranks that share a GPU in the one-GPU tests, where RCCL
    refuses two ranks per device -: the rows are staged through the host (hipMemcpy of
    libamdhip64).  A production C++ host uses euler_gpu_transport_rccl with its ncclComm_t
    instead.  counts_fn: an all-to-all of the per-peer counts on the host (ShmCounts), else
    the counts go through the group."""

    def __init__(self, group=None, counts_fn=None, device=None):
        import ctypes as C
        from . import _lib
        self._C = C
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.bytes_sent = 0
        self.log = None        # not None: every callback is appended as (kind, peers | row bytes) - tests
        self.counts_fn = counts_fn
        self.on_gpu = dist.get_backend(group) == "nccl"
        self.device = device
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self._hip = hip

        COUNTS = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64))
        A2AV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p,
                           C.POINTER(C.c_int64), C.c_int64, C.c_void_p)

        class Transport(C.Structure):
            _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("user", C.c_void_p),
                        ("alltoall_counts", COUNTS), ("alltoallv", A2AV)]

        def counts(_user, send, recv):
            try:
                if self.log is not None:
                    self.log.append(("counts", self.world))
                if self.counts_fn is not None:
                    got = self.counts_fn([send[p] for p in range(self.world)])
                    for p in range(self.world):
                        recv[p] = int(got[p])
                    return 0
                sc = torch.tensor([send[p] for p in range(self.world)], dtype=torch.int64,
                                  device=self.device if self.on_gpu else "cpu")
                rc = torch.empty_like(sc)
                dist.all_to_all_single(rc, sc, group=self.group)
                rc = rc.tolist()
                for p in range(self.world):
                    recv[p] = int(rc[p])
                return 0
            except Exception:                     # a callback must not raise into C
                return -3

        def a2av(_user, send_dev, send_rows, recv_dev, recv_rows, row_bytes, stream):
            try:
                if self.log is not None:
                    self.log.append(("alltoallv", int(row_bytes)))
                s_rows = [int(send_rows[p]) for p in range(self.world)]
                r_rows = [int(recv_rows[p]) for p in range(self.world)]
                if self.on_gpu:
                    # device buffers as they are; torch's RCCL work is ordered after the current
                    # stream, on which the C call enqueued the kernels that filled `send`
                    dev_ = self.device
                    sb = torch.as_tensor(_DevBytes(send_dev, max(sum(s_rows) * row_bytes, 1)), device=dev_)
                    rb = torch.as_tensor(_DevBytes(recv_dev, max(sum(r_rows) * row_bytes, 1)), device=dev_)
                    dist.all_to_all_single(rb[:sum(r_rows) * row_bytes], sb[:sum(s_rows) * row_bytes],
                                           output_split_sizes=[r * row_bytes for r in r_rows],
                                           input_split_sizes=[s_ * row_bytes for s_ in s_rows],
                                           group=self.group)
                    self.bytes_sent += (sum(s_rows) - s_rows[self.rank]) * row_bytes
                    return 0
                hip.hipStreamSynchronize(stream)          # the rows to send are ready
                sb = torch.empty(sum(s_rows) * row_bytes, dtype=torch.uint8)
                rb = torch.empty(sum(r_rows) * row_bytes, dtype=torch.uint8)
                if sb.numel():
                    if hip.hipMemcpy(sb.data_ptr(), send_dev, sb.numel(), 2) != 0:
                        return -3
                dist.all_to_all_single(rb, sb, output_split_sizes=[r * row_bytes for r in r_rows],
                                       input_split_sizes=[s * row_bytes for s in s_rows],
                                       group=self.group)
                if rb.numel():
                    if hip.hipMemcpy(recv_dev, rb.data_ptr(), rb.numel(), 1) != 0:
                        return -3
                self.bytes_sent += (sum(s_rows) - s_rows[self.rank]) * row_bytes
                return 0
            except Exception:
                return -3

        self._keep = (COUNTS(counts), A2AV(a2av))
        self.struct = Transport(self.rank, self.world, None, self._keep[0], self._keep[1])

    def ptr(self):
        return self._C.byref(self.struct)


def c_sharded_sample_fanout(graph, transport, roots, edge_types, counts, default_node=-1,
                            call_id=0, partitions=None):
    """tf_euler sample_fanout through euler_gpu_sharded_sample_fanout (the C entry a
    C++ host calls): same result as Graph.sample_fanout on the unsharded graph."""
    import ctypes as C
    import numpy as np
    from . import _lib
    L = _lib.lib()
    dev = graph.device
    roots = roots.reshape(-1).to(torch.int64).to(dev).contiguous()
    layers = len(counts)
    et = np.ascontiguousarray(np.asarray(edge_types, dtype=np.int32).reshape(layers, -1))
    k = et.shape[1] if layers else 0
    cnt = np.ascontiguousarray(np.asarray(counts, dtype=np.int32))
    n = roots.numel()
    outs_n, outs_w, outs_t, m = [], [], [], n
    for c in counts:
        m *= int(c)
        outs_n.append(torch.empty(m, dtype=torch.int64, device=dev))
        outs_w.append(torch.empty(m, dtype=torch.float32, device=dev))
        outs_t.append(torch.empty(m, dtype=torch.int32, device=dev))
    ws = torch.empty(max(int(L.euler_gpu_sample_fanout_workspace(
        n, cnt.ctypes.data_as(_lib.i32p), layers)), 16), dtype=torch.uint8, device=dev)
    pn = (C.c_void_p * layers)(*[t.data_ptr() for t in outs_n])
    pw = (C.c_void_p * layers)(*[t.data_ptr() for t in outs_w])
    pt = (C.c_void_p * layers)(*[t.data_ptr() for t in outs_t])
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(L.euler_gpu_sharded_sample_fanout(
            graph._h, transport.ptr() if hasattr(transport, "ptr") else transport, st, graph.seed,
            int(call_id) & 0xFFFFFFFF, C.c_void_p(roots.data_ptr()), n,
            et.ctypes.data_as(_lib.i32p), k, cnt.ctypes.data_as(_lib.i32p), layers,
            int(default_node), int(partitions or transport.world), pn, pw, pt,
            C.c_void_p(ws.data_ptr())))
    return [roots] + outs_n, outs_w, outs_t


def c_sharded_random_walk(graph, transport, starts, edge_types, default_node=-1, call_id=0,
                          partitions=None, cohorts=1, dense_table=None, return_stats=False):
    """tf_euler random_walk with p = q = 1 through euler_gpu_sharded_random_walk (the C entry a
    C++ host calls): [n, len(edge_types) + 1] int64, the same result as Graph.random_walk on
    the unsharded graph.  edge_types: a list (walk_len) of per-step edge type lists."""
    import ctypes as C
    import numpy as np
    from . import _lib
    L = _lib.lib()
    dev = graph.device
    starts = starts.reshape(-1).to(torch.int64).to(dev).contiguous()
    n = starts.numel()
    walk_len = len(edge_types)
    et = np.ascontiguousarray(np.asarray(edge_types, dtype=np.int32).reshape(walk_len, -1)) \
        if walk_len else np.zeros((0, 1), np.int32)
    k = et.shape[1] if walk_len else 1
    out = torch.empty((n, walk_len + 1), dtype=torch.int64, device=dev)
    stats = (C.c_int64 * 4)()
    limit = dense_table.numel() - 1 if dense_table is not None else 0
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(L.euler_gpu_sharded_random_walk(
            graph._h, transport.ptr() if hasattr(transport, "ptr") else transport, st, graph.seed,
            int(call_id) & 0xFFFFFFFF, C.c_void_p(starts.data_ptr()), n, et.ctypes.data_as(_lib.i32p), k,
            walk_len, int(default_node), int(partitions or transport.world), int(cohorts),
            C.c_void_p(dense_table.data_ptr()) if dense_table is not None else None, limit,
            C.c_void_p(out.data_ptr()), stats if return_stats else None))
    if return_stats:
        return out, {"host_waits": int(stats[0]), "level_entries": int(stats[1]),
                     "ids_sent": int(stats[2]), "cohorts": int(stats[3])}
    return out


def c_sharded_node2vec_walk(graph, transport, starts, edge_types, p, q, default_node=-1, call_id=0,
                            partitions=None, dense_table=None, return_stats=False):
    """tf_euler random_walk with p or q != 1 (node2vec) through euler_gpu_sharded_node2vec_walk
    (the C entry a C++ host calls): [n, len(edge_types) + 1] int64, the same result as
    Graph.random_walk(p, q) on the unsharded graph.  edge_types: a list (walk_len) of per-step
    edge type lists."""
    import ctypes as C
    import numpy as np
    from . import _lib
    L = _lib.lib()
    dev = graph.device
    starts = starts.reshape(-1).to(torch.int64).to(dev).contiguous()
    n = starts.numel()
    walk_len = len(edge_types)
    et = np.ascontiguousarray(np.asarray(edge_types, dtype=np.int32).reshape(walk_len, -1)) \
        if walk_len else np.zeros((0, 1), np.int32)
    k = et.shape[1] if walk_len else 1
    out = torch.empty((n, walk_len + 1), dtype=torch.int64, device=dev)
    stats = (C.c_int64 * 4)()
    limit = dense_table.numel() - 1 if dense_table is not None else 0
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(L.euler_gpu_sharded_node2vec_walk(
            graph._h, transport.ptr() if hasattr(transport, "ptr") else transport, st, graph.seed,
            int(call_id) & 0xFFFFFFFF, C.c_void_p(starts.data_ptr()), n, et.ctypes.data_as(_lib.i32p), k,
            walk_len, float(p), float(q), int(default_node), int(partitions or transport.world),
            C.c_void_p(dense_table.data_ptr()) if dense_table is not None else None, limit,
            C.c_void_p(out.data_ptr()), stats))
    if return_stats:
        return out, {"host_waits": int(stats[0]), "rows_asked": int(stats[1]),
                     "row_entries": int(stats[2]), "ids_sent": int(stats[3])}
    return out
