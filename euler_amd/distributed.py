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


class ShardedSampler:
    """Neighbor sampling over a graph sharded across the ranks of `group`.

    local_sample(roots, root_is_zero_mask, edge_types, count, default_node,
                 call_id) -> (ids [m,count] int64, weights f32, types i32,
                              row_mask [m] uint8)
    must return the TF-layout rows for the roots this rank owns.
    split_fn(ids, partitions, shards) -> (shard_off list, shard_ids, merge_idx)
    and merge_fn(rows, merge_idx) are the bucket / inverse-permutation ops
    (HIP kernels on GPUs).
    """

    def __init__(self, local_sample, split_fn, merge_fn, partitions=None,
                 group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.partitions = partitions or self.world
        self.local_sample = local_sample
        self.split_fn = split_fn
        self.merge_fn = merge_fn

    # -------------------------------------------------------------- helpers
    def _exchange(self, send, send_counts, recv_counts, row_elems=1):
        """all-to-all(v) of rows; counts are in rows."""
        out_rows = int(sum(recv_counts))
        shape = (out_rows,) + tuple(send.shape[1:])
        recv = torch.empty(shape, dtype=send.dtype, device=send.device)
        dist.all_to_all_single(recv, send.contiguous(),
                               output_split_sizes=[int(c) for c in recv_counts],
                               input_split_sizes=[int(c) for c in send_counts],
                               group=self.group)
        return recv

    def sample_neighbor(self, roots, edge_types, count, default_node=-1,
                        call_id=0, root_mask=None, root_group=1):
        """One hop for this rank's `roots` ([n] int64).  root_mask ([n /
        root_group] uint8) marks roots that stand for a missing row of the
        previous hop: they sample as node id 0 (the reference chains hops on
        its core tensors, whose empty rows hold the sentinel 0).
        Returns (ids [n,count], weights, types, row_mask [n])."""
        roots = roots.reshape(-1).to(torch.int64)
        n = roots.numel()
        if root_mask is not None:
            expand = root_mask.to(torch.bool).repeat_interleave(root_group)[:n]
            roots = torch.where(expand, torch.zeros_like(roots), roots)
        # C1: bucket by owner, tell every peer how many ids it gets
        shard_off, shard_ids, merge_idx = self.split_fn(roots, self.partitions,
                                                        self.world)
        send_counts = [int(shard_off[s + 1] - shard_off[s]) for s in range(self.world)]
        sc = torch.tensor(send_counts, dtype=torch.int64, device=roots.device)
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc, group=self.group)
        recv_counts = [int(x) for x in rc.tolist()]
        owned = self._exchange(shard_ids, send_counts, recv_counts)
        # local sampling on the rows this rank owns
        ids, w, t, mask = self.local_sample(owned, edge_types, count, default_node,
                                            call_id)
        # C2: results travel back along the reversed split
        ids_b = self._exchange(ids.reshape(-1, count), recv_counts, send_counts)
        w_b = self._exchange(w.reshape(-1, count), recv_counts, send_counts)
        t_b = self._exchange(t.reshape(-1, count), recv_counts, send_counts)
        m_b = self._exchange(mask.reshape(-1, 1), recv_counts, send_counts)
        # IDX_MERGE / DATA_MERGE: out[merge_idx[j]] = back[j]
        return (self.merge_fn(ids_b, merge_idx), self.merge_fn(w_b, merge_idx),
                self.merge_fn(t_b, merge_idx),
                self.merge_fn(m_b, merge_idx).reshape(-1))

    def sample_fanout(self, roots, edge_types, counts, default_node=-1, call_id=0):
        """Multi-hop fanout (tf_euler sample_fanout): returns (neighbors_list,
        weights_list, types_list) flattened like euler_ops.sample_fanout."""
        roots = roots.reshape(-1).to(torch.int64)
        neighbors, weights, types = [roots], [], []
        mask, group = None, 1
        cur = roots
        for h, count in enumerate(counts):
            ids, w, t, m = self.sample_neighbor(cur, edge_types[h], count,
                                                default_node, call_id + h, mask,
                                                group)
            neighbors.append(ids.reshape(-1))
            weights.append(w.reshape(-1))
            types.append(t.reshape(-1))
            cur, mask, group = ids.reshape(-1), m, count
        return neighbors, weights, types


def gpu_sharded_sampler(graph, partitions=None, group=None):
    """ShardedSampler over an euler_amd.Graph shard living on this rank's GPU."""
    from . import ops

    def local_sample(owned, edge_types, count, default_node, call_id):
        ids, w, t, mask = graph.sample_neighbor(owned, edge_types, count,
                                                default_node, layout="tf",
                                                call_id=call_id, return_mask=True)
        return ids, w, t, mask

    def merge(rows, merge_idx):
        if rows.element_size() * (rows.numel() // max(rows.shape[0], 1)) % 4 != 0:
            # 1-byte mask rows: widen to int32 for the 4-byte merge kernel
            return ops.merge_rows(rows.to(torch.int32), merge_idx).to(rows.dtype)
        return ops.merge_rows(rows, merge_idx)

    return ShardedSampler(local_sample, ops.id_split, merge, partitions, group)
