"""world_size-2 gloo test of the sharded sampling path's host logic
(euler_amd/distributed.py): bucket by owner -> all-to-all ids -> local sample
-> all-to-all results -> inverse permutation.  The local sampler / split /
merge are test doubles backed by the CPU oracle (the product's are HIP
kernels); the result must be bit-identical to the unsharded oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build_csr(O):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import make_random_graph
    rng = np.random.default_rng(4)
    ids, seg, nbr, w, nt, nw = make_random_graph(rng, 600, 3, max_deg=10,
                                                 id_space=4000)
    return O.csr_from_raw(ids, seg, nbr, w, 3, nt, nw), ids


def _shard_csr(O, csr, partitions, rank, world):
    own = O.shard_of(csr.row_id, partitions, world) == rank
    rows = np.nonzero(own)[0]
    T = csr.n_types
    row_ptr = [0]
    nbr, pw, te, tp = [], [], [], []
    for r in rows:
        b, e = csr.row_ptr[r], csr.row_ptr[r + 1]
        nbr.append(csr.nbr[b:e]); pw.append(csr.prefix_w[b:e])
        te.append(csr.type_end[r * T:(r + 1) * T]); tp.append(csr.type_prefix[r * T:(r + 1) * T])
        row_ptr.append(row_ptr[-1] + (e - b))
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
    return O.CSR(csr.row_id[rows], np.array(row_ptr, np.int64), cat(te, np.int32),
                 cat(nbr, np.uint64), cat(pw, np.float32), cat(tp, np.float32), T,
                 csr.node_type[rows], csr.node_weight[rows])


def _worker(rank, world, port, partitions, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from euler_amd.distributed import ShardedSampler, owner_of
    csr, ids = _build_csr(O)
    OG_local = O.OracleGraph(_shard_csr(O, csr, partitions, rank, world))
    OG_full = O.OracleGraph(csr)
    seed = 99

    def local_sample(owned, edge_types, count, default_node, call_id):
        q = owned.numpy().astype(np.int64)
        # every id routed here must be owned by this rank
        assert np.all(O.shard_of(q.astype(np.uint64), partitions, world) == rank)
        n, w, t = OG_local.sample_neighbor(seed, call_id, q, edge_types, count,
                                           default_node)
        _, cid, _, _ = OG_local.sample_neighbor_core(seed, call_id, q.astype(np.uint64),
                                                     edge_types, count)
        mask = (cid.reshape(len(q), count)[:, 0] == 0).astype(np.uint8) if count else \
            np.zeros(len(q), np.uint8)
        return (torch.as_tensor(n), torch.as_tensor(w), torch.as_tensor(t),
                torch.as_tensor(mask))

    def split_fn(roots, parts, shards):
        off, sid, mi = O.id_split(roots.numpy().astype(np.uint64), parts, shards)
        return off.tolist(), torch.as_tensor(sid.astype(np.int64)), torch.as_tensor(mi)

    def merge_fn(rows, merge_idx):
        out = torch.empty_like(rows)
        out[merge_idx.long()] = rows
        return out

    def unique_fn(ids):
        uq, gi = O.id_unique(ids.numpy().astype(np.uint64))
        return torch.as_tensor(uq.astype(np.int64)), torch.as_tensor(gi)

    def gather_fn(rows, gather_idx):
        return rows[gather_idx.long()]

    def dedup_split_fn(ids, parts, shards, root_mask=None, root_group=1):
        if root_mask is not None:
            m = root_mask.numpy().astype(bool).repeat(root_group)[:ids.numel()]
            ids = torch.where(torch.as_tensor(m), torch.zeros_like(ids), ids)
        uq, gi = O.id_unique(ids.numpy().astype(np.uint64))
        off, sid, mi = O.id_split(uq, parts, shards)
        inv = np.empty(len(uq), np.int64)
        inv[mi] = np.arange(len(uq))
        return off.tolist(), torch.as_tensor(sid.astype(np.int64)), torch.as_tensor(inv[gi])

    def expand_fn(pos, ids, w, t, mask, count):
        p = pos.long()
        return ids[p], w[p], t[p], mask[p]

    S_fused = ShardedSampler(local_sample, split_fn, merge_fn, partitions,
                             dedup_split_fn=dedup_split_fn, expand_fn=expand_fn)
    # the fused sampler gets its peer counts through the shared-memory mailbox
    # (the ranks are processes of one node), the other two through the collective
    from euler_amd.distributed import ShmCounts
    shm = ShmCounts()
    assert shm.ok, "shared-memory mailbox did not come up"
    assert not os.path.exists("/dev/shm" + shm.name), "the region's name must not outlive setup"
    for rnd in range(200):          # many rounds back to back, uneven pace
        msg = [rank * 100000 + rnd * 10 + p for p in range(world)]
        assert shm(msg) == [p * 100000 + rnd * 10 + rank for p in range(world)]
    S_fused.counts_fn = shm

    # the branch the GPU sampler takes: the shard samples straight into wire rows
    # (4 * count + 2 int32 words: ids | weights | types | mask, pad) and the
    # requester expands them by position
    def local_sample_packed(owned, edge_types, count, default_node, call_id):
        ids_, w_, t_, m_ = local_sample(owned, edge_types, count, default_node, call_id)
        m = ids_.shape[0]
        cols = 3 if len(edge_types) == 1 else 4        # one listed type: no type column
        rows = torch.zeros((m, (cols * count + 3) & ~1), dtype=torch.int32)
        if m == 0:                # a shard nobody asked anything (world 8, small batches)
            return rows
        rows[:, :2 * count] = ids_.reshape(m, count).contiguous().view(torch.int32)
        rows[:, 2 * count:3 * count] = w_.reshape(m, count).contiguous().view(torch.int32)
        if cols == 4:
            rows[:, 3 * count:4 * count] = t_.reshape(m, count)
        rows[:, cols * count] = m_.reshape(m).to(torch.int32)
        return rows

    def expand_packed(pos, rows, count, single_type=None):
        cols = 3 if single_type is not None else 4
        assert rows.shape[1] == (cols * count + 3) & ~1
        r = rows[pos.long()]
        mask_ = r[:, cols * count].to(torch.uint8)
        if cols == 4:
            types_ = r[:, 3 * count:4 * count].contiguous()
        else:
            types_ = torch.where(mask_.to(torch.bool)[:, None],
                                 torch.full((len(r), count), -1, dtype=torch.int32),
                                 torch.full((len(r), count), int(single_type), dtype=torch.int32))
        return (r[:, :2 * count].contiguous().view(torch.int64),
                r[:, 2 * count:3 * count].contiguous().view(torch.float32), types_, mask_)

    S_packed = ShardedSampler(local_sample, split_fn, merge_fn, partitions,
                              dedup_split_fn=dedup_split_fn, expand_fn=expand_packed)
    S_packed.local_sample_packed = local_sample_packed
    S_packed.counts_fn = shm
    # with and without the duplicate-root removal: both must equal the
    # unsharded oracle
    S_plain = ShardedSampler(local_sample, split_fn, merge_fn, partitions)
    S = ShardedSampler(local_sample, split_fn, merge_fn, partitions,
                       unique_fn=unique_fn, gather_fn=gather_fn)
    rng = np.random.default_rng(10 + rank)
    roots = np.concatenate([rng.choice(ids, 257 + 64 * rank),
                            [0, 5, 2 ** 63 + 11]]).astype(np.uint64).astype(np.int64)
    # owner_of agrees with the reference's unsigned modulo
    assert np.array_equal(owner_of(torch.as_tensor(roots), partitions, world).numpy(),
                          O.shard_of(roots.astype(np.uint64), partitions, world))
    for et, counts in (([[0, 1, 2], [0, 1, 2]], [5, 3]), ([[1], [2]], [4, 2]),
                       ([[0, 2], [2, 1]], [3, 3])):
        on, ow, ot = OG_full.sample_fanout(seed, 40, roots, et, counts, -1)
        for sampler in (S, S_plain, S_fused, S_packed):
            ns, ws, ts = sampler.sample_fanout(torch.as_tensor(roots), et, counts, -1, 40)
            for h in range(len(counts)):
                assert np.array_equal(ns[h + 1].numpy(), on[h]), (rank, et, h)
                assert np.array_equal(ws[h].numpy(), ow[h])
                assert np.array_equal(ts[h].numpy(), ot[h])
    # the typed draws of a heterogeneous minibatch over one front end / id exchange
    # (sample_neighbor_sets): == one sample_neighbor per set, fused samplers and the fallback
    sets = [[1], [0, 2], [0, 1, 2]]
    for sampler in (S_packed, S_plain):
        outs = sampler.sample_neighbor_sets(torch.as_tensor(roots), sets, 4, -1, 70)
        for c, et in enumerate(sets):
            on, ow, ot = OG_full.sample_neighbor(seed, 70 + c, roots, et, 4, -1)
            assert np.array_equal(outs[c][0].numpy().reshape(-1), on.reshape(-1)), (rank, et)
            assert np.array_equal(outs[c][1].numpy().reshape(-1), ow.reshape(-1))
            assert np.array_equal(outs[c][2].numpy().reshape(-1), ot.reshape(-1))
    # several minibatches in flight from one thread (run_interleaved): a two-phase
    # front end (begin = enqueue, end = wait) lets the hops of different batches
    # alternate; every rank must still issue the same sequence of collectives -
    # also when one rank's batch is empty
    from euler_amd.distributed import run_interleaved
    S_two = ShardedSampler(local_sample, split_fn, merge_fn, partitions,
                           dedup_split_fn=dedup_split_fn, expand_fn=expand_packed)
    S_two.local_sample_packed = local_sample_packed
    S_two.counts_fn = shm
    S_two.front_begin_fn = lambda ids_, parts, shards, rm, rg: dedup_split_fn(ids_, parts, shards,
                                                                            rm, rg)
    S_two.front_end_fn = lambda token: token
    # (world 8: ranks 0, 3 and 6 bring an EMPTY batch to the third minibatch - every rank must still
    # issue the same sequence of collectives)
    batches = [roots, roots[::2], roots[:0] if rank % 3 == 0 else roots[:7], roots[5:90], roots[::3]]
    et_i, cnt_i = [[0, 1, 2], [0, 1, 2]], [4, 3]
    for in_flight in (2, 3):
        got = run_interleaved(lambda j: S_two.sample_fanout_steps(
            torch.as_tensor(batches[j]), et_i, cnt_i, -1, 100 + 2 * j), len(batches), in_flight)
        for j, b in enumerate(batches):
            on, ow, ot = OG_full.sample_fanout(seed, 100 + 2 * j, b, et_i, cnt_i, -1)
            for h in range(2):
                assert np.array_equal(got[j][0][h + 1].numpy(), on[h]), (rank, j, h)
                assert np.array_equal(got[j][1][h].numpy(), ow[h])
                assert np.array_equal(got[j][2][h].numpy(), ot[h])
    # ... and the typed draws of several heterogeneous minibatches in flight
    # (sample_neighbor_sets_steps: one yield per minibatch, at the front end's wait)
    for in_flight in (2, 3):
        got = run_interleaved(lambda j: S_two.sample_neighbor_sets_steps(
            torch.as_tensor(batches[j]), sets, 4, -1, 300 + 3 * j), len(batches), in_flight)
        for j, b in enumerate(batches):
            for c, et in enumerate(sets):
                on, ow, ot = OG_full.sample_neighbor(seed, 300 + 3 * j + c, b, et, 4, -1)
                assert np.array_equal(got[j][c][0].numpy().reshape(-1), on.reshape(-1)), (rank, j, et)
                assert np.array_equal(got[j][c][1].numpy().reshape(-1), ow.reshape(-1))
                assert np.array_equal(got[j][c][2].numpy().reshape(-1), ot.reshape(-1))
    # calls without an explicit call_id draw fresh ids from the sampler's counter
    # (by hops for a fanout): two consecutive default calls differ, and equal the
    # explicit calls a single-GPU Graph would have made
    S.set_call_id(700)
    d1 = S.sample_fanout(torch.as_tensor(roots), et_i, cnt_i, -1)
    d2 = S.sample_fanout(torch.as_tensor(roots), et_i, cnt_i, -1)
    d3 = S.sample_neighbor(torch.as_tensor(roots), [0, 1, 2], 3, -1)
    assert not np.array_equal(d1[0][1].numpy(), d2[0][1].numpy())
    for got_d, cid in ((d1, 700), (d2, 702)):
        on, _, _ = OG_full.sample_fanout(seed, cid, roots, et_i, cnt_i, -1)
        assert np.array_equal(got_d[0][1].numpy(), on[0]) and np.array_equal(got_d[0][2].numpy(), on[1])
    on3, _, _ = OG_full.sample_neighbor(seed, 704, roots, [0, 1, 2], 3, -1)
    assert np.array_equal(d3[0].numpy().reshape(-1), np.asarray(on3).reshape(-1))
    # DeepWalk (config 4): p = q = 1 random walk over the sharded graph, one
    # exchange per step, identical to the unsharded walk
    L = 6
    et_walk = [[0, 1, 2]] * L
    walk = S_fused.random_walk(torch.as_tensor(roots), et_walk, default_node=-1, call_id=70)
    ref = OG_full.random_walk(seed, 70, roots, et_walk, L, 1.0, 1.0, -1)
    assert np.array_equal(walk.numpy(), ref), rank
    # ---- dense features of remote nodes: same exchange, rows of floats
    per = [[list(np.float32([i, i + 0.5, -i])), list(np.float32([2 * i]))] for i in
           range(csr.n_rows)]
    F_full = O.DenseFeatures.from_lists(per)
    own_rows = np.nonzero(O.shard_of(csr.row_id, partitions, world) == rank)[0]
    F_local = O.DenseFeatures.from_lists([per[i] for i in own_rows])

    def local_feature(owned, fids, dims):
        return [torch.as_tensor(x) for x in
                OG_local.get_dense_feature(F_local, owned.numpy(), fids, dims)]

    S_fused.local_feature = local_feature
    S_fused.row_gather_fn = lambda rows, pos: rows[pos.long()]
    S_plain.local_feature = local_feature
    S_plain.row_gather_fn = lambda rows, pos: rows[pos.long()]
    want_f = OG_full.get_dense_feature(F_full, roots, [0, 1, 3], [3, 2, 2])
    for sampler in (S_fused, S_plain):
        got_f = sampler.get_dense_feature(torch.as_tensor(roots), [0, 1, 3], [3, 2, 2])
        for g_, w_ in zip(got_f, want_f):
            assert np.array_equal(g_.numpy(), w_)
    # ---- full neighbours: variable-length rows, IDX_MERGE / DATA_MERGE semantics
    def local_full_neighbor(owned, et):
        i_, d_, w_, t_ = OG_local.get_full_neighbor(owned.numpy().astype(np.uint64), et)
        return (torch.as_tensor(i_), torch.as_tensor(d_.astype(np.int64)), torch.as_tensor(w_),
                torch.as_tensor(t_))

    def idx_gather_fn(idx, gi):
        o = O.idx_gather(idx.numpy(), gi.numpy())
        return torch.as_tensor(o), int(o[-1, 1]) if len(o) else 0

    def data_gather_fn(data, idx, gi):
        return torch.as_tensor(O.data_gather(data.numpy(), idx.numpy(), gi.numpy()))

    for sampler in (S_fused, S_plain):
        sampler.local_full_neighbor = local_full_neighbor
        sampler.idx_gather_fn = idx_gather_fn
        sampler.data_gather_fn = data_gather_fn
        for et_ in ([0, 1, 2], [2, 0], [1]):
            gi_, gd_, gw_, gt_ = sampler.get_full_neighbor(torch.as_tensor(roots), et_)
            wi_, wd_, ww_, wt_ = OG_full.get_full_neighbor(roots.astype(np.uint64), et_)
            assert np.array_equal(gi_.numpy(), wi_), (rank, et_)
            assert np.array_equal(gd_.numpy().astype(np.uint64), wd_)
            assert np.array_equal(gw_.numpy(), ww_) and np.array_equal(gt_.numpy(), wt_)
    # ---- node2vec (config 4's biased walk) over the sharded graph: the requester fetches
    # the rows of the walkers' nodes step by step and draws on them, as the reference's
    # client does (random_walk_op.cc:83-168); equal to the unsharded walk
    def n2v_step_fn(call_id, c_row, c_idx, c_ids, c_w, p_row, p_idx, p_ids, parent, p_, q_, dn):
        f = lambda x: None if x is None else x.numpy()
        return torch.as_tensor(O.node2vec_step_lists(seed, call_id, f(c_row), f(c_idx), f(c_ids),
                                                     f(c_w), f(p_row), f(p_idx), f(p_ids),
                                                     f(parent), p_, q_, dn))
    starts = roots[:60 + 5 * rank]
    for sampler in (S_fused, S_plain):
        sampler.n2v_step_fn = n2v_step_fn
        for (p_, q_, et_w) in ((0.25, 4.0, [[0, 1, 2]] * 4), (2.0, 0.5, [[1, 0], [0, 2], [2, 1]]),
                               (1.0, 1.0 + 1e-7, [[0, 1, 2]] * 2)):
            got_w = sampler.random_walk(torch.as_tensor(starts), et_w, p_, q_, default_node=-3,
                                        call_id=300)
            want_w = OG_full.random_walk(seed, 300, starts, et_w, len(et_w), p_, q_, -3)
            assert np.array_equal(got_w.numpy(), want_w), (rank, p_, q_)
    # ---- sparse (uint64) features: variable-length rows again, then the TF
    # kernel's default entries on the requester
    sper = [[[int(v)] * (i % 4), [int(v) + 7, 3][: (i % 3)], [2 ** 63 + i]] if i % 5 else [[]]
            for i, v in enumerate(csr.row_id)]
    SF_full = O.SparseFeatures.from_lists(sper)
    SF_local = O.SparseFeatures.from_lists([sper[i] for i in own_rows])
    local_rows = {int(v): k for k, v in enumerate(csr.row_id[own_rows])}

    def local_sparse_feature(owned, fid):
        idx = np.zeros((owned.numel(), 2), np.int32)
        vals = []
        U = SF_local.n_u64
        for j, v in enumerate(owned.numpy().astype(np.uint64)):
            idx[j, 0] = len(vals)
            r = local_rows.get(int(v))
            if r is not None and 0 <= fid < U:
                fi = SF_local.feat_idx[r * U:(r + 1) * U]
                pre = 0 if fid == 0 else fi[fid - 1]
                b0 = SF_local.feat_ptr[r]
                vals.extend(SF_local.feat_val[b0 + pre:b0 + fi[fid]].view(np.int64).tolist())
            idx[j, 1] = len(vals)
        return torch.as_tensor(idx), torch.as_tensor(np.array(vals, np.int64))

    q_sp = np.concatenate([roots, [0, 123456789]]).astype(np.int64)
    want_s = OG_full.get_sparse_feature(SF_full, q_sp.astype(np.uint64), [0, 1, 2, 5], [0, 9, -1, 4])
    for sampler in (S_fused, S_plain):
        sampler.local_sparse_feature = local_sparse_feature
        got_s = sampler.get_sparse_feature(torch.as_tensor(q_sp), [0, 1, 2, 5], [0, 9, -1, 4])
        for (gi_, gv_, gs_), (wi_, wv_, ws_) in zip(got_s, want_s):
            assert np.array_equal(gi_.numpy(), wi_) and np.array_equal(gv_.numpy(), wv_)
            assert list(gs_) == list(ws_)
    # ---- layerwise sampling: sums by id exchange, local root draw, position-keyed
    # layer draw on the owners, adjacency from the fetched rows
    def local_edge_sum_weight(owned, et):
        return torch.as_tensor(OG_local.get_edge_sum_weight(owned.numpy().astype(np.uint64), et))

    def sample_root_fn(r_, w_, m_, dn_, call_id):
        out = O.sample_root(seed, call_id, r_.numpy().astype(np.uint64), w_.numpy(), r_.shape[1],
                            m_, dn_)
        return torch.as_tensor(out.view(np.int64).reshape(r_.shape[0], m_))

    def local_sample_layer(ids_, pos_, et, dn_, call_id):
        q_ = ids_.numpy().astype(np.uint64)
        assert np.all(O.shard_of(q_, partitions, world) == rank)
        a_, b_, c_ = OG_local.sample_layer(seed, call_id, q_, et, dn_, positions=pos_.numpy())
        return torch.as_tensor(a_.view(np.int64)), torch.as_tensor(b_), torch.as_tensor(c_)

    # SparseGetAdj answered by the owners (hit masks ORed on the requester) for one sampler,
    # by fetched rows for the other: both must give the unsharded triple
    def local_adj_mask(q_nodes, q_nb, b_, n_, m_, et):
        words = (m_ + 63) // 64
        mk = np.zeros((b_ * n_, words), np.uint64)
        if b_ * n_ and m_:
            idx_, ids_, _w, _t = OG_local.get_full_neighbor(q_nodes.numpy().astype(np.uint64), et)
            nbv = q_nb.numpy().astype(np.uint64).reshape(b_, m_)
            for r_ in range(b_ * n_):
                row = set(ids_[idx_[r_, 0]:idx_[r_, 1]].tolist())
                for c_ in range(m_):
                    if int(nbv[r_ // n_, c_]) in row:
                        mk[r_, c_ // 64] |= np.uint64(1) << np.uint64(c_ % 64)
        return torch.as_tensor(mk.view(np.int64))

    def adj_from_mask_fn(mask, b_, n_, m_):
        mk = mask.numpy().view(np.uint64).reshape(b_ * n_, -1)
        ind, val = [], []
        for r_ in range(b_ * n_):
            for c_ in range(m_):
                hit = bool((mk[r_, c_ // 64] >> np.uint64(c_ % 64)) & np.uint64(1))
                if hit or (r_ % n_ == n_ - 1 and c_ == m_ - 1):
                    ind.append([r_ // n_, r_ % n_, c_]); val.append(1 if hit else 0)
        if not ind:
            return torch.zeros((0, 3), dtype=torch.int64), torch.zeros(0, dtype=torch.int64), [0, 0, 0]
        return torch.as_tensor(np.array(ind, np.int64)), torch.as_tensor(np.array(val, np.int64)), [b_, n_, m_]

    S_fused.local_adj_mask = local_adj_mask
    S_fused.adj_from_mask_fn = adj_from_mask_fn
    rng_l = np.random.default_rng(50 + rank)       # every rank asks for its own minibatch
    for sampler in (S_fused, S_plain):
        sampler.local_edge_sum_weight = local_edge_sum_weight
        sampler.sample_root_fn = sample_root_fn
        sampler.local_sample_layer = local_sample_layer
        for batch_l, n_l, cnt_l, et_l in ((3, 4, 6, [0, 1, 2]), (1, 30, 12, [1]), (5, 1, 2, [2, 0])):
            nodes_l = rng_l.choice(ids, (batch_l, n_l)).astype(np.uint64)
            nodes_l[0, 0] = nodes_l[0, -1]
            got_nb, (gi_, gv_, gs_) = sampler.sample_neighbor_layerwise(
                torch.as_tensor(nodes_l.view(np.int64)), et_l, cnt_l, -1, call_id=33)
            wnb, wi_, wv_, ws_ = OG_full.sample_neighbor_layerwise(seed, 33, nodes_l, et_l,
                                                                   cnt_l, -1)
            assert np.array_equal(got_nb.numpy(), wnb), (rank, batch_l, n_l)
            assert np.array_equal(gi_.numpy(), wi_) and np.array_equal(gv_.numpy(), wv_)
            assert list(gs_) == list(ws_)
            # ... and with a weight function: rows fetched, API_LOCAL_SAMPLE_L on the requester
            def local_layer_fn(idx_, ids_, w_, t_, b_, n_, m_, wf_, dn_, call_id):
                a_, bw_, c_ = OG_full.local_sample_layer(
                    seed, call_id, idx_.numpy(), ids_.numpy().astype(np.uint64), w_.numpy(),
                    t_.numpy(), n_, m_, wf_, dn_)
                return torch.as_tensor(a_.view(np.int64)), torch.as_tensor(bw_), torch.as_tensor(c_)
            sampler.local_layer_fn = local_layer_fn
            got_nb, (gi_, gv_, gs_) = sampler.sample_neighbor_layerwise(
                torch.as_tensor(nodes_l.view(np.int64)), et_l, cnt_l, -1, call_id=34,
                weight_func="sqrt")
            wf_want = OG_full.sample_neighbor_layerwise_func(seed, 34, nodes_l, et_l, cnt_l,
                                                             "sqrt", -1)
            assert np.array_equal(got_nb.numpy(), wf_want[0])
            assert np.array_equal(gi_.numpy(), wf_want[3]) and np.array_equal(gv_.numpy(), wf_want[4])
    # ---- SampleNode over the shards: SAMPLE_NODE_SPLIT + local draws + APPEND_MERGE
    OG_local.build_node_sampler()
    shard_graphs = [O.OracleGraph(_shard_csr(O, csr, partitions, r, world)) for r in range(world)]
    for g_ in shard_graphs:
        g_.build_node_sampler()

    def type_sum(g_, node_type):
        tot = np.float32(0)
        for t_, w_ in zip(g_.csr.node_type, g_.csr.node_weight):
            if node_type == -1 or t_ == node_type:
                tot = np.float32(tot + w_)
        return float(tot)

    S_fused.local_sample_node = lambda count, nt, call_id: torch.as_tensor(
        OG_local.sample_node(seed, call_id, [nt], count).astype(np.int64))
    S_fused.node_weight_sum = lambda nt: type_sum(OG_local, nt)
    S_fused.node_split_fn = lambda call_id, count, w: O.sample_node_split(seed, call_id, count, w)
    for nt, count in ((-1, 37), (0, 20), (1, 5)):
        got_n = S_fused.sample_node(count, nt, call_id=90).numpy()
        w = np.array([type_sum(g_, nt) for g_ in shard_graphs], np.float32)
        tot = w[0]
        for x in w[1:]:
            tot = np.float32(tot + x)
        split = O.sample_node_split(seed, 90, count, list(w) + [tot])
        want_n = np.concatenate([g_.sample_node(seed, 90, [nt], int(c)).astype(np.int64)
                                 for g_, c in zip(shard_graphs, split)])
        assert int(split.sum()) == count and np.array_equal(got_n, want_n), (nt, count)
    # empty request from one rank must not dead-lock the exchange
    empty = torch.zeros(0, dtype=torch.int64) if rank == 0 else torch.as_tensor(roots)
    want_e = OG_full.sample_neighbor(seed, 7, empty.numpy(), [0], 2, -1)
    for sampler in (S, S_fused, S_packed):
        n1, w1, t1, m1 = sampler.sample_neighbor(empty, [0], 2, -1, 7)
        assert n1.shape[0] == empty.numel()
        assert np.array_equal(n1.numpy().reshape(-1), want_e[0].reshape(-1))
    dist.barrier()
    open(os.path.join(out_dir, "ok_%d" % rank), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,partitions", [(2, 2), (2, 8), (3, 6), (8, 8)])
def test_sharded_fanout_matches_unsharded_gloo(O, tmp_path, world, partitions):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, partitions, str(tmp_path)), nprocs=world,
             join=True)
    for r in range(world):
        assert os.path.exists(tmp_path / ("ok_%d" % r))


def _rehearsal_worker(rank, world, port, partitions, out_dir):
    """An N-rank dress rehearsal on CPU doubles: 8 ranks, partitions = 1024 (the
    reference's file partitions: owner = (id % 1024) % 8, core/kernels/id_split_op.cc:46-49,
    core/graph/graph.cc:90-98), a rank whose shard is EMPTY, a batch whose roots one rank owns
    alone, ranks with an empty batch - every rank must issue the same sequence of collectives and
    the results must equal the unsharded graph's."""
    import json
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from euler_amd.distributed import ShardedSampler
    csr_all, _ids = _build_csr(O)
    # the graph without the rows rank `world - 1` would own: that shard is empty, ids that
    # pointed there are now unknown nodes (their owner answers default rows)
    keep = O.shard_of(csr_all.row_id, partitions, world) != world - 1
    rows = np.nonzero(keep)[0]
    T = csr_all.n_types
    rp, nbr, pw, te, tp = [0], [], [], [], []
    for r in rows:
        b, e = csr_all.row_ptr[r], csr_all.row_ptr[r + 1]
        nbr.append(csr_all.nbr[b:e]); pw.append(csr_all.prefix_w[b:e])
        te.append(csr_all.type_end[r * T:(r + 1) * T]); tp.append(csr_all.type_prefix[r * T:(r + 1) * T])
        rp.append(rp[-1] + (e - b))
    csr = O.CSR(csr_all.row_id[rows], np.array(rp, np.int64), np.concatenate(te).astype(np.int32),
                np.concatenate(nbr).astype(np.uint64), np.concatenate(pw).astype(np.float32),
                np.concatenate(tp).astype(np.float32), T, csr_all.node_type[rows], csr_all.node_weight[rows])
    shard = _shard_csr(O, csr, partitions, rank, world)
    assert (len(shard.row_id) == 0) == (rank == world - 1)
    OG_local, OG_full = O.OracleGraph(shard), O.OracleGraph(csr)
    seed = 31

    def local_sample(owned, edge_types, count, default_node, call_id):
        q = owned.numpy().astype(np.int64)
        assert np.all(O.shard_of(q.astype(np.uint64), partitions, world) == rank)
        n, w, t = OG_local.sample_neighbor(seed, call_id, q, edge_types, count, default_node)
        _, cid, _, _ = OG_local.sample_neighbor_core(seed, call_id, q.astype(np.uint64), edge_types, count)
        mask = (cid.reshape(len(q), count)[:, 0] == 0).astype(np.uint8) if count else np.zeros(len(q), np.uint8)
        return torch.as_tensor(n), torch.as_tensor(w), torch.as_tensor(t), torch.as_tensor(mask)

    def split_fn(roots, parts, shards):
        off, sid, mi = O.id_split(roots.numpy().astype(np.uint64), parts, shards)
        return off.tolist(), torch.as_tensor(sid.astype(np.int64)), torch.as_tensor(mi)

    def merge_fn(rows_, merge_idx):
        out = torch.empty_like(rows_)
        out[merge_idx.long()] = rows_
        return out

    def dedup_split_fn(ids, parts, shards, root_mask=None, root_group=1):
        if root_mask is not None:
            m = root_mask.numpy().astype(bool).repeat(root_group)[:ids.numel()]
            ids = torch.where(torch.as_tensor(m), torch.zeros_like(ids), ids)
        uq, gi = O.id_unique(ids.numpy().astype(np.uint64))
        off, sid, mi = O.id_split(uq, parts, shards)
        inv = np.empty(len(uq), np.int64)
        inv[mi] = np.arange(len(uq))
        return off.tolist(), torch.as_tensor(sid.astype(np.int64)), torch.as_tensor(inv[gi])

    def expand_fn(pos, ids, w, t, mask, count):
        p = pos.long()
        return ids[p], w[p], t[p], mask[p]

    S = ShardedSampler(local_sample, split_fn, merge_fn, partitions, dedup_split_fn=dedup_split_fn,
                       expand_fn=expand_fn)
    S.collective_log = []
    rng = np.random.default_rng(1000 + rank)
    all_ids = csr.row_id.astype(np.int64)
    owner = O.shard_of(csr.row_id, partitions, world)
    batches = [
        rng.choice(all_ids, 300 + 37 * rank),                              # ragged sizes
        rng.choice(all_ids[owner == 5], 256),                              # one rank owns every root
        rng.choice(all_ids, 200) if rank % 3 else np.zeros(0, np.int64),   # some ranks bring nothing
        np.concatenate([rng.choice(all_ids, 50), [0, 2 ** 40 + 7]]),       # unknown ids
    ]
    for b, roots in enumerate(batches):
        for et, counts in (([[0], [1]], [5, 3]), ([[0, 1, 2], [2, 0]], [4, 2])):
            k = max(len(x) for x in et)
            et_pad = [list(x) + [x[-1]] * (k - len(x)) for x in et] if len({len(x) for x in et}) > 1 else et
            got = S.sample_fanout(torch.as_tensor(roots), et_pad, counts, -1, call_id=10 * b)
            want = OG_full.sample_fanout(seed, 10 * b, roots, et_pad, counts, -1)
            for h in range(2):
                assert np.array_equal(got[0][h + 1].numpy().reshape(-1), want[0][h]), (rank, b, h)
                assert np.array_equal(got[1][h].numpy().reshape(-1), want[1][h]), (rank, b, h)
        walk = S.random_walk(torch.as_tensor(roots), [[0, 1, 2]] * 5, default_node=-1, call_id=500 + 10 * b)
        assert np.array_equal(walk.numpy(), OG_full.random_walk(seed, 500 + 10 * b, roots, [[0, 1, 2]] * 5, 5,
                                                                1.0, 1.0, -1)), (rank, b)
    with open(os.path.join(out_dir, "collectives_rank%d.json" % rank), "w") as f:
        json.dump(S.collective_log, f)
    dist.barrier()
    dist.destroy_process_group()
    with open(os.path.join(out_dir, "rehearsal_ok_%d" % rank), "w") as f:
        f.write("ok")


def test_eight_rank_rehearsal_partitions_1024(O, tmp_path):
    import json
    world = 8
    port = _free_port()
    mp.spawn(_rehearsal_worker, args=(world, port, 1024, str(tmp_path)), nprocs=world, join=True)
    logs = []
    for r in range(world):
        assert os.path.exists(os.path.join(str(tmp_path), "rehearsal_ok_%d" % r))
        logs.append(json.load(open(os.path.join(str(tmp_path), "collectives_rank%d.json" % r))))
    # the same sequence of collectives on every rank, whatever its batch held: 4 batches x (2
    # fanouts x 2 hops + 5 walk steps) x (counts, ids out, rows back)
    assert all(l == logs[0] for l in logs[1:]), "ranks issued different collective sequences"
    assert len(logs[0]) == 4 * (2 * 2 + 5) * 3
    assert [k for k, _ in logs[0][:3]] == ["counts", "alltoallv", "alltoallv"]


def _mailbox_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from euler_amd.distributed import ShmCounts
    import time
    shm = ShmCounts()
    assert shm.ok
    rng = np.random.default_rng(rank)
    for rnd in range(3000):
        if rng.random() < 0.01:
            time.sleep(0.001 * rng.random())         # ranks drift apart by whole rounds' worth
        msg = [(rank << 40) | (rnd << 8) | p for p in range(world)]
        assert shm(msg) == [(p << 40) | (rnd << 8) | rank for p in range(world)], (rank, rnd)
    shm.close()
    dist.barrier()
    open(os.path.join(out_dir, "ok_%d" % rank), "w").write("ok")
    dist.destroy_process_group()


def test_shared_memory_counts_mailbox_world8(tmp_path):
    """euler_shm_* (the per-hop peer counts of the multi-GPU sampler) with 8
    processes: 3000 all-to-all rounds back to back with ranks drifting apart,
    every message arrives once, in order, from the right peer."""
    world = 8
    port = _free_port()
    mp.spawn(_mailbox_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert os.path.exists(tmp_path / ("ok_%d" % r))


def test_adj_from_rows_and_sparse_from_core_vs_oracle(O):
    """The requester-side assembly of the sharded path (pure torch, runs on the
    CPU here): adjacency triple from fetched rows, sparse-feature triple from the
    GQL values() layout - against the oracle on random inputs."""
    sys.path.insert(0, ROOT)
    from euler_amd.distributed import adj_from_rows, sparse_from_core
    csr, ids = _build_csr(O)
    OG = O.OracleGraph(csr)
    rng = np.random.default_rng(12)
    for batch, n, m in ((1, 1, 1), (3, 4, 9), (2, 17, 70), (5, 1, 3), (1, 40, 200)):
        nodes = rng.choice(ids, (batch, n)).astype(np.uint64)
        nodes[0, 0] = 2 ** 62 + 5
        cand = rng.choice(ids, (batch, m)).astype(np.uint64)
        for b in range(batch):
            nb = OG.get_full_neighbor(nodes[b], [0, 1, 2])[1]
            if len(nb):
                take = rng.choice(nb, m // 2 + 1)
                cand[b, :len(take)] = take[:m]
        cand[-1, 0] = cand[-1, m - 1]
        for et in ([0], [2, 1], [0, 1, 2], []):
            idx, fid, _, _ = OG.get_full_neighbor(nodes.reshape(-1), et)
            got = adj_from_rows(torch.as_tensor(idx), torch.as_tensor(fid.view(np.int64)),
                                torch.as_tensor(cand.view(np.int64)), batch, n, m)
            want = OG.sparse_get_adj_tf(nodes, cand, et, n, m)
            assert np.array_equal(got[0].numpy(), want[0]) and np.array_equal(got[1].numpy(), want[1])
            assert list(got[2]) == list(want[2])
    assert adj_from_rows(torch.zeros((0, 2), dtype=torch.int32), torch.zeros(0, dtype=torch.int64),
                         torch.zeros(0, dtype=torch.int64), 0, 3, 4)[2] == [0, 0, 0]
    per = [[list(rng.integers(0, 2 ** 63, int(rng.integers(0, 4)), dtype=np.uint64)), [int(v)] * (i % 3)]
           if i % 4 else [[]] for i, v in enumerate(csr.row_id)]
    SF = O.SparseFeatures.from_lists(per)
    q = np.concatenate([rng.choice(ids, 300), [0, 77777777]]).astype(np.uint64)
    U = SF.n_u64
    pos = {int(v): i for i, v in enumerate(csr.row_id)}
    for fid, dv in ((0, 0), (1, 9), (3, -1)):
        idx = np.zeros((len(q), 2), np.int32)
        vals = []
        for j, v in enumerate(q):
            idx[j, 0] = len(vals)
            r = pos.get(int(v))
            if r is not None and 0 <= fid < U:
                fi = SF.feat_idx[r * U:(r + 1) * U]
                pre = 0 if fid == 0 else fi[fid - 1]
                vals.extend(SF.feat_val[SF.feat_ptr[r] + pre:SF.feat_ptr[r] + fi[fid]].view(np.int64).tolist())
            idx[j, 1] = len(vals)
        got = sparse_from_core(torch.as_tensor(idx), torch.as_tensor(np.array(vals, np.int64)), dv)
        want = OG.get_sparse_feature(SF, q, [fid], [dv])[0]
        assert np.array_equal(got[0].numpy(), want[0]) and np.array_equal(got[1].numpy(), want[1])
        assert list(got[2]) == list(want[2])
