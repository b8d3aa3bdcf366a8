"""The multi-GPU sampler with world > 1 and the REAL HIP pieces, on one GPU.

Two (or three) processes share cuda:0.  Each builds its shard of the graph
(euler_gpu_graph_create_shard / _create_synthetic with shard_index = rank), and a
hop runs euler_gpu_dedup_split -> id exchange -> euler_gpu_sample_neighbor_packed
on the owner -> row exchange -> euler_gpu_expand_packed.  RCCL refuses two ranks
on one device, so the transport is gloo over host-staged tensors
(ShardedSampler.host_staged); everything else is the code an N-GPU run executes.
The sharded result must equal the unsharded GPU graph's and the oracle's, bit
for bit: owner(id) = (id % partitions) % shards as core/kernels/id_split_op.cc:46-49,
merge order as idx_merge_op.cc:32-78 / data_merge_op.cc:44-67.

Covers BASELINE configs 3 (fanout [25,10]), 4 (random_walk length 40) and 5
(typed sampling: one listed type, 3 of 8, all) in their sharded form, empty
buckets, a rank with an empty batch, non-self exchanges and both id maps."""
import os
import socket
import sys
import traceback

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def t2n(t):
    return t.detach().cpu().numpy()


def _same(got, want, what):
    import torch
    for h, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a.reshape(-1), b.reshape(-1)), (what, h)


def _worker(rank, world, port, partitions, err_dir):
    try:
        _worker_body(rank, world, port, partitions)
    except BaseException:
        with open(os.path.join(err_dir, "rank%d.err" % rank), "w") as f:
            f.write(traceback.format_exc())
        raise


def _worker_body(rank, world, port, partitions):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import euler_amd as EA
    from euler_amd.distributed import (gpu_sharded_sampler, run_interleaved, CTransport,
                                       c_sharded_sample_fanout)
    from oracle import oracle as O
    from conftest import make_random_graph
    dev = torch.device("cuda", 0)

    # ---------------- config 3 shape: weighted single-type power-law graph
    N = 300_000
    p = EA.synth_params(20240521, N, 10 * N, weighted=True)
    G_full = EA.Graph.synthetic(p)
    G_shard = EA.Graph.synthetic(p, partitions=partitions, shard_index=rank, shards=world)
    assert G_shard.num_nodes < G_full.num_nodes
    G_full.set_seed(5)
    G_shard.set_seed(5)
    S = gpu_sharded_sampler(G_shard, partitions=partitions)
    assert S.host_staged and S.world == world
    rng = np.random.default_rng(100 + rank)                 # every rank its own batch
    B = 20_000
    roots = torch.as_tensor(rng.integers(1, N + 1, B).astype(np.int64)).to(dev)
    et2 = [[0], [0]]
    want = G_full.sample_fanout(roots, et2, [25, 10], N + 1, call_id=6)
    got = S.sample_fanout(roots, et2, [25, 10], N + 1, call_id=6)
    _same(got[0], want[0], "fanout ids")
    _same(got[1], want[1], "fanout weights")
    _same(got[2], want[2], "fanout types")
    assert S.bytes_sent > 0 and S.bytes_received > 0        # rows really crossed ranks
    # ... and against the oracle (host generator == device generator)
    po = O.SynthParams()
    for f, _ in po._fields_:
        setattr(po, f, getattr(p, f))
    OG = O.OracleGraph(O.synth_csr(po))
    sel = t2n(roots)[:300]
    on, ow, ot = OG.sample_fanout(5, 6, sel, et2, [25, 10], N + 1)
    assert np.array_equal(t2n(got[0][1])[:300 * 25], on[0])
    assert np.array_equal(t2n(got[0][2])[:300 * 250], on[1])
    assert np.array_equal(t2n(got[1][1])[:300 * 250], ow[1])

    # the C-level path a C++ host calls (euler_gpu_sharded_sample_fanout): the hop is
    # orchestrated inside libeuler_gpu.so, the exchange goes through the transport's
    # two callbacks (host-staged here; euler_gpu_transport_rccl in production)
    tr = CTransport()
    got_c = c_sharded_sample_fanout(G_shard, tr, roots, et2, [25, 10], N + 1, call_id=6,
                                    partitions=partitions)
    _same(got_c[0], want[0], "C fanout ids")
    _same(got_c[1], want[1], "C fanout weights")
    _same(got_c[2], want[2], "C fanout types")
    assert tr.bytes_sent > 0
    mine_c = roots[:500] if rank != world - 1 else roots[:0]          # one rank with no batch
    got_c = c_sharded_sample_fanout(G_shard, tr, mine_c, et2, [3, 2], N + 1, call_id=15,
                                    partitions=partitions)
    _same(got_c[0], G_full.sample_fanout(mine_c, et2, [3, 2], N + 1, call_id=15)[0], "C empty batch")

    # the same fanout with several minibatches in flight (what bench.py runs)
    samplers = [gpu_sharded_sampler(G_shard, partitions=partitions) for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    batches = [torch.as_tensor(rng.integers(1, N + 1, 3000 + 500 * j).astype(np.int64)).to(dev)
               for j in range(4)]
    torch.cuda.synchronize()
    res = run_interleaved(
        lambda j: samplers[j % 2].sample_fanout_steps(batches[j], et2, [25, 10], N + 1,
                                                      call_id=40 + 2 * j),
        4, 2, enter=lambda k: torch.cuda.stream(streams[k]))
    for st in streams:
        st.synchronize()
    for j in range(4):
        w = G_full.sample_fanout(batches[j], et2, [25, 10], N + 1, call_id=40 + 2 * j)
        _same(res[j][0], w[0], "interleaved ids %d" % j)

    # implicit call ids advance identically on all ranks and re-randomise the sample
    S.set_call_id(1000)
    a = S.sample_fanout(roots[:2000], et2, [5, 3], N + 1)
    b = S.sample_fanout(roots[:2000], et2, [5, 3], N + 1)
    assert not torch.equal(a[0][1], b[0][1])
    _same(a[0], G_full.sample_fanout(roots[:2000], et2, [5, 3], N + 1, call_id=1000)[0], "cid a")
    _same(b[0], G_full.sample_fanout(roots[:2000], et2, [5, 3], N + 1, call_id=1002)[0], "cid b")

    # empty bucket: every root of every rank belongs to shard (world - 1); then a
    # rank with an empty batch (it still answers its peers)
    last = world - 1
    base = rng.integers(0, N // partitions - 1, 5000).astype(np.int64) * partitions
    own_last = base + [r for r in range(partitions) if r % world == last][0]
    own_last = torch.as_tensor(own_last[own_last >= 1]).to(dev)
    want = G_full.sample_neighbor(own_last, [0], 25, N + 1, call_id=9)
    got = S.sample_neighbor(own_last, [0], 25, N + 1, call_id=9)
    _same(got[:3], want, "one-owner batch")
    mine = roots[:777] if rank != 0 else roots[:0]
    want = G_full.sample_fanout(mine, et2, [4, 3], N + 1, call_id=11)
    got = S.sample_fanout(mine, et2, [4, 3], N + 1, call_id=11)
    _same(got[0], want[0], "empty batch on rank 0")

    # ---------------- config 4: random_walk length 40, one exchange pair per step
    L = 40
    starts = torch.as_tensor(rng.integers(1, N + 1, 4000).astype(np.int64)).to(dev)
    etw = [[0]] * L
    want = G_full.random_walk(starts, etw, 1.0, 1.0, N + 1, call_id=100)
    got = S.random_walk(starts, etw, default_node=N + 1, call_id=100)
    assert torch.equal(got, want)
    assert np.array_equal(t2n(got)[:64], OG.random_walk(5, 100, t2n(starts)[:64], etw, L, 1.0,
                                                       1.0, N + 1))
    # the C entry a C++ host calls (euler_gpu_sharded_random_walk: levels of merged walkers, the
    # GATHER deferred to the end of the walk) with 1 and 3 cohorts, a rank without walkers, the
    # hash front end (no id-indexed table), an empty walk; S.random_walk above already ran it
    from euler_amd.distributed import c_sharded_random_walk
    assert getattr(S, "c_walk_fn", None) is not None
    trw = CTransport()
    # round 6: the walk ENQUEUED (tuning key 63, the default: levels and buckets in slab layout,
    # fixed-size messages, ONE host exchange per call - the ranks' walker counts) and the polled
    # form (a wait for every step's bucket sizes): the same walks, the same level sizes
    from euler_amd import _lib
    seen = {}
    for enq, tail, split in ((1, 16, 10), (1, 0, 25), (1, 16, 0), (0, 0, 0)):
        # (key 66: from step `tail` on the enqueued walk sends its levels as they are - 0: every
        # step looks for entries that share a node, as the polled form does; key 67: the level at
        # which the path writer splits into two passes - the levels behind it are walked once per
        # entry of that level, not once per walker - 0: one pass)
        _lib.check(_lib.lib().euler_gpu_set_tuning(63, enq))
        _lib.check(_lib.lib().euler_gpu_set_tuning(66, tail))
        _lib.check(_lib.lib().euler_gpu_set_tuning(67, split))
        for cohorts, dense in ((1, S.dense_table), (3, None)):
            gotc, stats = c_sharded_random_walk(G_shard, trw, starts, etw, N + 1, 100, partitions, cohorts,
                                                dense, return_stats=True)
            assert torch.equal(gotc, want), ("C walk", enq, cohorts)
            assert stats["host_waits"] == (1 if enq else cohorts * L) and stats["ids_sent"] > 0
            # merged walkers: the levels hold fewer entries than walkers x steps
            assert stats["level_entries"] < starts.numel() * L
            # (the dense table merges every duplicate, the hash front end nearly all: compare like with like)
            if tail == 0:
                seen.setdefault(cohorts, []).append((stats["level_entries"], stats["ids_sent"]))
        mine_w = starts[:900] if rank != 0 else starts[:0]
        gotc = c_sharded_random_walk(G_shard, trw, mine_w, etw[:7], N + 1, 100, partitions, 2, S.dense_table)
        assert torch.equal(gotc, G_full.random_walk(mine_w, etw[:7], 1.0, 1.0, N + 1, call_id=100))
        # all walkers of the call on one rank, all starting on nodes of ONE owner: a bucket as large as the batch
        own0 = starts[(starts % partitions) % world == 0][:500] if rank == 1 % world else starts[:0]
        gotc = c_sharded_random_walk(G_shard, trw, own0, etw[:5], N + 1, 100, partitions, 1, None)
        assert torch.equal(gotc, G_full.random_walk(own0, etw[:5], 1.0, 1.0, N + 1, call_id=100))
    assert seen[1][0] == seen[1][1], seen                     # dense table: exact in both forms
    _lib.check(_lib.lib().euler_gpu_set_tuning(63, 1))
    _lib.check(_lib.lib().euler_gpu_set_tuning(66, 16))
    _lib.check(_lib.lib().euler_gpu_set_tuning(67, 10))
    mine_w = starts[:900] if rank != 0 else starts[:0]
    gotc = c_sharded_random_walk(G_shard, trw, mine_w, etw[:7], N + 1, 100, partitions, 2, S.dense_table)
    assert torch.equal(gotc, G_full.random_walk(mine_w, etw[:7], 1.0, 1.0, N + 1, call_id=100))
    gotc = c_sharded_random_walk(G_shard, trw, starts[:10], [], N + 1, 100, partitions, 2, None)
    assert torch.equal(gotc.reshape(-1), starts[:10])
    # ... and its node2vec run (p = 0.25, q = 4): every step fetches the rows of the walkers'
    # nodes from their owners and draws on the requester (random_walk_op.cc:83-168)
    etn = [[0]] * 8
    want = G_full.random_walk(starts[:1500], etn, 0.25, 4.0, N + 1, call_id=200)
    got = S.random_walk(starts[:1500], etn, 0.25, 4.0, default_node=N + 1, call_id=200)
    assert torch.equal(got, want)
    assert np.array_equal(t2n(got)[:32], OG.random_walk(5, 200, t2n(starts)[:32], etn, 8, 0.25,
                                                       4.0, N + 1))
    # (S.random_walk just ran the C entry, euler_gpu_sharded_node2vec_walk; the same walk as the
    # reference's client loop in Python, through the entry itself with the hash front end, from
    # a rank without walkers and with no steps)
    from euler_amd.distributed import c_sharded_node2vec_walk
    assert getattr(S, "c_n2v_fn", None) is not None
    c_fn, S.c_n2v_fn = S.c_n2v_fn, None
    got_py = S.random_walk(starts[:1500], etn, 0.25, 4.0, default_node=N + 1, call_id=200)
    S.c_n2v_fn = c_fn
    assert torch.equal(got_py, want)
    gotn, nstats = c_sharded_node2vec_walk(G_shard, trw, starts[:1500], etn, 0.25, 4.0, N + 1, 200, partitions, None,
                                           return_stats=True)
    assert torch.equal(gotn, want)
    assert nstats["rows_asked"] <= 1500 * 8 and nstats["row_entries"] > 0 and nstats["ids_sent"] > 0
    # (the step's long rows by a workgroup each - N2vBigStepListKernel, tuning key 69 - with the bar
    # low enough for this graph's hubs)
    _lib.check(_lib.lib().euler_gpu_set_tuning(69, 48))
    gotn = c_sharded_node2vec_walk(G_shard, trw, starts[:1500], etn, 0.25, 4.0, N + 1, 200, partitions, None)
    _lib.check(_lib.lib().euler_gpu_set_tuning(69, 65536))
    assert torch.equal(gotn, want)
    mine_n = starts[:700] if rank != 0 else starts[:0]
    gotn = c_sharded_node2vec_walk(G_shard, trw, mine_n, etn[:5], 2.0, 0.5, N + 1, 210, partitions, S.dense_table)
    assert torch.equal(gotn, G_full.random_walk(mine_n, etn[:5], 2.0, 0.5, N + 1, call_id=210))
    gotn = c_sharded_node2vec_walk(G_shard, trw, starts[:10], [], 0.25, 4.0, N + 1, 220, partitions, None)
    assert torch.equal(gotn.reshape(-1), starts[:10])

    # the record DESIGN.md quotes: 20 000 walkers x 10 node2vec steps through the sharded sampler
    # (world ranks on ONE GPU, host-staged exchange), wave kernels on the fetched rows
    import time as _time
    etn10 = [[0]] * 10
    S.random_walk(starts[:20000], etn10, 0.25, 4.0, default_node=N + 1, call_id=300)
    torch.cuda.synchronize(); dist.barrier()
    _t0 = _time.perf_counter()
    got10 = S.random_walk(starts[:20000], etn10, 0.25, 4.0, default_node=N + 1, call_id=300)
    torch.cuda.synchronize(); dist.barrier()
    _ms = (_time.perf_counter() - _t0) * 1e3
    _same(got10, G_full.random_walk(starts[:20000], etn10, 0.25, 4.0, N + 1, call_id=300), "node2vec 20K x 10")
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "r6_sharded_n2v_world%d_rank%d.txt" % (world, rank)), "w") as fo:
            fo.write("sharded node2vec (p = 0.25, q = 4), %d ranks on one GPU, 20000 walkers x 10 steps per rank: "
                     "%.2f ms\n" % (world, _ms))
    # dedup="ops" (ID_UNIQUE / ID_SPLIT / merge_rows / gather as separate kernels)
    S_ops = gpu_sharded_sampler(G_shard, partitions=partitions, dedup="ops")
    got = S_ops.sample_fanout(roots[:5000], et2, [6, 4], N + 1, call_id=13)
    _same(got[0], G_full.sample_fanout(roots[:5000], et2, [6, 4], N + 1, call_id=13)[0], "ops")
    del G_full, G_shard, S, S_ops, samplers

    # ---------------- config 5: 8 edge types, typed sampling k = 1 / 3 of 8 / all
    N5, T = 60_000, 8
    p5 = EA.synth_params(77, N5, 40 * N5, n_types=T, weighted=True)
    G5 = EA.Graph.synthetic(p5)
    G5s = EA.Graph.synthetic(p5, partitions=partitions, shard_index=rank, shards=world)
    G5.set_seed(8)
    G5s.set_seed(8)
    S5 = gpu_sharded_sampler(G5s, partitions=partitions)
    po5 = O.synth_params(77, N5, 40 * N5, n_types=T, weighted=True)
    OG5 = O.OracleGraph(O.synth_csr(po5))
    r5 = rng.integers(1, N5 + 1, 8000).astype(np.int64)
    r5t = torch.as_tensor(r5).to(dev)
    for call, et in enumerate(([3], [1, 4, 6], list(range(T)), [])):
        want = G5.sample_neighbor(r5t, et, 10, N5 + 1, call_id=call)
        got = S5.sample_neighbor(r5t, et, 10, N5 + 1, call_id=call)
        _same(got[:3], want, ("typed", et))
        on, ow, ot = OG5.sample_neighbor(8, call, r5[:500], et, 10, N5 + 1)
        assert np.array_equal(t2n(got[0])[:500], on) and np.array_equal(t2n(got[2])[:500], ot)
    # the three typed draws of a heterogeneous minibatch behind ONE front end / id exchange
    sets = [[3], [1, 4, 6], list(range(T))]
    outs = S5.sample_neighbor_sets(r5t, sets, 10, N5 + 1, call_id=40)
    for c, et in enumerate(sets):
        _same(outs[c][:3], G5.sample_neighbor(r5t, et, 10, N5 + 1, call_id=40 + c), ("sets", et))
    # typed fanout + aggregation on the sharded sample (scatter_mean acts on the
    # minibatch-local block: replicas only, no collective)
    gn, gw, gt = S5.sample_fanout(r5t, [[1, 4, 6], [0, 2, 5]], [5, 5], N5 + 1, call_id=20)
    wn, ww, wt = G5.sample_fanout(r5t, [[1, 4, 6], [0, 2, 5]], [5, 5], N5 + 1, call_id=20)
    _same(gn, wn, "typed fanout ids"); _same(gt, wt, "typed fanout types")
    feat = torch.randn(N5 + 2, 32, device=dev, generator=torch.Generator(dev).manual_seed(3))
    x = EA.ops.gather(feat, gn[1].to(torch.int32))
    dst = torch.arange(len(r5), device=dev, dtype=torch.int32).repeat_interleave(5)
    agg = EA.ops.scatter_mean(x, dst, len(r5))
    assert np.array_equal(t2n(agg), O.scatter_mean(t2n(x), t2n(dst), len(r5)))
    del G5, G5s, S5

    # ---------------- arbitrary u64 ids (hash id map), 3 types, zero weights, missing rows
    grng = np.random.default_rng(4)                         # same graph on every rank
    ids, seg, nbr, w, nt, nw = make_random_graph(grng, 20000, 3, max_deg=30, id_space=10 ** 12)
    csr = O.csr_from_raw(ids, seg, nbr, w, 3, nt, nw)
    mk = lambda **kw: EA.Graph.from_csr(csr.row_id, csr.row_ptr, csr.type_end, csr.nbr,
                                        csr.prefix_w, csr.type_prefix, csr.n_types,
                                        csr.node_type, csr.node_weight, **kw)
    Gh = mk()
    Ghs = mk(partitions=partitions, shard_index=rank, shards=world)
    Gh.set_seed(31)
    Ghs.set_seed(31)
    Sh = gpu_sharded_sampler(Ghs, partitions=partitions)
    assert Sh.dense_table is None
    q = np.concatenate([rng.choice(ids, 6000), rng.choice(ids[:40], 3000), [0, 999]]).astype(np.int64)
    qt = torch.as_tensor(q).to(dev)
    OGh = O.OracleGraph(csr)
    for ets, cnts in (([[0, 1], [2, 0]], [6, 4]), ([[0], [1]], [7, 5]), ([[], []], [3, 3])):
        on, ow, ot = OGh.sample_fanout(31, 8, q, ets, cnts, -1)
        got = Sh.sample_fanout(qt, ets, cnts, -1, call_id=8)
        for h in range(2):
            assert np.array_equal(t2n(got[0][h + 1]), on[h]), (ets, h)
            assert np.array_equal(t2n(got[1][h]), ow[h]) and np.array_equal(t2n(got[2][h]), ot[h])
    # typed / multi-type hops through the C entry (type draws: sample + pack on the owner)
    trh = CTransport()
    for ets, cnts in (([[0, 1], [2, 0]], [6, 4]), ([[1], [1]], [5, 5])):
        on, ow, ot = OGh.sample_fanout(31, 8, q, ets, cnts, -1)
        got = c_sharded_sample_fanout(Ghs, trh, qt, ets, cnts, -1, call_id=8, partitions=partitions)
        for h in range(2):
            assert np.array_equal(t2n(got[0][h + 1]), on[h]), ("C", ets, h)
            assert np.array_equal(t2n(got[1][h]), ow[h]) and np.array_equal(t2n(got[2][h]), ot[h])
    gi_, gd_, gw_, gt_ = Sh.get_full_neighbor(qt, [0, 2])
    wi_, wd_, ww_, wt_ = OGh.get_full_neighbor(q.astype(np.uint64), [0, 2])
    assert np.array_equal(t2n(gi_), wi_) and np.array_equal(t2n(gd_).astype(np.uint64), wd_)
    assert np.array_equal(t2n(gw_), ww_) and np.array_equal(t2n(gt_), wt_)
    # SampleNode over the shards: SAMPLE_NODE_SPLIT + local draws + APPEND_MERGE; every
    # rank gets the same ids, all of the asked type
    sn = t2n(Sh.sample_node(500, 1, call_id=60))
    assert len(sn) == 500 and np.all(np.isin(sn.astype(np.uint64), ids[nt == 1]))
    gathered = [None] * world
    dist.all_gather_object(gathered, sn.tolist())
    assert all(g == gathered[0] for g in gathered)
    walk = Sh.random_walk(qt[:2000], [[0, 1, 2]] * 6, default_node=-1, call_id=50)
    assert np.array_equal(t2n(walk), OGh.random_walk(31, 50, q[:2000], [[0, 1, 2]] * 6, 6, 1.0,
                                                     1.0, -1))
    walk = Sh.random_walk(qt[:500], [[0, 1, 2]] * 4, 2.0, 0.5, default_node=-1, call_id=55)
    assert np.array_equal(t2n(walk), OGh.random_walk(31, 55, q[:500], [[0, 1, 2]] * 4, 4, 2.0,
                                                     0.5, -1))
    dist.barrier()
    dist.destroy_process_group()


def _worker8(rank, world, port, partitions, err_dir):
    try:
        _worker8_body(rank, world, port, partitions, err_dir)
    except BaseException:
        with open(os.path.join(err_dir, "rank%d.err" % rank), "w") as f:
            f.write(traceback.format_exc())
        raise


def _worker8_body(rank, world, port, partitions, out_dir):
    """8 ranks, partitions = 1024 (owner = (id % 1024) % 8: the reference's file partitions,
    core/graph/graph.cc:90-98), the C entries only - what a C++ host on an 8-GPU node calls -
    with a batch one rank owns alone, ranks without a batch, and the sequence of transport
    callbacks of every rank written out for the parent to compare."""
    import json
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import euler_amd as EA
    from euler_amd.distributed import (CTransport, c_sharded_sample_fanout, c_sharded_random_walk,
                                       c_sharded_node2vec_walk, owner_of)
    dev = torch.device("cuda", 0)
    N = 60_000
    p = EA.synth_params(20240521, N, 10 * N, weighted=True)
    G_full = EA.Graph.synthetic(p)
    G_shard = EA.Graph.synthetic(p, partitions=partitions, shard_index=rank, shards=world)
    G_full.set_seed(9); G_shard.set_seed(9)
    tr = CTransport()
    tr.log = []
    rng = np.random.default_rng(500 + rank)
    all_ids = torch.arange(1, N + 1, dtype=torch.int64)
    mine5 = all_ids[owner_of(all_ids, partitions, world) == 5]
    batches = [
        torch.as_tensor(rng.integers(1, N + 1, 3000 + 111 * rank).astype(np.int64)),
        mine5[torch.as_tensor(rng.integers(0, mine5.numel(), 2048))],            # rank 5 owns every root
        torch.as_tensor(rng.integers(1, N + 1, 1500).astype(np.int64)) if rank % 3 else all_ids[:0],
    ]
    for b, roots in enumerate(batches):
        roots = roots.to(dev)
        want = G_full.sample_fanout(roots, [[0], [0]], [25, 10], N + 1, call_id=20 * b)
        got = c_sharded_sample_fanout(G_shard, tr, roots, [[0], [0]], [25, 10], N + 1, call_id=20 * b,
                                      partitions=partitions)
        _same(got[0], want[0], "C fanout ids, batch %d" % b)
        _same(got[1], want[1], "C fanout weights, batch %d" % b)
        starts = roots[:1000]
        etw = [[0]] * 12
        gotw = c_sharded_random_walk(G_shard, tr, starts, etw, N + 1, 300 + 20 * b, partitions, 1, None)
        assert torch.equal(gotw, G_full.random_walk(starts, etw, 1.0, 1.0, N + 1, call_id=300 + 20 * b)), b
        gotn = c_sharded_node2vec_walk(G_shard, tr, starts[:400], etw[:4], 0.25, 4.0, N + 1, 600 + 20 * b,
                                       partitions, None)
        assert torch.equal(gotn, G_full.random_walk(starts[:400], etw[:4], 0.25, 4.0, N + 1, call_id=600 + 20 * b)), b
    with open(os.path.join(out_dir, "callbacks_rank%d.json" % rank), "w") as f:
        json.dump(tr.log, f)
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_partitions_1024_c_entries_on_one_gpu(torch_cuda, tmp_path):
    """The 8-GPU configuration's C entries (fanout, DeepWalk, node2vec) with 8 ranks sharing this
    GPU (host-staged gloo transport): results == the unsharded graph, and every rank makes the
    same sequence of transport callbacks whatever its batch holds."""
    import json
    import torch.multiprocessing as mp
    world, partitions = 8, 1024
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker8, args=(r, world, port, partitions, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
    errs = []
    for r, p in enumerate(procs):
        if p.is_alive():
            p.kill()
            errs.append("rank %d timed out" % r)
        f = tmp_path / ("rank%d.err" % r)
        if f.exists():
            errs.append("rank %d:\n%s" % (r, f.read_text()))
        elif p.exitcode not in (0, None):
            errs.append("rank %d exit code %s" % (r, p.exitcode))
    assert not errs, "\n".join(errs)
    logs = [json.load(open(str(tmp_path / ("callbacks_rank%d.json" % r)))) for r in range(world)]
    assert all(l == logs[0] for l in logs[1:]), "ranks made different sequences of transport callbacks"
    # per batch: fanout 2 hops x (counts, ids, rows) + the enqueued walk's ONE counts exchange
    # (the ranks' walker counts) + 12 x (ids, answers) + node2vec 4 x (counts, ids, lengths,
    # counts, ids, weights)
    assert len(logs[0]) == 3 * (2 * 3 + 1 + 12 * 2 + 4 * 6)


@pytest.mark.parametrize("world,partitions", [(2, 2), (2, 6), (3, 3)])
def test_sharded_hip_pieces_world_gt1_on_one_gpu(torch_cuda, tmp_path, world, partitions):
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, partitions, str(tmp_path)))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
    errs = []
    for r, p in enumerate(procs):
        if p.is_alive():
            p.kill()
            errs.append("rank %d timed out" % r)
        f = tmp_path / ("rank%d.err" % r)
        if f.exists():
            errs.append("rank %d:\n%s" % (r, f.read_text()))
        elif p.exitcode not in (0, None):
            errs.append("rank %d exit code %s" % (r, p.exitcode))
    assert not errs, "\n".join(errs)


def test_cpp_multi_gpu_host_over_rccl(torch_cuda):
    """examples/cpp/sharded_fanout: a C++ host (no Python, no torch) with one thread per
    GPU, euler_gpu_transport_rccl (ncclSend / ncclRecv groups, resolved at run time) and
    euler_gpu_sharded_sample_fanout; it compares every rank's result with the unsharded
    graph itself - the fanout and a random_walk of length 40 (euler_gpu_sharded_random_walk).
    One GPU here, so one rank: a rank's own ids never leave it (no exchange), the rest is the
    code path an 8-GPU node runs with N = 8; a second run makes the lone rank's exchanges anyway
    (EULER_GPU_SELF_EXCHANGE=1), so that the RCCL transport itself executes."""
    import subprocess
    exe = os.path.join(ROOT, "examples", "cpp", "sharded_fanout")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples", "cpp")])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([exe, "1", "300000", "2048"], capture_output=True, text=True, timeout=600,
                         env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "sharded_fanout OK: 1 rank(s)" in out.stdout, out.stdout + out.stderr
    # ... and with the lone rank's exchanges made (it sends itself what N ranks send one another):
    # the ncclSend / ncclRecv groups of the transport execute on this box
    env["EULER_GPU_SELF_EXCHANGE"] = "1"
    out = subprocess.run([exe, "1", "300000", "2048"], capture_output=True, text=True, timeout=600,
                         env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "sharded_fanout OK: 1 rank(s)" in out.stdout, out.stdout + out.stderr
