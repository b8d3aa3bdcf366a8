"""BASELINE.json `configs` as parity cases, each against the CPU oracle
(and the reference sampler build where it is present).

  config 3  the metric at FULL size (100M nodes / 1B edges, weighted, fanout [25,10],
            131 072 roots): the one-kernel step == the hop-by-hop kernels on all 36 M
            samples, == the (distinct rows, index) form, == the oracle on 96 roots fed
            with the rows exported from HBM; DeepWalk 1M x 40 over merged groups == the
            per-walker kernel (bench.py repeats the oracle checks on every run)

  config 1  Cora-shaped graph, GraphSAGE 2-hop fanout [10,5], batch 32 - the
            whole minibatch construction on the CPU oracle (plumbing, no GPU)
  config 2  ogbn-products-shaped CSR, uniform weights, fanout [25,10], 1 GPU
  config 4  DeepWalk: p = q = 1 random walk of length 40 (sharded N > 1 host
            logic: tests/test_distributed_cpu.py with gloo)
  config 5  heterogeneous typed graph: per-type sampling (k = 1, 3-of-8, all)
            + 128-d features + scatter_mean aggregation; and at bench.py's size (20M
            nodes, 131 072 roots): pivots == reference loop == duplicate path == oracle
"""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def t2n(t):
    return t.detach().cpu().numpy()


def cora_shaped(O):
    """tf_euler/python/dataset/cora.py:36-52 counts: 2 708 nodes, 5 429 cites
    edges (directed), 1 433-dim row-normalised bag-of-words features, node
    weight 1, edge weight 1, edge types train / train_removed.  The real files
    need a download, so the shape is synthetic with a fixed seed."""
    rng = np.random.default_rng(20240521)
    n, e, d = 2708, 5429, 1433
    ids = np.arange(1, n + 1).astype(np.uint64)
    src = rng.integers(0, n, e)
    dst = rng.integers(0, n, e)
    et = (rng.random(e) < 0.1).astype(np.int64)      # ~10 % train_removed
    order = np.lexsort((et, src))
    src, dst, et = src[order], dst[order], et[order]
    T = 2
    seg = np.zeros(n * T + 1, np.int64)
    np.add.at(seg, src * T + et + 1, 1)
    seg = np.cumsum(seg)
    nbr = ids[dst]
    w = np.ones(e, np.float32)
    feat = (rng.random((n, d)) < 0.012).astype(np.float32)
    feat /= np.maximum(feat.sum(1, keepdims=True), 1)
    F = O.DenseFeatures(1, np.arange(n + 1) * d, np.full(n, d), feat.reshape(-1))
    return ids, seg, nbr, w, T, F, feat


def test_config1_cora_graphsage_minibatch_on_cpu_oracle(O):
    """configs[0]: roots -> fanout [10,5] -> feature fetch -> mean aggregation
    of the sampled block, all on the oracle; the sampled ids are additionally
    compared with the reference sampler build when it is present."""
    ids, seg, nbr, w, T, F, feat = cora_shaped(O)
    csr = O.csr_from_raw(ids, seg, nbr, w, T)
    OG = O.OracleGraph(csr)
    rng = np.random.default_rng(1)
    roots = rng.choice(ids, 32).astype(np.int64)
    fan = [10, 5]
    et = [[0], [0]]
    nbrs, ws, ts = OG.sample_fanout(7, 0, roots, et, fan, -1)
    assert nbrs[0].shape == (32 * 10,) and nbrs[1].shape == (32 * 10 * 5,)
    if O.have_ref():
        R = O.RefGraph.build_raw(ids, seg, nbr, w, T)
        cur = roots.astype(np.uint64)
        for h in range(2):       # hop by hop on the reference's core tensors
            _, oid, ow, _ = OG.sample_neighbor_core(7, h, cur, et[h], fan[h])
            _, rid, rw, _ = R.sample_neighbor_core(7, h, cur, et[h], fan[h])
            assert np.array_equal(oid, rid) and np.array_equal(ow, rw)
            cur = oid
    # layer-2 block: neighbours' features averaged into their hop-1 parents
    x2 = OG.get_dense_feature(F, nbrs[1], [0], [1433])[0]
    valid = nbrs[1] > 0
    assert np.array_equal(x2[valid], feat[nbrs[1][valid] - 1])
    assert not x2[~valid].any()
    dst = np.repeat(np.arange(32 * 10, dtype=np.int32), 5)
    agg = O.scatter_mean(x2, dst, 32 * 10)
    want = x2.reshape(320, 5, -1).astype(np.float64).mean(1)
    assert np.allclose(agg, want, rtol=0, atol=1e-5)


@pytest.mark.gpu
def test_config2_products_shaped_uniform_fanout(EA, O, torch_cuda):
    """configs[1]: ogbn-products-shaped CSR (2 449 029 nodes, ~123.7 M directed
    edges, all weights 1.0), uniform SampleNeighbor fanout [25,10].  The GPU
    graph is built at full size; 512 roots are checked bit for bit against the
    oracle fed with the rows exported from HBM, the rest through determinism
    and membership."""
    torch = torch_cuda
    N, E = 2_449_029, 123_718_280
    p = EA.synth_params(42, N, E, weighted=False)
    G = EA.Graph.synthetic(p)
    assert G.num_nodes == N and abs(G.num_edges - E) < 0.02 * E
    rng = np.random.default_rng(2)
    roots = rng.integers(1, N + 1, 4096).astype(np.int64)
    G.set_seed(42)
    a = G.sample_fanout(torch.as_tensor(roots).cuda(), [[0], [0]], [25, 10], N + 1,
                        call_id=4)
    b = G.sample_fanout(torch.as_tensor(roots).cuda(), [[0], [0]], [25, 10], N + 1,
                        call_id=4)
    for x, y in zip(a[0], b[0]):
        assert torch.equal(x, y)
    assert bool((a[1][1] == 1.0).all())           # uniform weights come back as 1.0
    sel = np.arange(512)
    hop1 = t2n(a[0][1]).reshape(4096, 25)[sel]
    need = np.unique(np.concatenate([roots[sel], hop1.reshape(-1)])).astype(np.uint64)
    need = need[(need >= 1) & (need <= N)]
    rp, te, nb, pw, tp = G.export_rows(need)
    OG = O.OracleGraph(O.CSR(need, rp, te, nb, pw, tp, 1))
    on, ow, ot = OG.sample_fanout(42, 4, roots[sel], [[0], [0]], [25, 10], N + 1)
    assert np.array_equal(on[0], hop1.reshape(-1))
    assert np.array_equal(on[1], t2n(a[0][2]).reshape(4096, 250)[sel].reshape(-1))
    assert np.array_equal(ow[1], t2n(a[1][1]).reshape(4096, 250)[sel].reshape(-1))


@pytest.mark.gpu
def test_plain_graph_between_2_31_and_2_32_edges(EA, O, torch_cuda):
    """a plain weighted graph of 2.5 billion edges (250M nodes; 2.5x the headline
    graph, past the signed 32-bit edge numbers the one-kernel builds stopped at) is served by the
    one-kernel step of fanout_plain.h: every sample of a step == hop-by-hop sampling, and a step
    whose roots are taken from the rows past edge 2^31 == the CPU oracle on exported rows."""
    torch = torch_cuda
    from euler_amd import _lib
    L = _lib.lib()
    N, B = 250_000_000, 32768
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < 200 * 2 ** 30:
        pytest.skip("needs ~150 GB of free HBM")
    G = EA.Graph.synthetic(EA.synth_params(20240522, N, 10 * N, weighted=True))
    try:
        assert G.num_edges > 2 ** 31 + 2 ** 28
        G.set_seed(11)
        gen = torch.Generator(device="cuda"); gen.manual_seed(5)
        roots = torch.randint(1, N + 1, (B,), generator=gen, device="cuda", dtype=torch.int64)
        roots[: B // 2] = torch.randint(int(0.9 * N), N + 1, (B // 2,), generator=gen, device="cuda", dtype=torch.int64)
        try:
            L.euler_gpu_set_tuning(27, 1)
            a = G.sample_fanout(roots, [[0], [0]], [25, 10], N + 1, call_id=3)
            assert (L.euler_gpu_last_fanout_kernel() or b"").decode() == "SampleFanoutPlainKernel"
            L.euler_gpu_set_tuning(27, 0)
            h = G.sample_fanout(roots, [[0], [0]], [25, 10], N + 1, call_id=3)
            assert (L.euler_gpu_last_fanout_kernel() or b"").decode() == "hop by hop"
        finally:
            L.euler_gpu_set_tuning(27, 1)
        for hop in range(2):
            assert torch.equal(a[0][hop + 1], h[0][hop + 1]) and torch.equal(a[1][hop], h[1][hop])
            assert torch.equal(a[2][hop], h[2][hop])
        del h
        # the exported rows of the high roots really lie past edge 2^31
        rp = G.export_rows(np.array([N - 5, N - 4], np.uint64))[0]
        assert len(rp) == 3
        from oracle.step_check import check_fanout_step
        edges, distinct = check_fanout_step(G, O.OracleGraph, O.CSR, 11, 3, roots, a[0], a[1], a[2], [25, 10], N + 1, N)
        assert edges == B * 275 and distinct > 50_000
        # a DeepWalk of 100 000 walkers x 10 on the same graph: the walk kernels' own paths == one lane per walker
        starts = torch.randint(1, N + 1, (100_000,), generator=gen, device="cuda", dtype=torch.int64)
        w = G.random_walk(starts, [[0]] * 10, 1.0, 1.0, N + 1, call_id=50)
        walks = t2n(w[:16])          # (16 walkers against the oracle on exported rows)
        rows = np.unique(walks[walks <= N]).astype(np.uint64)
        rp, te, nb, pw, tp = G.export_rows(rows)
        OG = O.OracleGraph(O.CSR(rows, rp, te, nb, pw, tp, 1))
        assert np.array_equal(walks, OG.random_walk(11, 50, t2n(starts[:16]), [[0]] * 10, 10, 1.0, 1.0, N + 1))
    finally:
        del G
        torch.cuda.empty_cache()


@pytest.mark.gpu
def test_config3_metric_step_at_full_size(EA, O, torch_cuda):
    """configs[2], the headline workload, at its full size: two independent device
    paths agree bit for bit on every one of the step's 36 044 800 samples (the one-kernel
    fanout of fanout_local.h - through the weight-bucket index and through the pivot levels -
    against hop-by-hop sampling + global duplicate path + expansion), the (distinct rows, index) form reproduces the dense one, the result is
    a pure function of (seed, call id, roots), and the WHOLE step - the 1 000 largest hubs and
    1 000 rows with overflowing index buckets among its roots - equals the CPU oracle on rows
    exported from HBM (oracle/step_check.py).  Same for DeepWalk of 1M walkers x 40
    steps: groups of merged walkers == one lane per walker, 16 walkers == the oracle."""
    torch = torch_cuda
    from euler_amd import _lib
    L = _lib.lib()
    N, B = 100_000_000, 131072
    G = EA.Graph.synthetic(EA.synth_params(20240521, N, 10 * N, weighted=True))
    assert G.num_nodes == N and abs(G.num_edges - 10 * N) < 0.01 * 10 * N
    G.set_seed(20240521)
    gen = torch.Generator(device="cuda"); gen.manual_seed(77)
    roots = torch.randint(1, N + 1, (B,), generator=gen, device="cuda", dtype=torch.int64)
    # on purpose among the roots: the graph's largest hubs - the 1 000 nodes a
    # probe step's hop 1 draws most often, rows of 10^4 .. 5 x 10^5 edges past the 2^24 resolution
    # of an f32 running sum - and up to 1 000 rows in which a bucket of the weight-bucket index
    # overflows its block (their draws take the fallback search)
    probe = G.sample_fanout(roots, [[0], [0]], [25, 10], N + 1, call_id=2)[0][1]
    hub_ids, hub_cnt = torch.unique(probe, return_counts=True)
    hubs = hub_ids[torch.argsort(hub_cnt, descending=True)[:1000]]
    hubs = hubs[hubs <= N]
    ovf = torch.as_tensor(G.index_overflow_rows(1000).astype(np.int64)).cuda()
    assert ovf.numel() > 0, "the metric graph has buckets that overflow (1e-4 of them)"
    roots[:hubs.numel()] = hubs
    roots[hubs.numel():hubs.numel() + ovf.numel()] = ovf
    del probe, hub_ids, hub_cnt
    try:
        L.euler_gpu_set_tuning(27, 1)
        a = G.sample_fanout(roots, [[0], [0]], [25, 10], N + 1, call_id=6)
        a2 = G.sample_fanout(roots, [[0], [0]], [25, 10], N + 1, call_id=6)
        L.euler_gpu_set_tuning(53, 0)             # round 5's build of the one-kernel step (fanout_local.h)
        r5 = G.sample_fanout(roots, [[0], [0]], [25, 10], N + 1, call_id=6)
        L.euler_gpu_set_tuning(53, 1)
        for hop in range(2):
            assert torch.equal(a[0][hop + 1], r5[0][hop + 1]) and torch.equal(a[1][hop], r5[1][hop])
            assert torch.equal(a[2][hop], r5[2][hop])
        del r5
        L.euler_gpu_set_tuning(27, 0)
        h = G.sample_fanout(roots, [[0], [0]], [25, 10], N + 1, call_id=6)
        L.euler_gpu_set_tuning(27, 1); L.euler_gpu_set_tuning(45, 0)     # pivot levels instead of
        lv = G.sample_fanout(roots, [[0], [0]], [25, 10], N + 1, call_id=6)   # the weight-bucket index
    finally:
        L.euler_gpu_set_tuning(27, 1); L.euler_gpu_set_tuning(45, 1); L.euler_gpu_set_tuning(53, 1)
    assert a[0][2].numel() == B * 250
    for hop in range(2):
        assert torch.equal(a[0][hop + 1], h[0][hop + 1]) and torch.equal(a[0][hop + 1], a2[0][hop + 1])
        assert torch.equal(a[1][hop], h[1][hop]) and torch.equal(a[2][hop], h[2][hop])
        assert torch.equal(a[0][hop + 1], lv[0][hop + 1]) and torch.equal(a[1][hop], lv[1][hop])
    del h, a2, lv
    id1, w1, t1, ridx, rid, rw, rt = G.sample_fanout_unique(roots, [[0], [0]], [25, 10], N + 1, call_id=6)
    assert torch.equal(id1.reshape(-1), a[0][1])
    assert torch.equal(rid[ridx].reshape(-1), a[0][2]) and torch.equal(rw[ridx].reshape(-1), a[1][1])
    assert torch.equal(rt[ridx].reshape(-1), a[2][1])
    del rid, rw, rt
    # the WHOLE step against the CPU oracle (oracle/step_check.py): all 131 072 x 25 hop-1
    # samples on exported rows, every hop-2 position against the first position of its node on the
    # device, and one row per distinct hop-1 child - ~285 K rows of 10 - against the oracle
    from oracle.step_check import check_fanout_step
    edges, distinct = check_fanout_step(G, O.OracleGraph, O.CSR, 20240521, 6, roots, a[0], a[1], a[2], [25, 10],
                                        N + 1, N)
    assert edges == B * 275 and distinct > 200_000
    del a
    # DeepWalk at configs[3]'s size on the same graph
    W = 1_000_000
    starts = torch.randint(1, N + 1, (W,), generator=gen, device="cuda", dtype=torch.int64)
    et = [[0]] * 40
    try:
        L.euler_gpu_set_tuning(38, 262144)
        m = G.random_walk(starts, et, 1.0, 1.0, N + 1, call_id=50)
        L.euler_gpu_set_tuning(38, 0)
        w = G.random_walk(starts, et, 1.0, 1.0, N + 1, call_id=50)
    finally:
        L.euler_gpu_set_tuning(38, 262144)
    assert torch.equal(m, w)
    walks = t2n(m[:16])
    rows = np.unique(walks[walks <= N]).astype(np.uint64)
    rp, te, nb, pw, tp = G.export_rows(rows)
    OG = O.OracleGraph(O.CSR(rows, rp, te, nb, pw, tp, 1))
    assert np.array_equal(walks, OG.random_walk(20240521, 50, t2n(starts[:16]), et, 40, 1.0, 1.0, N + 1))


@pytest.mark.gpu
@pytest.mark.parametrize("weighted", [True, False], ids=["weighted", "uniform"])
def test_general_form_step_at_full_size(EA, O, torch_cuda, weighted):
    """The general builds of the one-kernel step (fanout_local.h, WB = 2 / 4 weighted, 6 / 5
    uniform) at the metric's size - 100M nodes / 1B edges with hashed u64 ids (a hash map of 268M
    slots) and two edge-type groups, hub rows of half a million edges in two groups: on all
    36 044 800 samples of a step the one-kernel form (record in the 64-byte hash slot, and
    through the 16-byte slot + row record) == the hop-by-hop kernels, for one listed type per
    hop and for hops that list both (a type draw per sample); 96 roots == the oracle."""
    torch = torch_cuda
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    from euler_amd import _lib
    L = _lib.lib()
    N, B = 100_000_000, 131072
    p = EA.synth_params(20240521, N, 10 * N, n_types=2, weighted=weighted, hashed_ids=True)
    G = EA.Graph.synthetic(p)
    assert G.num_nodes == N
    G.set_seed(20240521)
    gen = torch.Generator(device="cuda"); gen.manual_seed(78)
    roots = bench._mix64_t(torch.randint(1, N + 1, (B,), generator=gen, device="cuda", dtype=torch.int64))
    for et in ([[0], [0]], [[0, 1], [0, 1]]):
        try:
            L.euler_gpu_set_tuning(27, 1)
            a = G.sample_fanout(roots, et, [25, 10], -1, call_id=6)
            L.euler_gpu_set_tuning(49, 0)
            s16 = G.sample_fanout(roots, et, [25, 10], -1, call_id=6)
            L.euler_gpu_set_tuning(49, 1)
            L.euler_gpu_set_tuning(27, 0)
            h = G.sample_fanout(roots, et, [25, 10], -1, call_id=6)
        finally:
            L.euler_gpu_set_tuning(27, 1); L.euler_gpu_set_tuning(49, 1)
        assert a[0][2].numel() == B * 250
        for hop in range(2):
            for other in (s16, h):
                assert torch.equal(a[0][hop + 1], other[0][hop + 1]), (et, hop)
                assert torch.equal(a[1][hop], other[1][hop]) and torch.equal(a[2][hop], other[2][hop])
        del s16, h
        sel = np.random.default_rng(5).choice(B, 96, replace=False)
        r_sel = t2n(roots)[sel]
        hop1 = t2n(a[0][1]).reshape(B, 25)[sel]
        need = np.concatenate([r_sel, hop1.reshape(-1)])
        OG = bench._oracle_rows(G, p, need[need != -1], 2)
        on, ow, ot = OG.sample_fanout(20240521, 6, r_sel, et, [25, 10], -1)
        assert np.array_equal(on[0], hop1.reshape(-1))
        assert np.array_equal(on[1], t2n(a[0][2]).reshape(B, 250)[sel].reshape(-1))
        assert np.array_equal(ow[1], t2n(a[1][1]).reshape(B, 250)[sel].reshape(-1))
        assert np.array_equal(ot[1], t2n(a[2][1]).reshape(B, 250)[sel].reshape(-1))
        del a


@pytest.mark.gpu
def test_config4_deepwalk_walk_length_40(EA, O, torch_cuda):
    """configs[3] on one GPU: 100 000 walkers, p = q = 1, 40 steps on a 1M-node
    power-law graph; 256 walkers bit-exact against the oracle (rows exported
    from HBM), all walkers: every step follows an edge of the graph's CSR."""
    torch = torch_cuda
    N = 1_000_000
    p = EA.synth_params(9, N, 10 * N, weighted=True)
    G = EA.Graph.synthetic(p)
    rng = np.random.default_rng(3)
    starts = rng.integers(1, N + 1, 100_000).astype(np.int64)
    L = 40
    et = [[0]] * L
    G.set_seed(5)
    walks = t2n(G.random_walk(torch.as_tensor(starts).cuda(), et, 1.0, 1.0, N + 1,
                              call_id=100))
    assert walks.shape == (100_000, L + 1) and np.array_equal(walks[:, 0], starts)
    assert walks.min() >= 1 and walks.max() <= N       # min degree 1: no dead ends
    sel = np.arange(256)
    need = np.unique(walks[sel].reshape(-1)).astype(np.uint64)
    rp, te, nb, pw, tp = G.export_rows(need)
    OG = O.OracleGraph(O.CSR(need, rp, te, nb, pw, tp, 1))
    want = OG.random_walk(5, 100, starts[sel], et, L, 1.0, 1.0, N + 1)
    assert np.array_equal(walks[sel], want)


@pytest.mark.gpu
def test_config5_typed_sampling_and_aggregation(EA, O, torch_cuda):
    """configs[4] on one GPU: 8 edge types; per-type sampling with one listed
    type, 3 of 8 (sub-collection draw) and all 8 (type draw over all groups),
    128-d features gathered for the sampled block and scatter_mean into the
    roots - ids / weights / types bit-exact, aggregate within 1e-5 (it is
    bit-exact: the segment reduce keeps the reference's order of additions)."""
    torch = torch_cuda
    N, T, D = 60_000, 8, 128
    po = O.synth_params(77, N, 40 * N, n_types=T, weighted=True)
    csr = O.synth_csr(po)
    p = EA.synth_params(77, N, 40 * N, n_types=T, weighted=True)
    G = EA.Graph.synthetic(p)
    OG = O.OracleGraph(csr)
    rng = np.random.default_rng(4)
    roots = rng.integers(1, N + 1, 20_000).astype(np.int64)
    rt = torch.as_tensor(roots).cuda()
    G.set_seed(8)
    for call, et in enumerate(([3], [1, 4, 6], list(range(T)), [])):
        on, ow, ot = OG.sample_neighbor(8, call, roots, et, 10, N + 1)
        gn, gw, gt = G.sample_neighbor(rt, et, 10, N + 1, call_id=call)
        assert np.array_equal(t2n(gn), on), et
        assert np.array_equal(t2n(gw), ow)
        assert np.array_equal(t2n(gt), ot)
        if len(et) == 1:
            assert set(np.unique(ot).tolist()) <= {3, -1}
    # features of the sampled block -> mean per root (RGCN-style aggregation)
    feat = torch.randn(N + 2, D, device="cuda")
    idx = gn.reshape(-1).to(torch.int32)
    x = EA.ops.gather(feat, idx)
    want_x = O.gather(t2n(feat), t2n(idx))
    assert np.array_equal(t2n(x), want_x)
    dst = torch.arange(len(roots), device="cuda", dtype=torch.int32).repeat_interleave(10)
    agg = EA.ops.scatter_mean(x, dst, len(roots))
    want = O.scatter_mean(want_x, t2n(dst), len(roots))
    assert np.allclose(t2n(agg), want, rtol=0, atol=1e-5)
    assert np.array_equal(t2n(agg), want)


@pytest.mark.gpu
def test_config5_typed_sampling_at_full_size(EA, O, torch_cuda):
    """configs[4] at the size SURVEY 8(d) states and bench.py runs ("same N" as the metric graph):
    100M nodes / 1B edges, 8 edge types, D = 128 features (51 GB), 131 072 roots x 10 samples
    for one listed type, 3 of 8 and all 8.  Two independent device paths agree on every
    sample (type draws on the block pivots == the reference loop, tuning key 37), with and
    without the duplicate-root machinery forced (key 5), the result is a function of
    (seed, call id, roots), 96 roots equal the oracle fed with their rows exported from
    HBM, and the one-pass aggregation straight from the int64 ids equals gather +
    scatter_mean bit for bit."""
    torch = torch_cuda
    from euler_amd import _lib
    L = _lib.lib()
    N, T, B, CNT, D = 100_000_000, 8, 131072, 10, 128
    G = EA.Graph.synthetic(EA.synth_params(20240521, N, 10 * N, n_types=T, weighted=True))
    G.set_seed(20240521)
    gen = torch.Generator(device="cuda"); gen.manual_seed(9)
    roots = torch.randint(1, N + 1, (B,), generator=gen, device="cuda", dtype=torch.int64)
    sel = np.random.default_rng(1).choice(B, 96, replace=False)
    r_sel = t2n(roots)[sel]
    rp, te, nb, pw, tp = G.export_rows(np.unique(r_sel).astype(np.uint64))
    OG = O.OracleGraph(O.CSR(np.unique(r_sel).astype(np.uint64), rp, te, nb, pw, tp, T))
    feat = torch.randn(N + 2, D, device="cuda", generator=torch.Generator("cuda").manual_seed(7))
    dst = torch.arange(B, device="cuda", dtype=torch.int32).repeat_interleave(CNT)
    try:
        for call, et in enumerate(([3], [1, 4, 6], list(range(T)))):
            L.euler_gpu_set_tuning(37, 1); L.euler_gpu_set_tuning(5, 1)
            a = G.sample_neighbor(roots, et, CNT, N + 1, call_id=40 + call)
            a2 = G.sample_neighbor(roots, et, CNT, N + 1, call_id=40 + call)
            L.euler_gpu_set_tuning(37, 0)
            b = G.sample_neighbor(roots, et, CNT, N + 1, call_id=40 + call)
            L.euler_gpu_set_tuning(37, 1); L.euler_gpu_set_tuning(5, 2)
            c = G.sample_neighbor(roots, et, CNT, N + 1, call_id=40 + call)
            for x in range(3):
                assert torch.equal(a[x], a2[x]) and torch.equal(a[x], b[x]) and torch.equal(a[x], c[x]), (et, x)
            on, ow, ot = OG.sample_neighbor(20240521, 40 + call, r_sel, et, CNT, N + 1)
            assert np.array_equal(t2n(a[0])[sel], on.reshape(-1, CNT)), et
            assert np.array_equal(t2n(a[1])[sel], ow.reshape(-1, CNT))
            assert np.array_equal(t2n(a[2])[sel], ot.reshape(-1, CNT))
            agg = EA.ops.gather_segment_reduce("mean", feat, a[0].reshape(-1), B, count=CNT)
            ref = EA.ops.scatter_mean(EA.ops.gather(feat, a[0].reshape(-1).to(torch.int32)), dst, B)
            assert torch.equal(agg, ref), et
    finally:
        L.euler_gpu_set_tuning(37, 1); L.euler_gpu_set_tuning(5, 1)


@pytest.mark.gpu
def test_three_hop_fanout_through_the_duplicate_path(EA, O, torch_cuda):
    """Fanout [25, 10, 5] of 4096 roots on a 2M-node graph: hops 2 and 3
    (102 400 and 1 024 000 roots) go through the on-device duplicate-root path;
    64 roots are checked bit for bit against the oracle fed with the rows
    exported from HBM."""
    torch = torch_cuda
    N = 2_000_000
    p = EA.synth_params(11, N, 10 * N, weighted=True)
    G = EA.Graph.synthetic(p)
    rng = np.random.default_rng(6)
    roots = rng.integers(1, N + 1, 4096).astype(np.int64)
    fan = [25, 10, 5]
    et = [[0], [0], [0]]
    G.set_seed(13)
    nb, ws, ts = G.sample_fanout(torch.as_tensor(roots).cuda(), et, fan, N + 1, call_id=30)
    assert [x.numel() for x in nb] == [4096, 102400, 1024000, 5120000]
    sel = np.arange(64)
    h1 = t2n(nb[1]).reshape(4096, 25)[sel]
    h2 = t2n(nb[2]).reshape(4096, 250)[sel]
    need = np.unique(np.concatenate([roots[sel], h1.reshape(-1), h2.reshape(-1)])).astype(np.uint64)
    need = need[(need >= 1) & (need <= N)]
    rp, te, nbr, pw, tp = G.export_rows(need)
    OG = O.OracleGraph(O.CSR(need, rp, te, nbr, pw, tp, 1))
    on, ow, ot = OG.sample_fanout(13, 30, roots[sel], et, fan, N + 1)
    assert np.array_equal(on[0], h1.reshape(-1))
    assert np.array_equal(on[1], h2.reshape(-1))
    assert np.array_equal(on[2], t2n(nb[3]).reshape(4096, 1250)[sel].reshape(-1))
    assert np.array_equal(ow[2], t2n(ws[2]).reshape(4096, 1250)[sel].reshape(-1))
