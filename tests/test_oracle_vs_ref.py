"""The C restatement against the REFERENCE sampler itself (oracle/_ref =
reference sources + RNG seam), on fresh random graphs.  Skipped when the
prebuilt oracle/_ref/libeuler_ref.so is absent.  CPU only."""
import numpy as np
import pytest

from conftest import make_random_graph


@pytest.fixture(scope="module")
def pair(O):
    if not O.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(2024)
    ids, seg, nbr, w, nt, nw = make_random_graph(rng, 3000, 4, max_deg=20)
    # a hub row and an id-0 node exercise deep searches and the sentinel quirk
    R = O.RefGraph.build_raw(ids, seg, nbr, w, 4, nt, nw)
    csr = O.csr_from_raw(ids, seg, nbr, w, 4, nt, nw)
    ref_csr = R.export_csr(ids)
    for a in ("row_ptr", "type_end", "nbr", "prefix_w", "type_prefix"):
        assert np.array_equal(getattr(csr, a), getattr(ref_csr, a)), a
    return R, O.OracleGraph(csr), ids, rng


@pytest.mark.parametrize("et", [[0], [3], [1, 2], [3, 0, 1], [0, 1, 2, 3], [],
                                [2, 2], [9], [1, 9], [0, 1, 2, 3, 0]])
@pytest.mark.parametrize("count", [1, 10])
def test_sample_neighbor_matches_reference(pair, et, count):
    R, G, ids, rng = pair
    q = np.concatenate([rng.choice(ids, 800), [0, 2 ** 63 + 5]]).astype(np.uint64)
    for call in (0, 77):
        a = R.sample_neighbor_core(99, call, q, et, count)
        b = G.sample_neighbor_core(99, call, q, et, count)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_sample_node_matches_reference(pair):
    R, G, ids, rng = pair
    G.build_node_sampler(order=R.node_order())
    for call, types in enumerate([[-1], [0], [1], [0, 1], [1, 0, 1]]):
        a = R.sample_node(5, call, types, 500)
        b = G.sample_node(5, call, types, 500)
        assert len(a) == 500 and np.array_equal(a, b)


def test_full_neighbor_and_walks_match_reference(pair):
    R, G, ids, rng = pair
    q = np.concatenate([rng.choice(ids, 300), [0]]).astype(np.uint64)
    for et in ([0], [1, 3], [3, 1], [0, 1, 2, 3]):
        for x, y in zip(R.get_full_neighbor(q, et), G.get_full_neighbor(q, et)):
            assert np.array_equal(x, y)
    L = 8
    et = np.tile(np.array([0, 1, 2, 3], np.int32), (L, 1))
    starts = q.astype(np.int64)
    for p, qq, dn in ((1.0, 1.0, -1), (1.0, 1.0 + 5e-7, 12345), (0.5, 2.0, -1),
                      (4.0, 0.25, 4242)):
        a = R.random_walk(3, 1000, starts, et, L, p, qq, dn)
        b = G.random_walk(3, 1000, starts, et, L, p, qq, dn)
        assert np.array_equal(a, b), (p, qq)


def test_graph_with_node_id_zero(O):
    """Q1: a real node 0 is indistinguishable from the sentinel."""
    if not O.have_ref():
        pytest.skip("oracle/_ref not built")
    ids = np.array([0, 1, 2, 3], np.uint64)
    seg = np.array([0, 2, 4, 6, 7], np.int64)
    nbr = np.array([1, 2, 0, 2, 0, 3, 0], np.uint64)
    w = np.array([1, 1, 5, 1, 1, 1, 2], np.float32)
    R = O.RefGraph.build_raw(ids, seg, nbr, w, 1)
    G = O.OracleGraph(O.csr_from_raw(ids, seg, nbr, w, 1))
    q = np.array([0, 1, 2, 3, 9], np.uint64)
    for call in range(20):
        a = R.sample_neighbor_core(1, call, q, [0], 3)
        b = G.sample_neighbor_core(1, call, q, [0], 3)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    starts = q.astype(np.int64)
    et = np.zeros((5, 1), np.int32)
    assert np.array_equal(R.random_walk(1, 0, starts, et, 5, 1.0, 1.0, -1),
                          G.random_walk(1, 0, starts, et, 5, 1.0, 1.0, -1))


def test_dense_features_match_reference(O):
    """Random ragged float features pushed into the reference's Node storage:
    Node::GetFloat32Feature + TF copy loop == restatement."""
    rng = np.random.default_rng(21)
    n = 200
    ids = np.sort(rng.choice(np.arange(1, 10 ** 6), n, replace=False)).astype(np.uint64)
    seg = np.arange(n + 1, dtype=np.int64)
    nbr = rng.choice(ids, n).astype(np.uint64)
    w = np.ones(n, np.float32)
    csr = O.csr_from_raw(ids, seg, nbr, w, 1)
    per = [[list(rng.standard_normal(int(rng.integers(0, 9))).astype(np.float32))
            for _ in range(int(rng.integers(0, 4)))] for _ in range(n)]
    per[0] = [[1.0] * 8, [2.0] * 8, [3.0] * 8]
    F = O.DenseFeatures.from_lists(per)
    R = O.RefGraph.build_raw(ids, seg, nbr, w, 1)
    R.set_float_features(ids, F)
    q = np.concatenate([rng.choice(ids, 500), [0, 7]]).astype(np.int64)
    fids, dims = [0, 1, 2, 3, -1], [8, 8, 9, 4, 2]
    a = O.OracleGraph(csr).get_dense_feature(F, q, fids, dims)
    b = R.get_dense_feature(q, fids, dims)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_sparse_features_match_reference(O):
    """Random ragged uint64 features pushed into the reference's Node storage:
    Node::GetUint64Feature + the TF SparseTensor builder == restatement."""
    if not O.have_ref():
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(22)
    n = 200
    ids = np.sort(rng.choice(np.arange(1, 10 ** 6), n, replace=False)).astype(np.uint64)
    seg = np.arange(n + 1, dtype=np.int64)
    nbr = rng.choice(ids, n).astype(np.uint64)
    w = np.ones(n, np.float32)
    csr = O.csr_from_raw(ids, seg, nbr, w, 1)
    per = [[list(rng.integers(0, 2 ** 64, int(rng.integers(0, 9)), dtype=np.uint64))
            for _ in range(int(rng.integers(0, 4)))] for _ in range(n)]
    per[0] = [[1] * 8, [], [3] * 90]
    F = O.SparseFeatures.from_lists(per)
    R = O.RefGraph.build_raw(ids, seg, nbr, w, 1)
    R.set_u64_features(ids, F)
    q = np.concatenate([rng.choice(ids, 500), [0, 7]]).astype(np.uint64)
    fids, dvs = [0, 1, 2, 3, -1], [0, 5, -1, 4, 2]
    a = O.OracleGraph(csr).get_sparse_feature(F, q, fids, dvs)
    b = R.get_sparse_feature(q, fids, dvs)
    for (ai, av, ash), (bi, bv, bsh) in zip(a, b):
        assert np.array_equal(ai, bi) and np.array_equal(av, bv) and np.array_equal(ash, bsh)
    # empty query
    e = O.OracleGraph(csr).get_sparse_feature(F, np.zeros(0, np.uint64), [0])[0]
    assert e[0].shape == (0, 2) and list(e[2]) == [0, 0]


def test_neighbor_post_process_matches_reference_sort(O):
    """order_by id / weight, asc / desc, limit: restatement == the reference's
    comparators under std::sort, on rows whose keys are distinct (equal keys
    have no defined order in the reference: its comparator is `<=`)."""
    rng = np.random.default_rng(33)
    n, T = 300, 3
    ids = np.sort(rng.choice(np.arange(1, 10 ** 6), n, replace=False)).astype(np.uint64)
    deg = rng.integers(0, 30, size=(n, T))
    seg = np.zeros(n * T + 1, np.int64)
    seg[1:] = np.cumsum(deg.reshape(-1))
    E = int(seg[-1])
    nbr = np.zeros(E, np.uint64)
    w = np.zeros(E, np.float32)
    for i in range(n):          # distinct ids and distinct weights within a row
        b, e = seg[i * T], seg[(i + 1) * T]
        nbr[b:e] = rng.choice(ids, e - b, replace=False)
        w[b:e] = rng.permutation(e - b).astype(np.float32) * 0.37 + 0.5
    csr = O.csr_from_raw(ids, seg, nbr, w, T)
    OG = O.OracleGraph(csr)
    R = O.RefGraph.build_raw(ids, seg, nbr, w, T)
    q = np.concatenate([rng.choice(ids, 400), [0, 5]]).astype(np.uint64)
    for et in ([0, 1, 2], [2, 0], [1]):
        full = OG.get_full_neighbor(q, et)
        for order_by, desc, limit in (("id", False, None), ("id", True, 3),
                                      ("weight", True, 5), ("weight", False, None),
                                      (None, False, 2)):
            a = O.neighbor_post_process(*full, order_by=order_by, desc=desc, limit=limit)
            b = R.get_neighbor(q, et, order_by, desc, limit)
            for x, y in zip(a, b):
                assert np.array_equal(x, y), (et, order_by, desc, limit)


# ------------------------------------------------------------------ layerwise
@pytest.fixture(scope="module")
def lpair(O):
    """Own graph pair: the reference graph is a process-wide singleton and the
    tests above rebuild it, so it is (re)built here, with Edge records."""
    if not O.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(4048)
    ids, seg, nbr, w, nt, nw = make_random_graph(rng, 3000, 4, max_deg=20)
    R = O.RefGraph.build_raw(ids, seg, nbr, w, 4, nt, nw)
    assert R.add_edges_from_adjacency() > 0
    return R, O.OracleGraph(O.csr_from_raw(ids, seg, nbr, w, 4, nt, nw)), ids, rng


def _layer_batches(rng, ids, batch, n, dup=True):
    nodes = rng.choice(ids, (batch, n)).astype(np.uint64)
    if dup:                       # duplicates inside a row, an unknown id, a 0
        nodes[0, 1] = nodes[0, 0]
        nodes[1, 0] = 2 ** 62 + 3
        nodes[-1, -1] = 0
    return nodes


@pytest.mark.parametrize("et", [[0], [3], [1, 2], [3, 0, 1], [0, 1, 2, 3], [],
                                [9], [1, 9]])
def test_layerwise_primitives_match_reference(lpair, et):
    R, G, ids, rng = lpair
    q = np.concatenate([rng.choice(ids, 500), [0, 2 ** 63 + 5]]).astype(np.uint64)
    a, b = R.get_edge_sum_weight(q, et), G.get_edge_sum_weight(q, et)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    for call, dn in ((0, -1), (5, 424242)):
        x = R.sample_layer(7, call, q, et, dn)
        y = G.sample_layer(7, call, q, et, dn)
        for u, v in zip(x, y):
            assert np.array_equal(u, v)
    # a position-keyed stream: the same root at two positions draws twice
    rep = np.repeat(q[:1], 64)
    ids_rep = G.sample_layer(7, 1, rep, [0, 1, 2, 3], -1)[0]
    if G.get_edge_sum_weight(q[:1], [0, 1, 2, 3])[0] > 0 and \
            len(set(G.get_full_neighbor(q[:1], [0, 1, 2, 3])[1].tolist())) > 1:
        assert len(set(ids_rep.tolist())) > 1


def test_sample_root_matches_reference(lpair, O):
    R, G, ids, rng = lpair
    for n, m in ((1, 3), (2, 5), (7, 16), (64, 10), (301, 33)):
        batch = 9
        roots = rng.choice(ids, (batch, n)).astype(np.uint64)
        w = (rng.random((batch, n)) * 5).astype(np.float32)
        w[rng.random((batch, n)) < 0.3] = 0
        w[2] = 0                                     # zero-sum row -> default node
        if n > 2:
            w[3] = 0; w[3, 1] = 2.5                  # a single heavy entry
            w[4] = 1.0                               # uniform weights
        for call, dn in ((0, -1), (3, 77)):
            a = R._sample_root(11, call, roots, w, n, m, dn)
            b = O.sample_root(11, call, roots, w, n, m, dn)
            assert np.array_equal(a, b), (n, m)
            assert np.all(b.reshape(batch, m)[2] == np.uint64(dn & (2 ** 64 - 1)))


def test_sparse_get_adj_matches_reference(lpair):
    R, G, ids, rng = lpair
    assert R.num_edges() > 0
    # EdgeExist from the reference's Edge map == membership in the adjacency rows
    csr = G.csr
    for r in rng.choice(len(csr.row_id), 50):
        src = int(csr.row_id[r])
        full = G.get_full_neighbor(np.array([src], np.uint64), [0, 1, 2, 3])
        for dst, t in zip(full[1][:6], full[3][:6]):
            assert R.edge_exist(src, int(dst), int(t))
        assert not R.edge_exist(src, 2 ** 61 + 1, 0)
    for batch, n, m in ((1, 5, 7), (3, 4, 70), (2, 1, 1), (4, 9, 130)):
        nodes = _layer_batches(rng, ids, batch, n, dup=n > 1 and batch > 1)
        # candidates: real neighbours of the row's nodes mixed with random ids
        cand = rng.choice(ids, (batch, m)).astype(np.uint64)
        for b in range(batch):
            nb = G.get_full_neighbor(nodes[b], [0, 1, 2, 3])[1]
            if len(nb):
                take = rng.choice(nb, m // 2 + 1)
                cand[b, :len(take)] = take
        for et in ([0], [1, 3], [0, 1, 2, 3], [], [9]):
            x = R.sparse_get_adj(nodes, cand, batch, n, m, et)
            y = G.sparse_get_adj(nodes, cand, batch, n, m, et)
            assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
            sx = R._adj_to_sparse(nodes, cand, batch, n, m, *x)
            sy = G._adj_to_sparse(nodes, cand, batch, n, m, *y)
            for u, v in zip(sx, sy):
                assert np.array_equal(u, v)
            assert list(sy[2]) == [batch, n, m]


def test_sample_neighbor_layerwise_matches_reference(lpair):
    R, G, ids, rng = lpair
    for batch, n, count in ((4, 3, 10), (2, 16, 5), (6, 1, 4)):
        nodes = _layer_batches(rng, ids, batch, n, dup=n > 1)
        for et in ([0], [0, 1], [0, 1, 2, 3]):
            for call, dn in ((0, -1), (9, 31337)):
                x = R.sample_neighbor_layerwise(21, call, nodes, et, count, dn)
                y = G.sample_neighbor_layerwise(21, call, nodes, et, count, dn)
                for u, v in zip(x, y):
                    assert np.array_equal(u, v)


def test_node_type_and_sample_n_with_types_match_reference(lpair):
    R, G, ids, rng = lpair
    q = np.concatenate([rng.choice(ids, 300), [0, 2 ** 63 + 5]]).astype(np.uint64)
    assert np.array_equal(R.get_node_type(q), G.get_node_type(q))
    assert G.get_node_type(q)[-1] == -2 ** 31
    G.build_node_sampler(order=R.node_order())
    types = np.concatenate([G.get_node_type(q[:300]), [-1, -1, 0, 1]]).astype(np.int32)
    for call, count in ((0, 1), (1, 5), (2, 16)):
        a = R.sample_n_with_types(13, call, types, count)
        b = G.sample_n_with_types(13, call, types, count)
        assert a is not None and np.array_equal(a, b)
    # rows are independent SampleNode calls: row i == SampleNode with stream i
    assert R.sample_n_with_types(13, 0, [0, 7], 3) is None
    assert G.sample_n_with_types(13, 0, [0, 7], 3) is None


def test_unordered_map_order_and_local_sample_layer_match_reference(lpair, O):
    """oracle/eo_umap.c (libstdc++'s string hash, prime rehash policy, list
    insertion) == the real std::unordered_map inside oracle/_ref: iteration order
    on random key sets across many rehashes, then the whole API_LOCAL_SAMPLE_L."""
    R, G, ids, rng = lpair
    for trial in range(60):
        n = int(rng.integers(0, 50 if trial < 20 else 3000 if trial < 55 else 40000))
        keys = list({"%d%d" % (rng.integers(0, 2 ** 63), rng.integers(0, 5)) for _ in range(n)})
        want, hashes = O.ref_umap_iteration_order(keys)
        assert np.array_equal(hashes, np.array([O.std_hash(k) for k in keys], np.uint64))
        assert np.array_equal(want, O.umap_iteration_order(keys)), len(keys)
    # to_string(id) + to_string(type) can run two pairs together (id 12 / type 3,
    # id 1 / type 23): the reference merges them, so does the restatement
    idx = np.array([[0, 4]], np.int32)
    ids4 = np.array([12, 1, 12, 7], np.uint64)
    w4 = np.array([1.0, 2.0, 4.0, 8.0], np.float32)
    t4 = np.array([3, 23, 3, 0], np.int32)
    a = R.local_sample_layer(3, 1, idx, ids4, w4, t4, 1, 32, "sqrt", -1)
    b = G.local_sample_layer(3, 1, idx, ids4, w4, t4, 1, 32, "sqrt", -1)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert set(np.round(a[1] ** 2).astype(int).tolist()) <= {7, 8}
    for batch, n, count in ((4, 3, 10), (2, 50, 40), (7, 1, 5), (1, 300, 64)):
        nodes = rng.choice(ids, (batch, n)).astype(np.uint64)
        nodes[-1, -1] = 2 ** 62 + 1
        for et in ([0], [1, 2], [0, 1, 2, 3]):
            for wf, dn in (("sqrt", -1), ("sqrt", 261), ("id", 0)):
                x = R.sample_neighbor_layerwise_func(9, 2, nodes, et, count, wf, dn)
                y = G.sample_neighbor_layerwise_func(9, 2, nodes, et, count, wf, dn)
                for u, v in zip(x, y):
                    if u.dtype == np.float32:
                        u, v = u.view(np.uint32), v.view(np.uint32)
                    assert np.array_equal(u, v), (batch, n, et, wf)
