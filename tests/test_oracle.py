"""Pin the CPU oracle (oracle/euler_oracle.c): known-answer tests, the exact
expectations of the reference's own tests, and the golden vectors produced by
the reference sampler (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

SEED = 20240521


# ---------------------------------------------------------------- RNG KATs
def test_philox_known_answers(O):
    # Random123 kat_vectors, philox4x32-10
    assert O.philox([0, 0, 0, 0], [0, 0]) == [
        0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox([0xffffffff] * 4, [0xffffffff] * 2) == [
        0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                    [0xa4093822, 0x299f31d0]) == [
        0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_uniform_contract(O):
    # u = ((w[2h]>>5)*2^26 + (w[2h+1]>>6)) * 2^-53, block = draw>>1
    seed, call, stream = 0x1234567890abcdef, 17, 0xfedcba9876543210
    for domain, salt in ((0, 0), (1, 0x9E3779B9), (2, 0x7F4A7C15), (3, 0xF39CC060),
                         (4, 0x6A09E667), (5, 0xB5C0FBCF), (6, 0x3C6EF372)):
        for d in range(6):
            w = O.philox([call, stream & 0xffffffff, stream >> 32, d >> 1],
                         [seed & 0xffffffff, (seed >> 32) ^ salt])
            a, b = w[2 * (d & 1)], w[2 * (d & 1) + 1]
            u = ((a >> 5) * 67108864 + (b >> 6)) / 9007199254740992.0
            assert O.uniform_at(seed, call, domain, stream, d) == u
            assert 0.0 <= u < 1.0


# --------------------------------------------- reference test expectations
def test_mp_ops_reference_goldens(O, ref_tests):
    t = ref_tests
    assert np.array_equal(O.scatter_add(t["scatter_add_x"], t["scatter_idx"], 2),
                          t["scatter_add_out"])
    assert np.abs(O.scatter_mean(t["scatter_add_x"], t["scatter_idx"], 2)
                  - t["scatter_mean_out"]).sum() < 1e-6
    assert np.array_equal(O.scatter_max(t["scatter_max_x"], t["scatter_idx"], 2),
                          t["scatter_max_out"])
    assert np.array_equal(O.gather(t["gather_x"], t["gather_idx"]),
                          t["gather_out"])
    # ScatterMax empty segment = -1e9 (scatter_op.cc:78)
    out = O.scatter_max(t["scatter_max_x"], t["scatter_idx"], 3)
    assert np.all(out[2] == np.float32(-1e9))


def test_gen_pair_reference_golden(O, ref_tests):
    out = O.gen_pair(ref_tests["gen_pair_in"], 2, 2)
    assert np.array_equal(out, ref_tests["gen_pair_out"])


def test_unique_gather_reference_goldens(O):
    # euler/core/kernels/unique_gather_test.cc:28-160
    uq, gi = O.id_unique([1, 2, 3, 3, 2, 2, 4])
    assert uq.tolist() == [1, 2, 3, 4]
    assert gi.tolist() == [0, 1, 2, 2, 1, 1, 3]
    idx = np.array([[0, 2], [2, 5], [5, 6]], np.int32)
    gather_idx = [0, 1, 0, 2, 1]
    assert O.idx_gather(idx, gather_idx).reshape(-1).tolist() == [
        0, 2, 2, 5, 5, 7, 7, 8, 8, 11]
    data = np.array([11, 12, 21, 22, 23, 31], np.uint64)
    assert O.data_gather(data, idx, gather_idx).tolist() == [
        11, 12, 21, 22, 23, 11, 12, 31, 21, 22, 23]
    fdata = data.astype(np.float32)
    assert O.data_gather(fdata, idx, gather_idx).tolist() == [
        11, 12, 21, 22, 23, 11, 12, 31, 21, 22, 23]


def test_inflate_idx_reference_expectations(O):
    # tf_euler/python/euler_ops/util_ops_test.py:44-58; inflate_idx_op.cc:53-55 (range check)
    assert O.inflate_idx([0, 2, 1, 3]).tolist() == [0, 2, 1, 3]
    assert O.inflate_idx([0, 1, 0, 2, 1]).tolist() == [0, 2, 1, 4, 3]
    assert O.inflate_idx([]).tolist() == []
    for bad in ([0, 2], [0, -1]):
        try:
            O.inflate_idx(bad)
        except ValueError:
            continue
        assert False, bad


def test_full_neighbor_reference_goldens(O, fixture_csr):
    # tf_euler/python/euler_ops/neighbor_ops_test.py:46-75 and
    # GetFullNeighbor({1,2},{0,1}) -> [[2,4,3],[3,5]]
    G = O.OracleGraph(fixture_csr)
    idx, ids, w, t = G.get_full_neighbor([1, 2], [0, 1])
    assert idx.tolist() == [[0, 3], [3, 5]]
    assert ids.tolist() == [2, 4, 3, 3, 5]
    assert w.tolist() == [2, 4, 3, 3, 5]
    assert t.tolist() == [0, 0, 1, 1, 1]


def test_random_select_semantics(O):
    sw = np.array([1, 1, 3, 3, 6], np.float32)      # weights 1,0,2,0,3
    # zero-weight entries are never selected
    picks = {O.random_select(sw, 0, 4, u) for u in np.linspace(0, 0.999999, 2001)}
    assert picks == {0, 2, 4}
    # sub-range [2,4]: limit_begin = sw[1]
    picks = {O.random_select(sw, 2, 4, u) for u in np.linspace(0, 0.999999, 2001)}
    assert picks == {2, 4}
    # all-zero range: falls through and returns the last probed mid
    z = np.array([2, 2, 2, 2], np.float32)
    assert O.random_select(z, 1, 3, 0.5) == 3


# -------------------------------------------- goldens made by the reference
def _check_pack(O, csr, s):
    G = O.OracleGraph(csr)
    q = s["query_ids"]
    seed = int(s["seed"]) if "seed" in s.files else SEED
    n = 0
    while "nb_%d_1_et" % n in s.files:
        for count in (1, 5):
            key = "nb_%d_%d_" % (n, count)
            idx, oid, ow, ot = G.sample_neighbor_core(seed, 11 + n, q,
                                                      s[key + "et"], count)
            assert np.array_equal(idx, s[key + "idx"]), key
            assert np.array_equal(oid, s[key + "id"]), key
            assert np.array_equal(ow, s[key + "w"]), key
            assert np.array_equal(ot, s[key + "t"]), key
        n += 1
    assert n >= 7
    n = 0
    while "full_%d_et" % n in s.files:
        key = "full_%d_" % n
        idx, oid, ow, ot = G.get_full_neighbor(q, s[key + "et"])
        assert np.array_equal(idx, s[key + "idx"])
        assert np.array_equal(oid, s[key + "id"])
        assert np.array_equal(ow, s[key + "w"])
        assert np.array_equal(ot, s[key + "t"])
        n += 1
    # global node sampler: tables and samples
    G.build_node_sampler(order=s["node_order"])
    for t in (0, 1):
        sel = s["alias_%d_ids" % (t + 1)]
        # FastWeightedCollection::Init re-normalises by the sequential f32 sum
        w = s["alias_%d_w" % (t + 1)]
        acc = np.float32(0)
        for x in w:
            acc = np.float32(acc + x)
        prob, alias = O.alias_init(w / acc)
        assert np.array_equal(prob, s["alias_%d_prob" % (t + 1)])
        assert np.array_equal(alias, s["alias_%d_alias" % (t + 1)])
        assert len(sel) == int((csr.node_type == t).sum())
    for n in range(4):
        got = G.sample_node(seed, 100 + n, s["sn_%d_types" % n], 64)
        assert np.array_equal(got, s["sn_%d" % n]), n
    et = s["walk_et"]
    starts = q.astype(np.int64)
    L = et.shape[0]
    assert np.array_equal(G.random_walk(seed, 200, starts, et, L, 1.0, 1.0, -1),
                          s["walk_11"])
    assert np.array_equal(G.random_walk(seed, 300, starts, et, L, 0.25, 4.0, -1),
                          s["walk_n2v"])
    assert np.array_equal(G.random_walk(seed, 400, starts, et, L, 2.0, 0.5, 777),
                          s["walk_n2v_b"])


def test_fixture_goldens(O, fixture_csr, fixture_samples):
    _check_pack(O, fixture_csr, fixture_samples)


def test_random_graph_goldens(O, random_csr, random_samples):
    _check_pack(O, random_csr, random_samples)


def test_prefix_build_matches_reference_node_init(O, random_samples):
    s = random_samples
    csr = O.csr_from_raw(s["row_id"], s["raw_seg_ptr"], s["raw_nbr"], s["raw_w"],
                         int(s["n_types"]))
    assert np.array_equal(csr.prefix_w, s["prefix_w"])
    assert np.array_equal(csr.type_prefix, s["type_prefix"])
    assert np.array_equal(csr.type_end, s["type_end"])
    assert np.array_equal(csr.row_ptr, s["row_ptr"])


# ------------------------------------------------------- TF-level layouts
def test_tf_layout_and_sentinel(O, fixture_csr):
    G = O.OracleGraph(fixture_csr)
    nodes = np.array([1, 3, 99, 0, 6], np.int64)
    n, w, t = G.sample_neighbor(SEED, 5, nodes, [0], 4, default_node=-1)
    idx, cid, cw, ct = G.sample_neighbor_core(SEED, 5, nodes.astype(np.uint64),
                                              [0], 4)
    cid = cid.reshape(5, 4)
    # node 99/0 unknown, node 6 has no type-0 neighbour: default rows
    for r in (2, 3, 4):
        assert n[r].tolist() == [-1] * 4
        assert w[r].tolist() == [0.0] * 4
        assert t[r].tolist() == [-1] * 4
        assert cid[r].tolist() == [0] * 4
    assert np.array_equal(n[:2], cid[:2].astype(np.int64))
    assert set(n[0].tolist()) <= {2, 4} and set(n[1].tolist()) == {4}
    # fanout chains on CORE ids; flattening as neighbor_ops.py:122-158
    ns, ws, ts = G.sample_fanout(SEED, 9, nodes, [[0, 1], [0, 1]], [3, 2], -1)
    assert ns[0].shape == (15,) and ns[1].shape == (30,)
    n1, _, _ = G.sample_neighbor(SEED, 9, nodes, [0, 1], 3, -1)
    assert np.array_equal(ns[0], n1.reshape(-1))
    _, cid1, _, _ = G.sample_neighbor_core(SEED, 9, nodes.astype(np.uint64),
                                           [0, 1], 3)
    n2, _, _ = G.sample_neighbor(SEED, 10, cid1.astype(np.int64), [0, 1], 2, -1)
    assert np.array_equal(ns[1], n2.reshape(-1))


def test_duplicate_roots_get_identical_samples(O, random_csr):
    # F5: the GQL optimizer samples once per distinct id and gathers
    G = O.OracleGraph(random_csr)
    ids = random_csr.row_id[:20]
    dup = np.concatenate([ids, ids[::-1], ids[:5]])
    _, oid, ow, ot = G.sample_neighbor_core(SEED, 3, dup, [0, 2], 7)
    uq, gi = O.id_unique(dup)
    idx_u, oid_u, ow_u, ot_u = G.sample_neighbor_core(SEED, 3, uq, [0, 2], 7)
    assert np.array_equal(O.data_gather(oid_u, idx_u, gi), oid)
    assert np.array_equal(O.data_gather(ow_u, idx_u, gi), ow)
    assert np.array_equal(O.data_gather(ot_u, idx_u, gi), ot)


def test_statistical_ratio_like_reference(O, fixture_csr):
    # euler/core/graph/graph_test.cc:425-456 style: node 1 type-0 neighbours
    # 2 (w=2) and 4 (w=4) -> ratio 1:2 within +-20%
    G = O.OracleGraph(fixture_csr)
    cnt = {2: 0, 4: 0}
    for call in range(300):
        _, oid, _, _ = G.sample_neighbor_core(SEED, call, [1], [0], 30)
        for v in oid.tolist():
            cnt[v] += 1
    ratio = cnt[4] / cnt[2]
    assert 1.6 < ratio < 2.4


def test_shard_ops(O):
    ids = np.array([5, 12, 7, 8, 1024, 3, 16], np.uint64)
    off, sid, mi = O.id_split(ids, 8, 3)
    owner = O.shard_of(ids, 8, 3)
    assert owner.tolist() == [int((i % 8) % 3) for i in ids.tolist()]
    for s in range(3):
        seg = sid[off[s]:off[s + 1]]
        assert seg.tolist() == [int(i) for i in ids[owner == s].tolist()]
        assert np.array_equal(ids[mi[off[s]:off[s + 1]]], seg)
    split = O.sample_node_split(SEED, 1, 10, [1.0, 2.0, 0.0, 3.0])
    assert split.sum() == 10 and split[2] == 0
    assert split[0] >= 3 and split[1] >= 6


def test_synth_graph_properties(O):
    p = O.synth_params(99, 5000, 50000, n_types=2, weighted=True)
    csr = O.synth_csr(p)
    deg = np.diff(csr.row_ptr)
    assert deg.min() >= 1
    assert abs(int(deg.sum()) - 50000) < 2500
    assert deg.max() > 20 * np.median(deg)          # heavy tail
    assert csr.nbr.min() >= 1 and csr.nbr.max() <= 5000
    # rows regenerate identically from a sub-range
    sub = O.synth_csr(p, 100, 160)
    b, e = csr.row_ptr[100], csr.row_ptr[160]
    assert np.array_equal(sub.nbr, csr.nbr[b:e])
    assert np.array_equal(sub.prefix_w, csr.prefix_w[b:e])
    w0 = csr.prefix_w[csr.row_ptr[:-1]]
    assert w0.min() >= 0.5 and w0.max() < 8.0


def _golden_features(O, prefix):
    fg = np.load(os.path.join(os.path.dirname(__file__), "golden", "features.npz"))
    F = O.DenseFeatures(int(fg[prefix + "n_float"]), fg[prefix + "feat_ptr"],
                        fg[prefix + "feat_idx"], fg[prefix + "feat_val"])
    return fg, F


def test_dense_feature_goldens(O, fixture_csr, random_csr):
    """Restated GetDenseFeature == rows produced by the reference's
    GetFloat32Feature + the TF kernel's copy loop (tests/golden/features.npz):
    the fixture's own features as loaded by Node::DeSerialize, and ragged random
    features (missing slots, short rows, unknown nodes, slot out of range)."""
    for prefix, csr in (("fx_", fixture_csr), ("rg_", random_csr)):
        fg, F = _golden_features(O, prefix)
        got = O.OracleGraph(csr).get_dense_feature(F, fg[prefix + "query"],
                                                   fg[prefix + "fids"], fg[prefix + "dims"])
        for k, o in enumerate(got):
            assert np.array_equal(o, fg[prefix + "dense_%d" % k]), (prefix, k)
    # the fixture's values are the ones of tools/test_data/graph.json
    fg, F = _golden_features(O, "fx_")
    assert np.allclose(fg["fx_dense_0"][0], [1.1, 1.2])


def test_sparse_feature_goldens(O, fixture_csr, random_csr):
    """Restated GetSparseFeature == the SparseTensor triples produced by the
    reference's GetUint64Feature + the TF kernel's builder
    (tests/golden/sparse_features.npz): default entries for empty slots, unknown
    nodes, slots out of range (also negative)."""
    sg = np.load(os.path.join(os.path.dirname(__file__), "golden", "sparse_features.npz"))
    for prefix, csr in (("fx_", fixture_csr), ("rg_", random_csr)):
        F = O.SparseFeatures(int(sg[prefix + "n_u64"]), sg[prefix + "feat_ptr"],
                             sg[prefix + "feat_idx"], sg[prefix + "feat_val"])
        got = O.OracleGraph(csr).get_sparse_feature(F, sg[prefix + "query"],
                                                    sg[prefix + "fids"],
                                                    sg[prefix + "defaults"])
        assert len(got) == len(sg[prefix + "fids"])
        for k, (ind, val, shape) in enumerate(got):
            assert np.array_equal(ind, sg[prefix + "sp_%d_ind" % k]), (prefix, k)
            assert np.array_equal(val, sg[prefix + "sp_%d_val" % k]), (prefix, k)
            assert np.array_equal(shape, sg[prefix + "sp_%d_shape" % k]), (prefix, k)
    # node 1 of tools/test_data/graph.json: f1 = [11, 12]
    assert sg["fx_sp_0_val"][:2].tolist() == [11, 12]


def test_sorted_and_top_k_neighbor_reference_goldens(O, fixture_csr):
    """tf_euler/python/euler_ops/neighbor_ops_test.py:75-86 (sorted) and
    :101-109 (top-k) on the reference's fixture graph."""
    OG = O.OracleGraph(fixture_csr)
    full = OG.get_full_neighbor(np.array([1, 2], np.uint64), [0, 1])
    idx, ids, w, t = O.neighbor_post_process(*full, order_by="id")
    assert idx.tolist() == [[0, 3], [3, 5]]
    assert ids.tolist() == [2, 3, 4, 3, 5] and t.tolist() == [0, 1, 0, 1, 1]
    assert w.tolist() == [2.0, 3.0, 4.0, 3.0, 5.0]
    top = O.neighbor_post_process(*full, order_by="weight", desc=True, limit=2)
    di, dw, dt = O.neighbor_to_dense(*top, 2, -1)
    assert di.tolist() == [[4, 3], [5, 3]] and dt.tolist() == [[0, 1], [1, 1]]
    assert dw.tolist() == [[4.0, 3.0], [5.0, 3.0]]
    # fewer than k neighbours: default_node / 0.0 / -1 fill (:70-75)
    top = O.neighbor_post_process(*full, order_by="weight", desc=True, limit=4)
    di, dw, dt = O.neighbor_to_dense(*top, 4, -1)
    assert di[1].tolist() == [5, 3, -1, -1] and dt[1].tolist() == [1, 1, -1, -1]


# ------------------------------------------------------------------ layerwise
def test_layerwise_goldens(O, fixture_csr, random_csr):
    """sampleLNB op chain + SparseGetAdj against vectors produced by the
    reference (fixture loaded with its Edge records; random graph)."""
    import os
    from conftest import GOLDEN
    from layer_cases import OracleBackend, check_layer_pack
    L = np.load(os.path.join(GOLDEN, "layerwise.npz"))
    G = O.OracleGraph(fixture_csr)
    check_layer_pack(OracleBackend(O, G), L, "fx_", 2)
    # EdgeExist answered from the rows == the reference's Edge records
    have = {tuple(x) for x in L["fx_edges"].tolist()}
    ids = fixture_csr.row_id
    for s in ids:
        for d in list(ids) + [0, 99]:
            for t in (0, 1, 2):
                idx, vals = G.sparse_get_adj([s], [d], 1, 1, 1, [t])
                assert (len(vals) == 1) == ((int(s), int(d), t) in have)
    # the reference's own test batches (neighbor_ops_test.py:142-175)
    nb, ind, val, shape = G.sample_neighbor_layerwise(int(L["seed"]), 90, L["fx_t_nodes"],
                                                      [0, 1], 10, -1)
    assert np.array_equal(nb, L["fx_t_nb"]) and np.array_equal(ind, L["fx_t_ind"])
    assert np.array_equal(val, L["fx_t_val"]) and list(shape) == [4, 3, 10]
    assert set(nb[0].tolist()) <= {2, 3, 4, 5} and set(nb[2].tolist()) <= {3, 4, 5}
    assert set(nb[3].tolist()) <= {3, 5}
    check_layer_pack(OracleBackend(O, O.OracleGraph(random_csr)), L, "rg_", 3)


def test_layerwise_weight_func_goldens(O, fixture_csr, random_csr):
    """sampleLNB with a weight function (API_LOCAL_SAMPLE_L): the C restatement -
    libstdc++'s std::hash<std::string>, prime bucket policy and list insertion
    restated in oracle/eo_umap.c - against the reference harness' vectors."""
    import os
    from conftest import GOLDEN
    from layer_cases import check_layer_func_pack
    L = np.load(os.path.join(GOLDEN, "layerwise.npz"))
    for prefix, csr in (("fx_", fixture_csr), ("rg_", random_csr)):
        check_layer_func_pack(O.OracleGraph(csr).sample_neighbor_layerwise_func, L, prefix)
    # known answers of the restated hash (std::hash<std::string> of libstdc++,
    # GLIBCXX_3.4.30): values produced by the real library
    assert O.std_hash("") == 6142509188972423790
    assert O.std_hash("a") == 4993892634952068459
    assert O.std_hash("euler") == 6969579379935283931
    assert O.std_hash("12345678901234567890") == 3825371124932007023
    assert O.umap_iteration_order([]).shape == (0,)
