"""GPU parity of the layerwise-sampling ops (sampleLNB without a weight
function) and SparseGetAdj, through the C ABI: the reference's golden vectors
(tests/golden/layerwise.npz), random graphs against the oracle, empty / ragged
inputs, the euler_ops surface and the dataflows built on it.  Bit-exact."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, make_random_graph
from layer_cases import GpuBackend, OracleBackend, check_layer_func_pack, check_layer_pack

pytestmark = pytest.mark.gpu


def gpu_graph(EA, csr, **kw):
    return EA.Graph.from_csr(csr.row_id, csr.row_ptr, csr.type_end, csr.nbr,
                             csr.prefix_w, csr.type_prefix, csr.n_types,
                             csr.node_type, csr.node_weight, **kw)


def t2n(t):
    return t.detach().cpu().numpy()


def test_layerwise_goldens_gpu(EA, O, torch_cuda, fixture_csr, random_csr):
    L = np.load(os.path.join(GOLDEN, "layerwise.npz"))
    G = gpu_graph(EA, fixture_csr)
    B = GpuBackend(torch_cuda, G)
    check_layer_pack(B, L, "fx_", 2)
    nb, ind, val, shape = B.sample_neighbor_layerwise(int(L["seed"]), 90, L["fx_t_nodes"],
                                                      [0, 1], 10, -1)
    assert np.array_equal(nb, L["fx_t_nb"]) and np.array_equal(ind, L["fx_t_ind"])
    assert np.array_equal(val, L["fx_t_val"]) and list(shape) == [4, 3, 10]
    # EdgeExist from the rows == the reference's Edge records, pair by pair
    have = {tuple(x) for x in L["fx_edges"].tolist()}
    ids = [int(x) for x in fixture_csr.row_id]
    for t in (0, 1, 2):
        src = np.repeat(ids, len(ids) + 2)
        dst = np.tile(ids + [0, 99], len(ids))
        idx, vals = B.sparse_get_adj(src, dst, len(src), 1, 1, [t])
        got = (idx[:, 1] - idx[:, 0]) == 1
        want = np.array([(s, d, t) in have for s, d in zip(src, dst)])
        assert np.array_equal(got, want), t
    check_layer_pack(GpuBackend(torch_cuda, gpu_graph(EA, random_csr)), L, "rg_", 3)


def test_layerwise_weight_func_goldens_gpu(EA, O, torch_cuda, fixture_csr, random_csr):
    """sampleLNB with a weight function (API_GET_NB_NODE -> API_LOCAL_SAMPLE_L ->
    adjacency) == the reference harness: candidate order of the op's
    std::unordered_map, accumulated / sqrt'ed weights, the memset fill."""
    L = np.load(os.path.join(GOLDEN, "layerwise.npz"))
    for prefix, csr in (("fx_", fixture_csr), ("rg_", random_csr)):
        B = GpuBackend(torch_cuda, gpu_graph(EA, csr))
        check_layer_func_pack(B.sample_neighbor_layerwise_func, L, prefix)


@pytest.fixture(scope="module")
def lw_pair(EA, O):
    rng = np.random.default_rng(4711)
    ids, seg, nbr, w, nt, nw = make_random_graph(rng, 20000, 4, max_deg=40,
                                                 id_space=10 ** 12)
    # one hub row: a long sequential weight sum and a long EdgeExist scan
    csr = O.csr_from_raw(ids, seg, nbr, w, 4, nt, nw)
    return gpu_graph(EA, csr), O.OracleGraph(csr), ids, rng


@pytest.mark.parametrize("et", [[0], [3], [1, 2], [0, 1, 2, 3], [], [9], [1, 9]])
def test_layerwise_primitives_vs_oracle(EA, O, torch_cuda, lw_pair, et):
    G, OG, ids, rng = lw_pair
    B, OB = GpuBackend(torch_cuda, G), OracleBackend(O, OG)
    q = np.concatenate([rng.choice(ids, 70000), [0, 2 ** 63 + 5]]).astype(np.uint64)
    a, b = B.get_edge_sum_weight(q, et), OB.get_edge_sum_weight(q, et)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    for call, dn in ((0, -1), (6, 31337)):
        for x, y in zip(B.sample_layer(17, call, q, et, dn),
                        OB.sample_layer(17, call, q, et, dn)):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("host_rows", [1, 0, 2],
                         ids=["root_auto", "root_device", "root_host"])
def test_sample_root_vs_oracle(EA, O, torch_cuda, lw_pair, host_rows):
    """Both builders of the alias tables (one lane per batch row on the device /
    the host's cores for calls with few rows, tuning key 15) give the
    reference's draws."""
    from euler_amd import _lib
    _lib.lib().euler_gpu_set_tuning(15, host_rows)
    try:
        _sample_root_cases(EA, O, torch_cuda, lw_pair)
    finally:
        _lib.lib().euler_gpu_set_tuning(15, 1)


def _sample_root_cases(EA, O, torch_cuda, lw_pair):
    G, OG, ids, rng = lw_pair
    B, OB = GpuBackend(torch_cuda, G), OracleBackend(O, OG)
    for batch, n, m in ((1, 1, 1), (7, 2, 5), (300, 25, 10), (1000, 3, 64),
                        (5, 1000, 200), (64, 257, 33)):
        roots = rng.choice(ids, (batch, n)).astype(np.uint64)
        w = (rng.random((batch, n)) * 5).astype(np.float32)
        w[rng.random((batch, n)) < 0.3] = 0
        w[0] = 0
        if batch > 2:
            w[1] = 1.0
            w[2] = 0
            w[2, n - 1] = 1e-30
        for call, dn in ((0, -1), (3, 77)):
            assert np.array_equal(B.sample_root(11, call, roots, w, n, m, dn),
                                  OB.sample_root(11, call, roots, w, n, m, dn)), (n, m)


@pytest.mark.parametrize("adj_scan,long_row", [(0, 16384), (0, 20), (1, 16384)],
                         ids=["adj_hash", "adj_hash_split", "adj_scan"])
def test_sparse_get_adj_and_layerwise_vs_oracle(EA, O, torch_cuda, lw_pair, adj_scan,
                                                long_row):
    """The mask builders: LDS hash table of the candidates (rows streamed by one
    wave / rows above tuning key 17 split over workgroups) and the direct scan
    (key 16), incl. more candidates than one table chunk holds."""
    from euler_amd import _lib
    _lib.lib().euler_gpu_set_tuning(16, adj_scan)
    _lib.lib().euler_gpu_set_tuning(17, long_row)
    try:
        _adj_cases(EA, O, torch_cuda, lw_pair)
    finally:
        _lib.lib().euler_gpu_set_tuning(16, 0)
        _lib.lib().euler_gpu_set_tuning(17, 16384)


def _adj_cases(EA, O, torch_cuda, lw_pair):
    torch = torch_cuda
    G, OG, ids, rng = lw_pair
    B, OB = GpuBackend(torch, G), OracleBackend(O, OG)
    for batch, n, m in ((1, 5, 7), (3, 4, 70), (2, 1, 1), (4, 9, 130), (40, 25, 10),
                        (1, 300, 300), (2, 37, 4500), (1, 6, 2048), (1, 3, 2049)):
        nodes = rng.choice(ids, (batch, n)).astype(np.uint64)
        if n > 1:
            nodes[0, 1] = nodes[0, 0]
        nodes[-1, -1] = 2 ** 62 + 3
        cand = rng.choice(ids, (batch, m)).astype(np.uint64)
        for b in range(batch):
            nb = OG.get_full_neighbor(nodes[b], [0, 1, 2, 3])[1]
            if len(nb):
                take = rng.choice(nb, m // 2 + 1)
                cand[b, :len(take)] = take[:m]
        cand[0, m - 1] = 2 ** 64 - 1           # default_node = -1 among the candidates
        cand[-1, 0] = cand[-1, m // 2]           # duplicate candidates
        for et in ([0], [1, 3], [0, 1, 2, 3], [], [9]):
            x = B.sparse_get_adj(nodes, cand, batch, n, m, et)
            y = OB.sparse_get_adj(nodes, cand, batch, n, m, et)
            assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
            ind, val, shape = G.sparse_get_adj(
                torch.as_tensor(nodes.view(np.int64)).cuda(),
                torch.as_tensor(cand.view(np.int64)).cuda(), et, n, m)
            wi, wv, ws = OG._adj_to_sparse(nodes, cand, batch, n, m, *y)
            assert np.array_equal(t2n(ind), wi) and np.array_equal(t2n(val), wv)
            assert list(shape) == list(ws)
    for batch, n, count in ((4, 3, 10), (2, 16, 5), (6, 1, 4), (128, 25, 10), (1, 500, 64)):
        nodes = rng.choice(ids, (batch, n)).astype(np.uint64)
        nodes[-1, -1] = 0
        for et in ([0], [0, 1], [0, 1, 2, 3]):
            for call, dn in ((0, -1), (9, 31337)):
                x = B.sample_neighbor_layerwise(21, call, nodes, et, count, dn)
                y = OB.sample_neighbor_layerwise(21, call, nodes, et, count, dn)
                for u, v in zip(x, y):
                    assert np.array_equal(u, v)


def test_layerwise_empty_and_errors(EA, O, torch_cuda, fixture_csr):
    torch = torch_cuda
    G = gpu_graph(EA, fixture_csr)
    e = torch.zeros(0, dtype=torch.int64, device="cuda")
    assert G.get_edge_sum_weight(e, [0]).numel() == 0
    assert G.sample_layer(e, [0])[0].numel() == 0
    idx, vals = G.sparse_get_adj_core(e, e, 3, 4, [0])
    assert idx.shape == (0, 2) and vals.numel() == 0
    ind, val, shape = G.sparse_get_adj(e, e, [0], 3, 4)
    assert ind.shape == (0, 3) and shape == [0, 0, 0]
    nodes = torch.as_tensor(fixture_csr.row_id.astype(np.int64)).cuda()
    # no candidates: every root has an empty slice, the TF form has no entries
    idx, vals = G.sparse_get_adj_core(nodes, e, 1, 0, [0])
    assert t2n(idx).sum() == 0 and vals.numel() == 0
    nb, (ind, val, shape) = G.sample_neighbor_layerwise(nodes.reshape(1, -1), [0, 1], 0)
    assert nb.shape == (1, 0) and ind.shape[0] == 0
    with pytest.raises(ValueError):
        G.sample_neighbor_layerwise(nodes, [0], 4)
    from euler_amd._lib import EulerGpuError
    with pytest.raises(EulerGpuError):
        G.get_edge_sum_weight(nodes, list(range(40)))


def test_layerwise_ops_surface_and_dataflows(EA, O, torch_cuda, lw_pair):
    """euler_ops.sample_fanout_layerwise(_each_node) / get_multi_hop_neighbor
    and Layerwise / Whole dataflows == the same compositions on the oracle."""
    torch = torch_cuda
    from euler_amd import euler_ops as ops
    from euler_amd.euler_ops import base
    G, OG, ids, rng = lw_pair
    OB = OracleBackend(O, OG)
    base.set_default_graph(G)
    roots = rng.choice(ids, 32).astype(np.int64)
    rt = torch.as_tensor(roots).cuda()

    # sample_fanout_layerwise: one batch row per hop (call ids 700, 701)
    G.set_seed(5, call_id=700)
    nl, al = ops.sample_fanout_layerwise(rt, [[0, 1], [2, 3]], [20, 10], default_node=-1)
    cur, last = roots.reshape(1, -1), len(roots)
    for h, (et, c) in enumerate((([0, 1], 20), ([2, 3], 10))):
        nb, ind, val, shape = OB.sample_neighbor_layerwise(5, 700 + h,
                                                           cur.view(np.uint64), et, c, -1)
        assert np.array_equal(t2n(nl[h + 1]), nb.reshape(-1))
        assert np.array_equal(t2n(al[h][0]), ind) and np.array_equal(t2n(al[h][1]), val)
        assert list(al[h][2]) == list(shape)
        cur = nb.reshape(1, -1)

    # sample_fanout_layerwise_each_node: hop 1 sample_neighbor (call 710), hop 2
    # layerwise over each root's own 6 neighbours (call 711)
    G.set_seed(5, call_id=710)
    nl, al = ops.sample_fanout_layerwise_each_node(rt, [[0], [0, 1, 2, 3]], [6, 4])
    n1 = OG.sample_neighbor(5, 710, roots, [0], 6, -1)[0]
    assert np.array_equal(t2n(nl[1]), n1.reshape(-1))
    nb, ind, val, shape = OB.sample_neighbor_layerwise(
        5, 711, n1.reshape(-1, 6).view(np.uint64), [0, 1, 2, 3], 4, -1)
    assert np.array_equal(t2n(nl[2]), nb.reshape(-1)) and np.array_equal(t2n(al[0][0]), ind)

    # get_multi_hop_neighbor: distinct full neighbours + weighted adjacency
    nodes_list, adj_list = ops.get_multi_hop_neighbor(rt, [[0, 1], [3]])
    cur = roots.astype(np.uint64)
    for h, et in enumerate(([0, 1], [3])):
        idx, fid, fw, _ = OG.get_full_neighbor(cur, et)
        rows = np.repeat(np.arange(len(cur)), idx[:, 1] - idx[:, 0])
        uq, gi = O.id_unique(fid)
        order = np.argsort(rows * max(len(uq), 1) + gi.astype(np.int64), kind="stable")
        assert np.array_equal(t2n(nodes_list[h + 1]).view(np.uint64), uq)
        gi_, gv_, gs_ = adj_list[h]
        assert np.array_equal(t2n(gi_), np.stack([rows[order], gi.astype(np.int64)[order]], 1))
        assert np.array_equal(t2n(gv_), fw[order]) and gs_ == [len(cur), len(uq)]
        cur = uq

    # WholeDataFlow: sparse_get_adj(n_id, n_id) read as the reference reads it
    flow = EA.dataflow.WholeDataFlow(G, [[0, 1], [0, 1]], add_self_loops=True)
    df = flow(rt)
    idx, vals = OB.sparse_get_adj(roots.view(np.uint64), roots.view(np.uint64), 1,
                                  len(roots), len(roots), [0, 1])
    wi, wv, ws = OG._adj_to_sparse(roots.view(np.uint64), roots.view(np.uint64), 1,
                                   len(roots), len(roots), idx, vals)
    inv = np.arange(len(roots))
    want = np.stack([np.concatenate([wi[:, 0], inv]), np.concatenate([wi[:, 1], inv])])
    assert len(df) == 2
    for blk in df:
        assert np.array_equal(t2n(blk.edge_index), want)
        assert np.array_equal(t2n(blk.n_id), roots) and blk.size == [32, 32]

    # LayerwiseDataFlow: hop 1 layerwise over the batch (call 720), last hop full
    G.set_seed(5, call_id=720)
    flow = EA.dataflow.LayerwiseDataFlow(G, [8, 8], [[0, 1, 2, 3], [0]],
                                         add_self_loops=False)
    df = flow(rt)

    def uniq(a):
        uq, gi = O.id_unique(np.asarray(a).astype(np.uint64))
        return uq.astype(np.int64), gi.astype(np.int64)

    nb, ind, val, shape = OB.sample_neighbor_layerwise(
        5, 720, roots.reshape(1, -1).view(np.uint64), [0, 1, 2, 3], 8, -1)
    nbrs = [nb.reshape(-1)[ind[:, 2]]]
    srcs = [ind[:, 1]]
    n_id, _ = uniq(np.concatenate([nbrs[0], roots]))
    idx, fid, _, _ = OG.get_full_neighbor(n_id.astype(np.uint64), [0])
    nbrs.append(fid.astype(np.int64))
    srcs.append(np.repeat(np.arange(len(n_id)), idx[:, 1] - idx[:, 0]))
    n_id = roots.copy()
    for i, blk in enumerate(df.blocks):
        new_n_id, inv = uniq(np.concatenate([nbrs[i], n_id]))
        res = inv[-len(n_id):]
        inv = inv[:-len(n_id)]
        assert np.array_equal(t2n(blk.n_id), new_n_id)
        assert np.array_equal(t2n(blk.res_n_id), res)
        assert np.array_equal(t2n(blk.edge_index), np.stack([srcs[i], inv]))
        n_id = new_n_id


def test_layerwise_plugin_dag(EA, O, torch_cuda, lw_pair):
    """The six registered op kernels executed as the reference's translator
    chains them (euler_op_run_sample_lnb) == the oracle's op-by-op chain."""
    import ctypes as C
    from euler_amd import _lib
    L = _lib.lib()
    G, OG, ids, rng = lw_pair
    for batch, n, m in ((3, 4, 6), (1, 20, 50), (16, 1, 3)):
        nodes = rng.choice(ids, (batch, n)).astype(np.uint64)
        nodes[0, 0] = 12345678901234            # unknown node: weight 0
        for et in ([0], [1, 2, 3]):
            eta = np.asarray(et, np.int32)
            cap = batch * n * m
            adj_idx = np.zeros((batch * n, 2), np.int32)
            adj_id = np.zeros(cap, np.uint64)
            l_nb = np.zeros(batch * m, np.uint64)
            flat = np.ascontiguousarray(nodes.reshape(-1))
            got = L.euler_op_run_sample_lnb(
                G._h, 31, 7, flat.ctypes.data_as(_lib.u64p), batch, n,
                eta.ctypes.data_as(_lib.i32p), len(eta), m, b"", -1, cap,
                adj_idx.ctypes.data_as(_lib.i32p), adj_id.ctypes.data_as(_lib.u64p),
                l_nb.ctypes.data_as(_lib.u64p))
            assert got >= 0, got
            w = OG.get_edge_sum_weight(flat, et)
            l_root = O.sample_root(31, 7, flat, w, n, m, -1)
            want_nb = OG.sample_layer(31, 8, l_root, et, -1)[0]
            widx, wvals = OG.sparse_get_adj(flat, want_nb, batch, n, m, et)
            assert np.array_equal(l_nb, want_nb)
            assert np.array_equal(adj_idx, widx) and got == len(wvals)
            assert np.array_equal(adj_id[:got], wvals)
            # the weight-function form of the DAG == the direct C-ABI composition
            got = L.euler_op_run_sample_lnb(
                G._h, 31, 7, flat.ctypes.data_as(_lib.u64p), batch, n,
                eta.ctypes.data_as(_lib.i32p), len(eta), m, b"sqrt", -1, cap,
                adj_idx.ctypes.data_as(_lib.i32p), adj_id.ctypes.data_as(_lib.u64p),
                l_nb.ctypes.data_as(_lib.u64p))
            assert got >= 0, got
            G.set_seed(31)
            import torch
            nb, _adj = G.sample_neighbor_layerwise(
                torch.as_tensor(nodes.view(np.int64)).cuda(), et, m, -1, "sqrt", call_id=7)
            assert np.array_equal(l_nb.view(np.int64), t2n(nb).reshape(-1))
            widx, wvals = OG.sparse_get_adj(flat, l_nb, batch, n, m, et)
            assert np.array_equal(adj_idx, widx) and np.array_equal(adj_id[:got], wvals)


def test_node_type_and_sample_node_with_src(EA, O, torch_cuda, random_csr):
    """get_node_type / sample_n_with_types (sample_node_with_src) == oracle."""
    torch = torch_cuda
    from euler_amd._lib import EulerGpuError
    from euler_amd import euler_ops as ops
    from euler_amd.euler_ops import base
    G = gpu_graph(EA, random_csr)
    OG = O.OracleGraph(random_csr)
    OG.build_node_sampler()
    rng = np.random.default_rng(3)
    q = np.concatenate([rng.choice(random_csr.row_id, 5000), [0, 2 ** 63 + 5]]).astype(np.uint64)
    qt = torch.as_tensor(q.view(np.int64)).cuda()
    assert np.array_equal(t2n(G.get_node_type(qt)), OG.get_node_type(q))
    types = np.concatenate([OG.get_node_type(q[:5000]), [-1, -1, 0, 1]]).astype(np.int32)
    for call, count in ((0, 1), (1, 5), (2, 16)):
        G.set_seed(13)
        got = G.sample_n_with_types(count, types, call_id=call)
        want = OG.sample_n_with_types(13, call, types, count)
        assert np.array_equal(t2n(got).view(np.uint64), want)
    base.set_default_graph(G)
    G.set_seed(13, call_id=40)
    got = ops.sample_node_with_src(qt[:100], 7)
    want = OG.sample_n_with_types(13, 40, OG.get_node_type(q[:100]), 7)
    assert np.array_equal(t2n(got).view(np.uint64), want)
    with pytest.raises(EulerGpuError):          # the TF kernel aborts here
        G.sample_n_with_types(3, [0, 7])
    with pytest.raises(EulerGpuError):          # unknown src node: type INT32_MIN
        ops.sample_node_with_src(qt[-1:], 2)
    assert G.sample_n_with_types(0, [0]).shape == (1, 0)


def test_long_rows(EA, O, torch_cuda):
    """Rows longer than the lane-per-node limit: the wave-cooperative weight sum
    (chunks of 64 / groups of 256 edges, every boundary) and the row streaming
    of the adjacency kernel."""
    torch = torch_cuda
    rng = np.random.default_rng(99)
    degs = [0, 1, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 320, 511, 512,
            513, 1000, 3000, 8191, 8193, 20000]
    n, T = 64, 2
    ids = np.arange(10, 10 + n, dtype=np.uint64) * 7
    deg = np.zeros((n, T), np.int64)
    for i, d in enumerate(degs):
        deg[i, 0] = d
        deg[i, 1] = degs[-1 - i]
        deg[i + len(degs), 1] = d          # type 1 only
    seg = np.zeros(n * T + 1, np.int64)
    seg[1:] = np.cumsum(deg.reshape(-1))
    E = int(seg[-1])
    nbr = rng.choice(ids, E).astype(np.uint64)
    w = (rng.random(E) * 7.5 + 0.01).astype(np.float32)
    csr = O.csr_from_raw(ids, seg, nbr, w, T)
    G, OG = gpu_graph(EA, csr), O.OracleGraph(csr)
    B, OB = GpuBackend(torch, G), OracleBackend(O, OG)
    q = np.concatenate([ids, ids[::-1], [5]]).astype(np.uint64)
    from euler_amd import _lib
    for et in ([0], [1], [0, 1], [1, 0], [1, 1]):
        b = OB.get_edge_sum_weight(q, et)
        for scalar in (1, 0):        # both long-row chains (tuning key 18)
            _lib.lib().euler_gpu_set_tuning(18, scalar)
            try:
                a = B.get_edge_sum_weight(q, et)
            finally:
                _lib.lib().euler_gpu_set_tuning(18, 0)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (et, scalar)
        y = OB.sparse_get_adj(ids, ids, 1, n, n, et)
        # rows of one wave / rows split into segments of 8192 edges over workgroups
        for long_row in (16384, 1000, 50):
            _lib.lib().euler_gpu_set_tuning(17, long_row)
            try:
                x = B.sparse_get_adj(ids, ids, 1, n, n, et)
            finally:
                _lib.lib().euler_gpu_set_tuning(17, 16384)
            assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]), (et, long_row)


def test_sparse_feature_vs_goldens_and_loader(EA, O, torch_cuda, fixture_csr, random_csr):
    """get_sparse_feature on device == the reference's SparseTensor triples
    (sparse_features.npz), from a CSR upload and from the .dat loader."""
    torch = torch_cuda
    from conftest import ROOT
    sg = np.load(os.path.join(GOLDEN, "sparse_features.npz"))
    for prefix, csr in (("fx_", fixture_csr), ("rg_", random_csr)):
        feats = (int(sg[prefix + "n_u64"]), sg[prefix + "feat_ptr"], sg[prefix + "feat_idx"],
                 sg[prefix + "feat_val"])
        graphs = [gpu_graph(EA, csr, sparse_features=feats)]
        if prefix == "fx_":
            graphs.append(EA.Graph.load(os.path.join(GOLDEN, "fixture_dat")))
        q = torch.as_tensor(sg[prefix + "query"].view(np.int64)).cuda()
        for G in graphs:
            assert G.num_u64_features() == feats[0]
            got = G.get_sparse_feature(q, sg[prefix + "fids"].tolist(),
                                       sg[prefix + "defaults"].tolist())
            for k, (ind, val, shape) in enumerate(got):
                assert np.array_equal(t2n(ind), sg[prefix + "sp_%d_ind" % k]), (prefix, k)
                assert np.array_equal(t2n(val), sg[prefix + "sp_%d_val" % k]), (prefix, k)
                assert list(shape) == sg[prefix + "sp_%d_shape" % k].tolist(), (prefix, k)
    # a graph without uint64 features: every node yields the default entry
    G = gpu_graph(EA, fixture_csr)
    ind, val, shape = G.get_sparse_feature(q[:4], [0], [9])[0]
    assert t2n(val).tolist() == [9, 9, 9, 9] and shape == [4, 1]
    e = torch.zeros(0, dtype=torch.int64, device="cuda")
    ind, val, shape = G.get_sparse_feature(e, [0])[0]
    assert ind.shape == (0, 2) and shape == [0, 0]
    # the ops surface
    from euler_amd import euler_ops as ops
    from euler_amd.euler_ops import base
    base.set_default_graph(graphs[0])
    got = ops.get_sparse_feature(q, ["0", 1])
    assert np.array_equal(t2n(got[0][1]), sg["rg_sp_0_val"])


def test_cpp_host_example(EA, O, torch_cuda, fixture_csr):
    """examples/cpp/sage_minibatch: a C++ host driving the C ABI with nothing but
    the HIP runtime (InitQueryProxy -> SampleNode -> SampleFanout ->
    GetDenseFeature -> MPScatterAdd) prints the same bits as the Python surface,
    and the sampled block matches the oracle."""
    import subprocess
    from conftest import ROOT
    torch = torch_cuda
    exe = os.path.join(ROOT, "examples", "cpp", "sage_minibatch")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(exe)])
    dat = os.path.join(GOLDEN, "fixture_dat")
    seed, batch, c1, c2, fid, dim = 42, 8, 3, 2, 0, 2
    out = subprocess.run([exe, dat, str(seed), str(batch), str(c1), str(c2), str(fid), str(dim)],
                         check=True, capture_output=True, text=True, timeout=120).stdout
    got = {}
    for line in out.strip().splitlines():
        name, *vals = line.split()
        got[name] = vals
    ids = lambda k: np.array([int(v) for v in got[k]], np.uint64)
    bits = lambda k: np.array([int(v, 16) for v in got[k]], np.uint32)
    G = EA.Graph.load(dat)
    G.set_seed(seed)
    roots = G.sample_node(batch, -1, call_id=0)
    nb, w, t = G.sample_fanout(roots, [[0, 1], [0, 1]], [c1, c2], -1, call_id=1)
    feat = G.get_dense_feature(nb[2], [fid], [dim])[0]
    parent = torch.arange(batch * c1, device="cuda", dtype=torch.int32).repeat_interleave(c2)
    agg = EA.ops.scatter_add(feat, parent, batch * c1)
    assert np.array_equal(ids("roots"), t2n(roots).view(np.uint64))
    assert np.array_equal(ids("hop1"), t2n(nb[1]).view(np.uint64))
    assert np.array_equal(ids("hop2"), t2n(nb[2]).view(np.uint64))
    assert np.array_equal(bits("w2"), t2n(w[1]).reshape(-1).view(np.uint32))
    assert np.array_equal(bits("agg"), t2n(agg).reshape(-1).view(np.uint32))
    # ... and against the oracle, given the roots
    OG = O.OracleGraph(fixture_csr)
    on, ow, ot = OG.sample_fanout(seed, 1, ids("roots").view(np.int64), [[0, 1], [0, 1]],
                                  [c1, c2], -1)
    assert np.array_equal(ids("hop1").view(np.int64), on[0].reshape(-1))
    assert np.array_equal(ids("hop2").view(np.int64), on[1].reshape(-1))
    assert np.array_equal(bits("w2"), ow[1].reshape(-1).view(np.uint32))


def test_python_example_runs(torch_cuda, capsys, monkeypatch):
    """examples/python/graphsage_minibatch.py: the tf_euler-surface input pipeline (SageDataFlow ->
    get_dense_feature -> mean aggregation per block) on the reference's fixture directory and on a
    synthetic graph (run in this process: a fresh interpreter per run costs a torch import)."""
    import runpy
    import sys
    from conftest import ROOT
    exe = os.path.join(ROOT, "examples", "python", "graphsage_minibatch.py")
    for extra in (["--data", os.path.join(GOLDEN, "fixture_dat"), "--dim", "2", "--batch", "4"],
                  ["--nodes", "100000", "--dim", "16", "--batch", "256"]):
        monkeypatch.setattr(sys, "argv", [exe, "--steps", "2"] + extra)
        runpy.run_path(exe, run_name="__main__")
        out = capsys.readouterr().out
        assert "ms per training-step input" in out, out

def test_python_deepwalk_example(EA, O, torch_cuda, capsys, monkeypatch):
    """examples/python/deepwalk_minibatch.py: BaseNode2Vec.to_sample of the reference
    (examples/deepwalk/deepwalk.py:47-63) on the tf_euler surface - random_walk -> gen_pair ->
    sample_node(batch * pairs * num_negs) - runs on the fixture directory and on a synthetic graph,
    and its three outputs equal the same composition on the oracle."""
    import runpy
    import sys
    import torch
    from conftest import ROOT
    exe = os.path.join(ROOT, "examples", "python", "deepwalk_minibatch.py")
    for extra in (["--data", os.path.join(GOLDEN, "fixture_dat"), "--batch", "4"],
                  ["--nodes", "100000", "--batch", "256", "--walk-len", "5"]):
        monkeypatch.setattr(sys, "argv", [exe, "--steps", "2"] + extra)
        mod = runpy.run_path(exe, run_name="__main__")
        out = capsys.readouterr().out
        assert "ms per training-step input" in out, out
    # the composition against the oracle
    N = 30000
    p = EA.synth_params(5, N, 8 * N, weighted=True)
    G = EA.Graph.synthetic(p)
    rng = np.random.default_rng(3)
    w = (0.25 + rng.random(N)).astype(np.float32)
    ty = rng.integers(0, 3, N).astype(np.int32)
    G.set_node_sampler(None, ty, w, 3)
    G.set_seed(77, 10)
    EA.euler_ops.set_default_graph(G)
    po = O.SynthParams()
    for f, _ in po._fields_:
        setattr(po, f, getattr(p, f))
    csr = O.synth_csr(po)
    csr.node_type, csr.node_weight = ty, w
    OG = O.OracleGraph(csr)
    OG.build_node_sampler()
    inputs = rng.integers(1, N + 1, 300).astype(np.int64)
    for (wl, pp, qq, lw, rw, negs, nt) in ((3, 1.0, 1.0, 1, 1, 5, 1), (6, 0.5, 2.0, 2, 1, 3, -1)):
        G.set_seed(77, 10)
        src, pos, ng = mod["to_sample"](torch.as_tensor(inputs).cuda(), nt, [0], N, wl, pp, qq, lw, rw, negs)
        path = OG.random_walk(77, 10, inputs, [[0]] * wl, wl, pp, qq, N + 1)
        pair = O.gen_pair(path, lw, rw)
        want_neg = OG.sample_node(77, 10 + wl, [nt], pair.shape[0] * pair.shape[1] * negs)
        assert np.array_equal(src.cpu().numpy().reshape(-1), pair[..., 0].reshape(-1))
        assert np.array_equal(pos.cpu().numpy().reshape(-1), pair[..., 1].reshape(-1))
        assert np.array_equal(ng.cpu().numpy().reshape(-1).view(np.uint64), want_neg)


def test_layerwise_weight_func_vs_oracle(EA, O, torch_cuda, lw_pair):
    """sampleLNB with a weight function on a 20 000-node graph: the library (host
    tables in the real std::unordered_map, draws on the device) == the C oracle
    (libstdc++'s container order restated, oracle/eo_umap.c), incl. batches
    large enough for the multi-threaded host build."""
    G, OG, ids, rng = lw_pair
    B = GpuBackend(torch_cuda, G)
    for batch, n, count in ((8, 25, 10), (1, 2000, 256), (64, 60, 8), (3, 1, 4)):
        nodes = rng.choice(ids, (batch, n)).astype(np.uint64)
        nodes[-1, -1] = 2 ** 62 + 9
        for et in ([0], [1, 3], [0, 1, 2, 3]):
            for wf, dn in (("sqrt", -1), ("sqrt", 261), ("other", 0)):
                x = B.sample_neighbor_layerwise_func(23, 4, nodes, et, count, wf, dn)
                y = OG.sample_neighbor_layerwise_func(23, 4, nodes, et, count, wf, dn)
                for u, v in zip(x, y):
                    u, v = np.asarray(u), np.asarray(v)
                    if u.dtype == np.float32:
                        u, v = u.view(np.uint32), v.view(np.uint32)
                    assert np.array_equal(u, v), (batch, n, et, wf)
