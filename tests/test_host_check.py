"""The per-item functions of euler_amd/csrc/layer_fns.h - the source the HIP
kernels call one item per lane - compiled for the HOST by tests/csrc/
host_check.hip and compared with the oracle / the reference's golden vectors.
CPU only: a check of the kernel logic where no GPU exists (the lane mapping is
covered by the -m gpu tests).  Skipped when hipcc is absent."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, make_random_graph

HERE = os.path.dirname(os.path.abspath(__file__))
u64p, i64p, i32p = C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)
f32p, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint8)


def _p(a, t):
    return a.ctypes.data_as(t)


@pytest.fixture(scope="module")
def HC():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out_dir = os.path.join(HERE, "csrc", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libhost_check.so")
    src = os.path.join(HERE, "csrc", "host_check.hip")
    deps = [src] + [os.path.join(ROOT, "euler_amd", "csrc", f)
                    for f in ("layer_fns.h", "device_fns.h", "common.h", "philox.h",
                              "local_layer_host.h", "wb_index.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so)
                                     for d in deps):
        subprocess.check_call(
            [hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared",
             "-ffp-contract=off", "-I" + os.path.join(ROOT, "euler_amd", "csrc"),
             "-I" + os.path.join(ROOT, "include"), src, "-o", so])
    L = C.CDLL(so)
    L.hc_graph_create.restype = C.c_void_p
    L.hc_graph_create.argtypes = [C.c_int64, C.c_int32, u64p, i64p, i32p, u64p, f32p,
                                  f32p, C.c_int32]
    L.hc_graph_destroy.argtypes = [C.c_void_p]
    L.hc_wb_build.restype = C.c_int64
    L.hc_wb_build.argtypes = [C.c_void_p]
    L.hc_wb_overflows.restype = C.c_int64
    L.hc_wb_overflows.argtypes = [C.c_void_p]
    L.hc_wb_sample.argtypes = [C.c_void_p, i64p, C.POINTER(C.c_double), C.c_int64, i64p, f32p, u64p, i32p]
    L.hc_edge_sum_weight.argtypes = [C.c_void_p, u64p, C.c_int64, i32p, C.c_int32, f32p]
    L.hc_sample_root.argtypes = [C.c_uint64, C.c_uint32, u64p, f32p, C.c_int64,
                                 C.c_int32, C.c_int32, C.c_int64, u64p]
    L.hc_sample_layer.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, u64p, C.c_int64,
                                  i32p, C.c_int32, C.c_int64, u64p, f32p, i32p]
    L.hc_local_sample_layer.argtypes = [C.c_uint64, C.c_uint32, i32p, u64p, f32p, i32p,
                                        C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                        C.c_char_p, C.c_int64, u64p, f32p, i32p]
    L.hc_sample_neighbor_core.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, u64p, C.c_int64,
                                          i32p, C.c_int32, C.c_int32, u64p, f32p, i32p]
    L.hc_edge_exist_mask.argtypes = [C.c_void_p, u64p, u64p, C.c_int64, C.c_int32,
                                     C.c_int32, i32p, C.c_int32, u8p]
    return L


class HostBackend:
    """layer_cases backend over the host build of layer_fns.h.  The adjacency
    results are assembled from EdgeExistAny masks the way AdjCount/AdjFill do
    (hits in candidate order, the TF corner zero)."""

    def __init__(self, L, csr, force_hash=False):
        self.L, self.csr = L, csr
        self.keep = [np.ascontiguousarray(a) for a in (
            csr.row_id.astype(np.uint64), csr.row_ptr.astype(np.int64),
            csr.type_end.astype(np.int32), csr.nbr.astype(np.uint64),
            csr.prefix_w.astype(np.float32), csr.type_prefix.astype(np.float32))]
        k = self.keep
        self.h = L.hc_graph_create(len(k[0]), csr.n_types, _p(k[0], u64p), _p(k[1], i64p),
                                   _p(k[2], i32p), _p(k[3], u64p), _p(k[4], f32p),
                                   _p(k[5], f32p), 1 if force_hash else 0)

    def __del__(self):
        self.L.hc_graph_destroy(self.h)

    @staticmethod
    def _u64(a):
        a = np.ascontiguousarray(np.asarray(a).reshape(-1))
        return a.view(np.uint64) if a.dtype == np.int64 else a.astype(np.uint64)

    def get_edge_sum_weight(self, q, et):
        q = self._u64(q)
        et = np.ascontiguousarray(et, np.int32)
        out = np.zeros(len(q), np.float32)
        self.L.hc_edge_sum_weight(self.h, _p(q, u64p), len(q), _p(et, i32p), len(et),
                                  _p(out, f32p))
        return out

    def sample_layer(self, seed, call, q, et, dn):
        q = self._u64(q)
        et = np.ascontiguousarray(et, np.int32)
        oid = np.zeros(len(q), np.uint64)
        ow = np.zeros(len(q), np.float32)
        ot = np.zeros(len(q), np.int32)
        self.L.hc_sample_layer(self.h, seed, call, _p(q, u64p), len(q), _p(et, i32p),
                               len(et), dn, _p(oid, u64p), _p(ow, f32p), _p(ot, i32p))
        return oid, ow, ot

    def sample_root(self, seed, call, roots, w, n, m, dn):
        roots = self._u64(roots)
        w = np.ascontiguousarray(np.asarray(w, np.float32).reshape(-1))
        batch = len(roots) // n
        out = np.zeros(batch * m, np.uint64)
        self.L.hc_sample_root(seed, call, _p(roots, u64p), _p(w, f32p), batch, n, m, dn,
                              _p(out, u64p))
        return out

    def _mask(self, nodes, nb, batch, n, m, et):
        nodes, nb = self._u64(nodes), self._u64(nb)
        et = np.ascontiguousarray(et, np.int32)
        mask = np.zeros(batch * n * m, np.uint8)
        self.L.hc_edge_exist_mask(self.h, _p(nodes, u64p), _p(nb, u64p), batch, n, m,
                                  _p(et, i32p), len(et), _p(mask, u8p))
        return mask.reshape(batch * n, m).astype(bool), nb.reshape(batch, m)

    def sparse_get_adj(self, nodes, nb, batch, n, m, et):
        mask, nb = self._mask(nodes, nb, batch, n, m, et)
        counts = mask.sum(1)
        idx = np.zeros((batch * n, 2), np.int32)
        idx[:, 1] = np.cumsum(counts)
        idx[1:, 0] = idx[:-1, 1]
        vals = np.concatenate([nb[r // n][mask[r]] for r in range(batch * n)]
                              + [np.zeros(0, np.uint64)])
        return idx, vals

    def _sparse(self, nodes, l_nb, batch, n, count, et):
        mask, _ = self._mask(nodes, l_nb, batch, n, count, et)
        mask = mask.reshape(batch, n, count)
        emit = mask.copy()
        emit[:, n - 1, count - 1] = True
        return (np.argwhere(emit).astype(np.int64), mask[emit].astype(np.int64),
                np.array([batch, n, count], np.int64))

    def sample_neighbor_layerwise_func(self, OG, seed, call, nodes, et, count, wf, dn):
        """sampleLNB with a weight function: the library's host table builder +
        LocalLayerPick (full neighbours supplied by the oracle)."""
        nodes = np.asarray(nodes)
        batch, n = nodes.shape
        flat = self._u64(nodes)
        idx, ids, w, t = OG.get_full_neighbor(flat, et)
        idx = np.ascontiguousarray(idx.reshape(-1), np.int32)
        oid = np.zeros(batch * count, np.uint64)
        ow = np.zeros(batch * count, np.float32)
        ot = np.zeros(batch * count, np.int32)
        rc = self.L.hc_local_sample_layer(seed, call, _p(idx, i32p), _p(ids, u64p),
                                          _p(w, f32p), _p(t, i32p), len(ids), batch, n,
                                          count, wf.encode(), dn, _p(oid, u64p),
                                          _p(ow, f32p), _p(ot, i32p))
        assert rc == 0
        return (oid.view(np.int64).reshape(batch, count), ow.reshape(batch, count),
                ot.reshape(batch, count)) + self._sparse(nodes, oid, batch, n, count, et)

    def sample_neighbor_layerwise(self, seed, call, nodes, et, count, dn):
        nodes = np.asarray(nodes)
        batch, n = nodes.shape
        w = self.get_edge_sum_weight(nodes, et)
        l_root = self.sample_root(seed, call, nodes, w, n, count, dn)
        l_nb = self.sample_layer(seed, call, l_root, et, dn)[0]
        mask, _ = self._mask(nodes, l_nb, batch, n, count, et)
        mask = mask.reshape(batch, n, count)
        emit = mask.copy()
        emit[:, n - 1, count - 1] = True
        ind = np.argwhere(emit).astype(np.int64)
        val = mask[emit].astype(np.int64)
        return (l_nb.view(np.int64).reshape(batch, count), ind, val,
                np.array([batch, n, count], np.int64))


def test_local_sample_layer_host_vs_goldens(HC, O, fixture_csr, random_csr):
    """sampleLNB with a weight function: the library's host table builder (the
    real std::unordered_map<std::string, ...>) + LocalLayerPick == the reference
    harness (accumulated weights, sqrt, iteration order, memset fill)."""
    from layer_cases import check_layer_func_pack
    L = np.load(os.path.join(GOLDEN, "layerwise.npz"))
    for prefix, csr in (("fx_", fixture_csr), ("rg_", random_csr)):
        H, OG = HostBackend(HC, csr), O.OracleGraph(csr)
        check_layer_func_pack(
            lambda *a: H.sample_neighbor_layerwise_func(OG, *a), L, prefix)


def test_layer_fns_host_vs_goldens(HC, O, fixture_csr, random_csr):
    from layer_cases import check_layer_pack
    L = np.load(os.path.join(GOLDEN, "layerwise.npz"))
    check_layer_pack(HostBackend(HC, fixture_csr), L, "fx_", 2)
    check_layer_pack(HostBackend(HC, fixture_csr, force_hash=True), L, "fx_", 2)
    check_layer_pack(HostBackend(HC, random_csr), L, "rg_", 3)


@pytest.mark.parametrize("T,n_nodes,id_space", [(1, 400, "identity"), (4, 1500, None),
                                                (3, 300, 10 ** 12)])
def test_layer_fns_host_vs_oracle_random(HC, O, T, n_nodes, id_space):
    from layer_cases import OracleBackend
    rng = np.random.default_rng(100 + T)
    ids, seg, nbr, w, nt, nw = make_random_graph(
        rng, n_nodes, T, max_deg=25, id_space=None if id_space == "identity" else id_space)
    if id_space == "identity":        # contiguous ids: the identity id map
        ids = np.arange(1, n_nodes + 1, dtype=np.uint64)
        nbr = rng.choice(ids, len(nbr)).astype(np.uint64)
    csr = O.csr_from_raw(ids, seg, nbr, w, T, nt, nw)
    H, G = HostBackend(HC, csr), OracleBackend(O, O.OracleGraph(csr))
    q = np.concatenate([rng.choice(ids, 700), [0, 2 ** 63 + 9]]).astype(np.uint64)
    for et in ([0], [T - 1], list(range(T)), [], [T + 2], [0, T - 1, 0][:max(2, T - 1)]):
        a, b = H.get_edge_sum_weight(q, et), G.get_edge_sum_weight(q, et)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), et
        for call, dn in ((1, -1), (2, 99)):
            for x, y in zip(H.sample_layer(5, call, q, et, dn),
                            G.sample_layer(5, call, q, et, dn)):
                assert np.array_equal(x, y), et
    for n, m in ((1, 2), (2, 7), (5, 64), (33, 65), (257, 20), (1000, 3)):
        batch = 11
        roots = rng.choice(ids, (batch, n)).astype(np.uint64)
        wts = (rng.random((batch, n)) * 3).astype(np.float32)
        wts[rng.random((batch, n)) < 0.4] = 0
        wts[0] = 0
        wts[1] = 0.125
        if n > 1:
            wts[2] = 0; wts[2, n - 1] = 1e-30
        assert np.array_equal(H.sample_root(9, 4, roots, wts, n, m, 5),
                              G.sample_root(9, 4, roots, wts, n, m, 5)), (n, m)
    for batch, n, count in ((3, 4, 9), (1, 70, 130), (5, 1, 1)):
        nodes = rng.choice(ids, (batch, n)).astype(np.uint64)
        for et in ([0], list(range(T))):
            x = H.sample_neighbor_layerwise(3, 8, nodes, et, count, -1)
            y = G.sample_neighbor_layerwise(3, 8, nodes, et, count, -1)
            for u, v in zip(x, y):
                assert np.array_equal(u, v)
            nbq = y[0].reshape(-1).view(np.uint64)
            for u, v in zip(H.sparse_get_adj(nodes, nbq, batch, n, count, et),
                            G.sparse_get_adj(nodes, nbq, batch, n, count, et)):
                assert np.array_equal(u, v)


def test_local_sample_layer_threaded_vs_reference(HC, O):
    """Batches large enough for the multi-threaded host table build == the
    reference harness (one container per batch row there too)."""
    if not O.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(808)
    ids, seg, nbr, w, nt, nw = make_random_graph(rng, 2000, 3, max_deg=20)
    csr = O.csr_from_raw(ids, seg, nbr, w, 3, nt, nw)
    R = O.RefGraph.build_raw(ids, seg, nbr, w, 3, nt, nw)
    R.add_edges_from_adjacency()
    H, OG = HostBackend(HC, csr), O.OracleGraph(csr)
    for batch, n, count in ((64, 60, 8), (9, 400, 16)):
        nodes = rng.choice(ids, (batch, n)).astype(np.uint64)
        assert len(OG.get_full_neighbor(nodes.reshape(-1), [0, 1, 2])[1]) > (1 << 14)
        for wf, dn in (("sqrt", -1), ("id", 7)):
            a = R.sample_neighbor_layerwise_func(5, 3, nodes, [0, 1, 2], count, wf, dn)
            b = H.sample_neighbor_layerwise_func(OG, 5, 3, nodes, [0, 1, 2], count, wf, dn)
            for x, y in zip(a, b):
                if x.dtype == np.float32:
                    x, y = x.view(np.uint32), np.asarray(y, np.float32).view(np.uint32)
                assert np.array_equal(x, y)


def test_generic_sampler_logic_host_vs_goldens_and_oracle(HC, O, fixture_csr, random_csr,
                                                          fixture_samples, random_samples):
    """InitRowSampler + SampleAt (device_fns.h: the per-sample logic of the generic
    K1 kernel, every edge-type mode) compiled for the host == the reference
    sampler's golden vectors and the oracle on a random graph."""
    def run(H, seed, call, q, et, count):
        q = np.ascontiguousarray(q, np.uint64)
        et = np.ascontiguousarray(et, np.int32)
        oid = np.zeros(len(q) * count, np.uint64)
        ow = np.zeros(len(q) * count, np.float32)
        ot = np.zeros(len(q) * count, np.int32)
        HC.hc_sample_neighbor_core(H.h, seed, call, _p(q, u64p), len(q), _p(et, i32p), len(et),
                                   count, _p(oid, u64p), _p(ow, f32p), _p(ot, i32p))
        return oid, ow, ot

    for csr, s in ((fixture_csr, fixture_samples), (random_csr, random_samples)):
        for force_hash in (False, True):
            H = HostBackend(HC, csr, force_hash=force_hash)
            seed = int(s["seed"])
            q = s["query_ids"]
            n = 0
            while "nb_%d_1_et" % n in s.files:
                for count in (1, 5):
                    key = "nb_%d_%d_" % (n, count)
                    oid, ow, ot = run(H, seed, 11 + n, q, s[key + "et"], count)
                    assert np.array_equal(oid, s[key + "id"]), key
                    assert np.array_equal(ow.view(np.uint32), s[key + "w"].view(np.uint32)), key
                    assert np.array_equal(ot, s[key + "t"]), key
                n += 1
            assert n >= 7
    rng = np.random.default_rng(31)
    ids, seg, nbr, w, nt, nw = make_random_graph(rng, 2500, 4, max_deg=30, id_space=10 ** 12)
    csr = O.csr_from_raw(ids, seg, nbr, w, 4, nt, nw)
    H, OG = HostBackend(HC, csr), O.OracleGraph(csr)
    q = np.concatenate([rng.choice(ids, 1500), [0, 2 ** 63 + 5]]).astype(np.uint64)
    for et in ([0], [3], [1, 2], [3, 0, 1], [0, 1, 2, 3], [], [2, 2], [9], [1, 9]):
        for count in (1, 10):
            _, oid, ow, ot = OG.sample_neighbor_core(99, 5, q, et, count)
            hid, hw, ht = run(H, 99, 5, q, et, count)
            assert np.array_equal(hid, oid) and np.array_equal(ht, ot), et
            assert np.array_equal(hw.view(np.uint32), ow.view(np.uint32)), et


def _wb_case(L, O, degs, weight_fn, rng, draws_per_row=64, extra_u=()):
    """rows of the given degrees with weights from weight_fn(deg) -> f32 array; returns
    (n draws, n cold draws) after checking every draw against the oracle's RandomSelect."""
    segs, ws = [0], []
    for d in degs:
        w = np.asarray(weight_fn(d), np.float32)
        assert len(w) == d
        ws.append(w)
        segs.append(segs[-1] + d)
    n = len(degs)
    w_all = np.concatenate(ws) if ws else np.zeros(0, np.float32)
    nbr = rng.integers(1, 1 << 40, len(w_all)).astype(np.uint64)
    csr = O.csr_from_raw(np.arange(1, n + 1, dtype=np.uint64), np.asarray(segs, np.int64), nbr, w_all, 1)
    H = HostBackend(L, csr)
    blocks = L.hc_wb_build(H.h)
    assert blocks == sum((1 if 0 < d <= 10 else (d + 3) // 4) for d in degs if d > 0)
    _wb_case.last_overflow_frac = L.hc_wb_overflows(H.h) / max(1, blocks)
    rows = np.repeat(np.arange(n, dtype=np.int64), draws_per_row + len(extra_u))
    us = np.concatenate([np.concatenate([rng.random(draws_per_row), np.asarray(extra_u, np.float64)])
                         for _ in range(n)]) if n else np.zeros(0)
    m = np.zeros(len(rows), np.int64); w = np.zeros(len(rows), np.float32)
    ids = np.zeros(len(rows), np.uint64); cold = np.zeros(len(rows), np.int32)
    L.hc_wb_sample(H.h, _p(rows, i64p), us.ctypes.data_as(C.POINTER(C.c_double)), len(rows),
                   _p(m, i64p), _p(w, f32p), _p(ids, u64p), _p(cold, i32p))
    for i in range(len(rows)):
        r = int(rows[i]); d = degs[r]
        if d == 0:
            assert m[i] == -1
            continue
        b = int(csr.row_ptr[r])
        sw = csr.prefix_w[b:b + d]
        want = O.random_select(sw, 0, d - 1, float(us[i]))
        assert m[i] == want, (r, d, us[i], int(m[i]), want, int(cold[i]))
        assert ids[i] == csr.nbr[b + want]
        assert w[i] == np.float32(sw[want]) - (np.float32(sw[want - 1]) if want else np.float32(0))
    return len(rows), int(cold.sum())


def test_weight_bucket_index_vs_random_select(HC, O):
    """wb_index.h (the direct-address search of the one-kernel fanout): for every
    draw the index, id and weight RandomSelect gives (common/compact_weighted_collection.h:
    30-52, oracle/euler_oracle.c: eo_random_select) - on smooth rows through ONE block, on
    rows whose blocks cannot bracket the draw (heavy tails, zero weights, huge dynamic range,
    denormal totals, draws that round up to the total) through the cold path; and smooth
    rows must hardly ever go cold (that is the whole point of the layout)."""
    rng = np.random.default_rng(11)
    degs = [0, 1, 2, 3, 9, 10, 11, 12, 13, 14, 15, 16, 17, 20, 21, 39, 40, 41, 64, 65, 100, 257, 1000, 4099, 20011]
    edge_u = (0.0, 1.0 - 2.0 ** -53, 0.5, 2.0 ** -40)
    # i.i.d. uniform [0.5, 8): the metric graph's weights
    n, c = _wb_case(HC, O, degs, lambda d: 0.5 + 7.5 * rng.random(d), rng, extra_u=edge_u)
    assert c <= n * 0.002, (n, c)
    # ... and the builder's own statistic says so (<= 2 overflowing buckets in a thousand: the lean
    # kernels may use the index), while a heavy-tailed graph is told apart (they keep the levels)
    assert _wb_case.last_overflow_frac <= 0.002, _wb_case.last_overflow_frac
    # all equal (non-unit) weights; unit weights
    n, c = _wb_case(HC, O, degs, lambda d: np.full(d, 0.37), rng, extra_u=edge_u)
    assert c <= n * 0.002, (n, c)
    _wb_case(HC, O, degs, lambda d: np.ones(d), rng, extra_u=edge_u)
    # heavy tail (Pareto), zeros mixed in, one giant among dust, increasing and decreasing ramps
    _wb_case(HC, O, degs, lambda d: (rng.pareto(0.7, d) + 1e-3), rng, extra_u=edge_u)
    assert _wb_case.last_overflow_frac > 0.002, _wb_case.last_overflow_frac
    _wb_case(HC, O, degs, lambda d: np.where(rng.random(d) < 0.4, 0.0, rng.random(d)), rng, extra_u=edge_u)

    def giant(d):
        w = np.full(d, 1e-3)
        if d:
            w[int(rng.integers(0, d))] = 1e6
        return w
    _wb_case(HC, O, degs, giant, rng, extra_u=edge_u)
    _wb_case(HC, O, degs, lambda d: np.arange(1, d + 1, dtype=np.float64), rng, extra_u=edge_u)
    _wb_case(HC, O, degs, lambda d: np.arange(d, 0, -1, dtype=np.float64) ** 2, rng, extra_u=edge_u)
    # all-zero rows (every draw is cold: total = 0), denormal totals, totals near f32 max
    _wb_case(HC, O, [1, 5, 30, 200], lambda d: np.zeros(d), rng, draws_per_row=8, extra_u=edge_u)
    _wb_case(HC, O, [1, 5, 30, 200], lambda d: np.full(d, 1e-42), rng, draws_per_row=8, extra_u=edge_u)
    _wb_case(HC, O, [1, 5, 30, 200], lambda d: np.full(d, 1e36), rng, draws_per_row=8, extra_u=edge_u)
