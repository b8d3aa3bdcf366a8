"""The layerwise golden battery (tests/golden/layerwise.npz, written by
make_golden.py from oracle/_ref) replayed through any backend that offers the
five entry points below - the C restatement on the CPU, the HIP library on the
GPU.  Everything is compared bit-exactly."""
import numpy as np


def _eq(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype == np.float32:
        a, b = a.view(np.uint32), np.asarray(b, np.float32).view(np.uint32)
    assert a.shape == b.shape and np.array_equal(a, b), what


def check_layer_pack(B, L, prefix, T):
    """B: backend with get_edge_sum_weight / sample_layer / sample_root /
    sample_neighbor_layerwise / sparse_get_adj returning numpy arrays."""
    seed = int(L["seed"])
    q = L[prefix + "q"]
    e = 0
    while "%set_%d" % (prefix, e) in L:
        et = L["%set_%d" % (prefix, e)]
        _eq(L["%ssumw_%d" % (prefix, e)], B.get_edge_sum_weight(q, et), ("sumw", e))
        for c, (call, dn) in enumerate(((3, -1), (4, 4242))):
            key = "%slayer_%d_%d_" % (prefix, e, c)
            lid, lw, lt = B.sample_layer(seed, call, q, et, dn)
            _eq(L[key + "id"], lid, key)
            _eq(L[key + "w"], lw, key)
            _eq(L[key + "t"], lt, key)
        e += 1
    c = 0
    while "%sroot_%d_roots" % (prefix, c) in L:
        key = "%sroot_%d_" % (prefix, c)
        roots, w, m = L[key + "roots"], L[key + "w"], int(L[key + "m"])
        _eq(L[key + "out"], B.sample_root(seed, 50 + c, roots, w, roots.shape[1], m, -1),
            key)
        c += 1
    c = 0
    while "%slw_%d_0_nodes" % (prefix, c) in L:
        e = 0
        while "%slw_%d_%d_nodes" % (prefix, c, e) in L:
            key = "%slw_%d_%d_" % (prefix, c, e)
            nodes, et = L[key + "nodes"], L[key + "et"]
            batch, n = nodes.shape
            count = L[key + "nb"].shape[1]
            nb, ind, val, shape = B.sample_neighbor_layerwise(seed, 70 + c, nodes, et,
                                                              count, -1)
            _eq(L[key + "nb"], nb, key + "nb")
            _eq(L[key + "ind"], ind, key + "ind")
            _eq(L[key + "val"], val, key + "val")
            _eq(L[key + "shape"], shape, key + "shape")
            idx, vals = B.sparse_get_adj(nodes, L[key + "nb"].view(np.uint64), batch, n,
                                         count, et)
            _eq(L[key + "adj_idx"], idx, key + "adj_idx")
            _eq(L[key + "adj_val"], vals, key + "adj_val")
            e += 1
        c += 1
    assert e > 0 and c > 0


def check_layer_func_pack(fn, L, prefix):
    """sampleLNB with a weight function against the reference harness' outputs:
    fn(seed, call_id, nodes, edge_types, count, weight_func, default_node) ->
    (nb [batch, count], w, t, indices, values, dense_shape)."""
    seed = int(L["seed"])
    keys = ("nb", "w", "t", "ind", "val", "shape")

    def one(tag, call, nodes, et, count, wf, dn):
        got = fn(seed, call, nodes, et, count, wf, dn)
        for k, g in zip(keys, got):
            _eq(L["%s%s_%s" % (prefix, tag, k)], g, (prefix, tag, k))

    if prefix == "fx_":
        src = L["fx_t_nodes"]
        for c, (wf, dn) in enumerate((("sqrt", -1), ("sqrt", 261), ("none", -1))):
            one("lf_%d" % c, 95 + c, src, [0, 1], 10, wf, dn)
        one("lf_lone", 99, L["fx_lf_lone_nodes"], [0], 4, "sqrt", 261)
        # the reference's own assertions (neighbor_ops_test.py:159-175)
        nb = L["fx_lf_0_nb"]
        assert set(nb[0].tolist()) <= {2, 3, 4, 5} and set(nb[2].tolist()) <= {3, 4, 5}
        assert set(nb[3].tolist()) <= {3, 5}
    else:
        c = 0
        while "%slf_%d_nodes" % (prefix, c) in L:
            nodes, et = L["%slf_%d_nodes" % (prefix, c)], L["%slf_%d_et" % (prefix, c)]
            one("lf_%d" % c, 120 + c, nodes, et, L["%slf_%d_nb" % (prefix, c)].shape[1],
                "sqrt", -1)
            c += 1
        assert c >= 4


class OracleBackend:
    def __init__(self, O, G):
        self.O, self.G = O, G
        self.get_edge_sum_weight = G.get_edge_sum_weight
        self.sample_layer = G.sample_layer
        self.sample_neighbor_layerwise = G.sample_neighbor_layerwise
        self.sparse_get_adj = G.sparse_get_adj

    def sample_root(self, seed, call, roots, w, n, m, dn):
        return self.O.sample_root(seed, call, roots, w, n, m, dn)


class GpuBackend:
    """euler_amd.Graph behind the same five calls (numpy in / numpy out)."""

    def __init__(self, torch, G):
        self.t, self.G = torch, G

    def _dev(self, a, dt=np.int64):
        a = np.ascontiguousarray(np.asarray(a))
        if a.dtype == np.uint64:
            a = a.view(np.int64)
        return self.t.as_tensor(a.astype(dt, copy=False)).cuda()

    def get_edge_sum_weight(self, q, et):
        return self.G.get_edge_sum_weight(self._dev(q), et).cpu().numpy()

    def sample_layer(self, seed, call, q, et, dn):
        self.G.set_seed(seed)
        i, w, t = self.G.sample_layer(self._dev(q), et, dn, call_id=call)
        return i.cpu().numpy().view(np.uint64), w.cpu().numpy(), t.cpu().numpy()

    def sample_root(self, seed, call, roots, w, n, m, dn):
        self.G.set_seed(seed)
        out = self.G.sample_root(self._dev(roots), self._dev(w, np.float32), m, dn,
                                 call_id=call)
        return out.cpu().numpy().view(np.uint64).reshape(-1)

    def sample_neighbor_layerwise(self, seed, call, nodes, et, count, dn):
        self.G.set_seed(seed)
        nb, (ind, val, shape) = self.G.sample_neighbor_layerwise(
            self._dev(nodes), et, count, dn, call_id=call)
        return (nb.cpu().numpy(), ind.cpu().numpy(), val.cpu().numpy(),
                np.asarray(shape, np.int64))

    def sample_neighbor_layerwise_func(self, seed, call, nodes, et, count, wf, dn):
        """Graph.sample_neighbor_layerwise(weight_func=wf) + the sampled weights and
        types of its API_LOCAL_SAMPLE_L step."""
        self.G.set_seed(seed)
        nd = self._dev(nodes)
        batch, n = nd.shape
        nb, (ind, val, shape) = self.G.sample_neighbor_layerwise(nd, et, count, dn, wf,
                                                                 call_id=call)
        idx, ids, w, t = self.G.get_full_neighbor(nd.reshape(-1), et)
        nb2, lw, lt = self.G.local_sample_layer(idx, ids, w, t, batch, n, count, wf, dn,
                                                call_id=call)
        assert bool((nb2.reshape(batch, count) == nb).all())
        return (nb.cpu().numpy(), lw.cpu().numpy().reshape(batch, count),
                lt.cpu().numpy().reshape(batch, count), ind.cpu().numpy(),
                val.cpu().numpy(), np.asarray(shape, np.int64))

    def sparse_get_adj(self, nodes, nb, batch, n, m, et):
        idx, vals = self.G.sparse_get_adj_core(self._dev(nodes), self._dev(nb), n, m, et)
        return idx.cpu().numpy(), vals.cpu().numpy().view(np.uint64)
