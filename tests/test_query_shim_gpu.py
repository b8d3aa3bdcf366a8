"""The client-side boundary: euler::Query / QueryProxy::RunAsyncGremlin fed with
the LITERAL query strings the reference's TF kernels build (include/euler_query.h),
the API_SAMPLE_NB post-process of core/kernels/sample_neighbor_op.cc:86-132, and
the call-id behaviour of an untouched OpKernelContext.  Results are compared with
the oracle (bit-exact)."""
import ctypes as C

import numpy as np
import pytest

from conftest import make_random_graph

pytestmark = pytest.mark.gpu

K_INT32, K_UINT64 = 2, 7          # euler::DataType (core/framework/types.h:26-39)


def t2n(t):
    return t.detach().cpu().numpy()


def run_query(L, gremlin, inputs, result, dtype, capacity):
    """inputs: list of (name, euler dtype, numpy array or python int for a scalar)."""
    n = len(inputs)
    names = (C.c_char_p * n)(*[nm.encode() for nm, _, _ in inputs])
    dts = (C.c_int32 * n)(*[dt for _, dt, _ in inputs])
    arrs, counts = [], []
    for _, dt, v in inputs:
        npdt = np.int32 if dt == K_INT32 else np.uint64
        if np.isscalar(v):
            arrs.append(np.array([v], npdt)); counts.append(-1)
        else:
            arrs.append(np.ascontiguousarray(v, npdt)); counts.append(len(v))
    cnt = (C.c_int64 * n)(*counts)
    ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
    out = np.zeros(capacity, dtype)
    rc = L.euler_query_run(gremlin.encode(), n, names, dts, cnt, ptrs, result.encode(),
                           out.ctypes.data_as(C.c_void_p), out.nbytes)
    if rc < 0:
        return rc, None
    return rc, out[:rc // out.itemsize]


@pytest.fixture(scope="module")
def pair(EA, O):
    rng = np.random.default_rng(21)
    ids, seg, nbr, w, nt, nw = make_random_graph(rng, 3000, 3, max_deg=14, id_space=10 ** 9)
    csr = O.csr_from_raw(ids, seg, nbr, w, 3, nt, nw)
    G = EA.Graph.from_csr(csr.row_id, csr.row_ptr, csr.type_end, csr.nbr, csr.prefix_w,
                          csr.type_prefix, csr.n_types, csr.node_type, csr.node_weight)
    OG = O.OracleGraph(csr)
    OG.build_node_sampler()
    return G, OG, ids, rng


def test_literal_tf_kernel_queries(EA, O, torch_cuda, pair):
    from euler_amd import _lib
    L = _lib.lib()
    G, OG, ids, rng = pair
    # the graph the proxy serves (InitQueryProxy("...") installs the process default
    # for a data directory; here QueryProxy::Init(graph) with a borrowed handle)
    L.euler_query_set_graph(G._h)
    q = np.concatenate([rng.choice(ids, 200), [0, 12345]]).astype(np.uint64)
    n = len(q)

    # ---- tf_euler/kernels/sample_neighbor_op.cc:36-40 (condition empty)
    default_node, count = -1, 5
    gremlin = "v(nodes).sampleNB(edge_types, nb_count," + str(default_node) + ").as(nb)"
    et = np.array([0, 2], np.int32)
    L.euler_query_set_seed(77)                       # call ids restart at 0
    inputs = [("nodes", K_UINT64, q), ("edge_types", K_INT32, et), ("nb_count", K_INT32, [count])]
    rc, got_ids = run_query(L, gremlin, inputs, "nb:1", np.uint64, n * count)
    assert rc == n * count * 8
    ridx, rid, rw, rt = OG.sample_neighbor_core(77, 0, q, et, count)
    assert np.array_equal(got_ids, rid)
    L.euler_query_set_seed(77)
    for name, dt, want in (("nb:0", np.int32, ridx.reshape(-1)), ("nb:2", np.float32, rw),
                           ("nb:3", np.int32, rt)):
        L.euler_query_set_seed(77)
        rc, got = run_query(L, gremlin, inputs, name, dt, len(want))
        assert rc == want.nbytes and np.array_equal(got, want), name

    # ---- a query whose result tensors are large enough for the pinned host blocks (>= 16 KB,
    # op_framework.cc: HostAlloc): run three times so that blocks come back from the cache
    big = rng.choice(ids, 20000).astype(np.uint64)
    binputs = [("nodes", K_UINT64, big), ("edge_types", K_INT32, et), ("nb_count", K_INT32, [count])]
    _, _rid, _rw, _rt = OG.sample_neighbor_core(91, 0, big, et, count)
    for _ in range(3):
        L.euler_query_set_seed(91)
        rc, got_ids = run_query(L, gremlin, binputs, "nb:1", np.uint64, len(big) * count)
        assert rc == len(big) * count * 8 and np.array_equal(got_ids, _rid)
        L.euler_query_set_seed(91)
        rc, got_w = run_query(L, gremlin, binputs, "nb:2", np.float32, len(big) * count)
        assert np.array_equal(got_w, _rw)

    # ---- tf_euler/kernels/sample_fanout_op.cc:37-42, counts [4, 3]
    counts = [4, 3]
    ss = "v(nodes)"
    for i in range(len(counts)):
        ss += ".sampleNB(et_%d,nb_count_%d,%d).as(nb_%d)" % (i, i, default_node, i)
    fin = [("nodes", K_UINT64, q)]
    for i, c in enumerate(counts):
        fin += [("et_%d" % i, K_INT32, np.array([i, 2], np.int32)), ("nb_count_%d" % i, K_INT32, [c])]
    L.euler_query_set_seed(78)
    rc, hop2 = run_query(L, ss, fin, "nb_1:1", np.uint64, n * 12)
    assert rc == n * 12 * 8
    _, h1, _, _ = OG.sample_neighbor_core(78, 0, q, [0, 2], 4)
    _, h2, _, _ = OG.sample_neighbor_core(78, 1, h1, [1, 2], 3)   # hop 2 from the core ids (0 = sentinel)
    assert np.array_equal(hop2, h2)
    L.euler_query_set_seed(78)
    rc, idx2 = run_query(L, ss, fin, "nb_1:0", np.int32, n * 4 * 2)
    assert np.array_equal(idx2.reshape(-1, 2)[:, 1] - idx2.reshape(-1, 2)[:, 0], np.full(n * 4, 3))

    # ---- tf_euler/kernels/random_walk_op.cc:181-185 (p = q = 1), walk_len 3
    ss = "v(nodes)"
    for i in range(3):
        ss += ".sampleNB(et_%d, nb_count_, %d).as(nb_%d)" % (i, default_node, i)
    win = [("nodes", K_UINT64, q), ("nb_count_", K_INT32, [1])]
    for i in range(3):
        win.append(("et_%d" % i, K_INT32, np.array([0, 1, 2], np.int32)))
    L.euler_query_set_seed(79)
    rc, step3 = run_query(L, ss, win, "nb_2:1", np.uint64, n)
    cur = q
    for s in range(3):
        _, cur, _, _ = OG.sample_neighbor_core(79, s, cur, [0, 1, 2], 1)
    assert rc == n * 8 and np.array_equal(step3, cur)

    # ---- tf_euler/kernels/random_walk_op.cc:73 (node2vec neighbour fetch)
    rc, nb_idx = run_query(L, "v(nodes).outV(edge_types).as(nb)",
                           [("nodes", K_UINT64, q), ("edge_types", K_INT32, np.array([0, 1], np.int32))],
                           "nb:0", np.int32, n * 2)
    widx, wid, ww, wt = OG.get_full_neighbor(q, [0, 1])
    assert np.array_equal(nb_idx.reshape(-1, 2), widx)
    rc, nb_ids = run_query(L, "v(nodes).outV(edge_types).as(nb)",
                           [("nodes", K_UINT64, q), ("edge_types", K_INT32, np.array([0, 1], np.int32))],
                           "nb:1", np.uint64, len(wid) + 8)
    assert np.array_equal(nb_ids, wid)
    # ... and the input tensor is a result under its own name (:129)
    rc, back = run_query(L, "v(nodes).outV(edge_types).as(nb)",
                         [("nodes", K_UINT64, q), ("edge_types", K_INT32, np.array([0, 1], np.int32))],
                         "nodes", np.uint64, n)
    assert np.array_equal(back, q)

    # ---- tf_euler/kernels/sample_node_op.cc:63,72
    L.euler_query_set_seed(80)
    rc, nodes = run_query(L, "sampleN(node_type, count).as(id)",
                          [("node_type", K_INT32, 1), ("count", K_INT32, 64)], "id:0", np.uint64, 64)
    assert rc == 64 * 8
    assert np.array_equal(nodes, OG.sample_node(80, 0, [1], 64))

    # conditions / unknown shapes: logged, no result (Q12)
    rc, _ = run_query(L, "v(nodes).sampleNB(edge_types, nb_count,-1).has(price gt 3).as(nb)", inputs,
                      "nb:1", np.uint64, 8)
    assert rc == -1
    rc, _ = run_query(L, "e(edges).as(x)", inputs, "x:0", np.uint64, 8)
    assert rc == -1
    # without set_seed two runs of the same query differ (fresh call ids per op)
    a = run_query(L, gremlin, inputs, "nb:1", np.uint64, n * count)[1]
    b = run_query(L, gremlin, inputs, "nb:1", np.uint64, n * count)[1]
    assert not np.array_equal(a, b)
    L.euler_query_set_graph(None)


def test_api_sample_nb_post_process(EA, O, torch_cuda, pair):
    """DAGNodeProto.post_process of API_SAMPLE_NB (sample_neighbor_op.cc:86-132):
    order_by / limit act on rows with samples; empty rows get count x (0, 0.0, 0)
    afterwards."""
    from euler_amd import _lib
    L = _lib.lib()
    G, OG, ids, rng = pair
    q = np.concatenate([rng.choice(ids, 300), [0, 999]]).astype(np.uint64)
    n, count = len(q), 6
    et = np.array([1], np.int32)
    for post, kw in ((b"order_by id", dict(order_by="id")),
                     (b"order_by weight desc;limit 2", dict(order_by="weight", desc=True, limit=2)),
                     (b"limit 4", dict(limit=4))):
        idx = np.zeros((n, 2), np.int32); oid = np.zeros(n * count, np.uint64)
        ow = np.zeros(n * count, np.float32); ot = np.zeros(n * count, np.int32)
        got = L.euler_op_run_sample_nb_post(G._h, 5, 9, q.ctypes.data_as(_lib.u64p), n,
                                            et.ctypes.data_as(_lib.i32p), 1, count, post,
                                            idx.ctypes.data_as(_lib.i32p),
                                            oid.ctypes.data_as(_lib.u64p),
                                            ow.ctypes.data_as(_lib.f32p),
                                            ot.ctypes.data_as(_lib.i32p))
        ridx, rid, rw, rt = OG.sample_neighbor_core(5, 9, q, et, count)
        rows_id = rid.reshape(n, count); rows_w = rw.reshape(n, count); rows_t = rt.reshape(n, count)
        empty = OG.sample_neighbor(5, 9, q.astype(np.int64), et, count, -7)[0][:, 0] == -7
        # post-process the non-empty rows with the oracle's restatement of
        # get_neighbor_op.cc:117-168 (same grammar and comparator)
        keep = ~empty
        kidx = np.zeros((n, 2), np.int32)
        kidx[:, 1] = np.cumsum(np.where(keep, count, 0)); kidx[:, 0] = kidx[:, 1] - np.where(keep, count, 0)
        pidx, pid, pw, pt = O.neighbor_post_process(kidx, rows_id[keep].reshape(-1),
                                                    rows_w[keep].reshape(-1),
                                                    rows_t[keep].reshape(-1), **kw)
        w_id, w_w, w_t, w_idx, o = [], [], [], [], 0
        for i in range(n):
            if empty[i]:
                w_id += [0] * count; w_w += [0.0] * count; w_t += [0] * count; ln = count
            else:
                b, e = pidx[i]
                w_id += list(pid[b:e]); w_w += list(pw[b:e]); w_t += list(pt[b:e]); ln = e - b
            w_idx.append((o, o + ln)); o += ln
        assert got == o, post
        assert np.array_equal(idx, np.array(w_idx, np.int32))
        assert np.array_equal(oid[:o], np.array(w_id, np.uint64))
        assert np.array_equal(ow[:o], np.array(w_w, np.float32))
        assert np.array_equal(ot[:o], np.array(w_t, np.int32))


def test_eight_queries_in_flight(EA, O, torch_cuda, pair):
    """The reference's client keeps 8 queries in flight (client/query_proxy.cc:205-210): 8 caller threads
    run `outV` queries (no draws, so the answers do not depend on the order the calls are numbered in)
    whose result tensors are large enough for the pinned host blocks - concurrent proxy threads, staging
    arenas and block cache; every answer == the oracle's."""
    import threading
    from euler_amd import _lib
    L = _lib.lib()
    G, OG, ids, rng = pair
    L.euler_query_set_graph(G._h)
    try:
        et = np.array([0, 1, 2], np.int32)
        sets = [rng.choice(ids, 30000 + 1000 * k).astype(np.uint64) for k in range(8)]
        want = [OG.get_full_neighbor(s, [0, 1, 2]) for s in sets]
        errors = []

        def worker(k):
            try:
                for _ in range(6):
                    inputs = [("nodes", K_UINT64, sets[k]), ("edge_types", K_INT32, et)]
                    rc, nb_ids = run_query(L, "v(nodes).outV(edge_types).as(nb)", inputs, "nb:1", np.uint64,
                                           len(want[k][1]) + 8)
                    if rc != len(want[k][1]) * 8 or not np.array_equal(nb_ids, want[k][1]):
                        errors.append((k, "ids", rc))
                    rc, nb_w = run_query(L, "v(nodes).outV(edge_types).as(nb)", inputs, "nb:2", np.float32,
                                         len(want[k][2]) + 8)
                    if not np.array_equal(nb_w, want[k][2]):
                        errors.append((k, "weights", rc))
            except Exception as e:        # noqa: BLE001
                errors.append((k, repr(e)))
        ts = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors, errors[:3]
    finally:
        L.euler_query_set_graph(None)


def test_cpp_query_clients_example(torch_cuda):
    """examples/cpp/query_clients: a C++ client of include/euler_query.h (AllocInput, RunAsyncGremlin,
    GetResult on the chain sample_fanout_op.cc builds) with four caller threads; every query returns its
    nb_1:1 tensor (the program exits 2 otherwise)."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "examples", "cpp", "query_clients")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(exe)])
    out = subprocess.run([exe, "200000", "256", "4", "20"], check=True, capture_output=True, text=True,
                         timeout=300).stdout
    lines = [l for l in out.strip().splitlines() if "sampled edges/s" in l]
    assert len(lines) == 2 and lines[0].startswith("callers 1 ") and lines[1].startswith("callers 4 "), out
