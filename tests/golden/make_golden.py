#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/.

Runs ONLY where /root/reference exists (the build container); the GPU box and
the CPU test-suite consume the committed .npz files.  Everything written here
comes out of REFERENCE code:

  fixture_graph.npz   the reference's 6-node test graph
                      (tools/test_data/graph.json) converted by the reference's
                      own euler/tools (json2partdat) to .dat partitions, loaded
                      by the reference's own GraphBuilder/Node::DeSerialize
                      (through oracle/_ref) and exported from its Node storage.
  fixture_samples.npz sampled ids / weights / types, SampleNode ids, random
                      walks and full neighbors produced by the reference
                      sampler (oracle/_ref = reference sources + RNG seam) on
                      that fixture for fixed (seed, call_id).
  random_graph.npz    a 300-node heterogeneous random graph (raw adjacency) +
                      the same kinds of reference outputs.
  fixture_dat/        euler.meta + Node/*.dat + Edge/*.dat exactly as written by the
                      reference's euler/tools for that fixture (2 partitions).
  features.npz        dense float features of both graphs as the reference holds
                      them + the rows GetFloat32Feature / TF GetDenseFeature
                      produce for a battery of (node, feature id, dim) queries.
  layerwise.npz       layerwise sampling (sampleLNB without a weight function) and
                      SparseGetAdj on the fixture - loaded WITH its Edge/*.dat
                      partitions by the reference's own loader, so EdgeExist
                      consults the reference's Edge records - and on the random
                      graph: per-op outputs (edge weight sums, API_SAMPLE_ROOT,
                      API_SAMPLE_L, API_SPARSE_GET_ADJ) and the TF kernels'
                      results (neighbors + sparse adjacency), including the
                      batches of neighbor_ops_test.py:142-175.
                      `python make_golden.py layerwise` writes only this file.
  sparse_features.npz uint64 ("sparse") features of both graphs as the reference
                      holds them (fixture: parsed from the .dat files by its own
                      Node::DeSerialize) + the SparseTensor triples its
                      GetUint64Feature + the TF GetSparseFeature builder produce.
                      `python make_golden.py sparse` writes only this file.
  ref_tests.npz       exact expectations copied from the reference's own tests
                      (mp_ops_test.py:30-86, walk_ops_test.py:49-58,
                      unique_gather_test.cc:28-160, neighbor_ops_test.py:46-75).
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

REF = os.environ.get("EULER_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
SEED = 20240521


def convert_fixture(scratch):
    """graph.json -> .dat with the reference's python tools (py3-compatible
    part: meta + Node/Edge partitions; the index converter is py2-only)."""
    pkg = os.path.join(scratch, "euler")
    os.makedirs(pkg)
    shutil.copytree(os.path.join(REF, "euler", "tools"), os.path.join(pkg, "tools"))
    open(os.path.join(pkg, "__init__.py"), "w").close()
    tools = os.path.join(pkg, "tools")
    if not os.path.exists(os.path.join(tools, "__init__.py")):
        open(os.path.join(tools, "__init__.py"), "w").close()
    subprocess.check_call(
        ["g++", "-std=c++11", "-O2", "-fPIC", "-shared", "-I" + REF,
         os.path.join(REF, "euler/util/python_api.cc"),
         os.path.join(REF, "euler/common/hash.cc"),
         "-o", os.path.join(tools, "libeuler_util.so")])
    empty = os.path.join(scratch, "empty.cc")
    open(empty, "w").close()
    subprocess.check_call(["g++", "-shared", "-fPIC", empty, "-o",
                           os.path.join(tools, "libcommon.so")])
    data = os.path.join(scratch, "data")
    os.makedirs(data)
    subprocess.check_call(
        [sys.executable, os.path.join(tools, "generate_euler_data.py"),
         os.path.join(REF, "tools/test_data/graph.json"), data, "2"],
        cwd=scratch, env=dict(os.environ, PYTHONPATH=scratch))
    return data


def sample_pack(R, ids, T, prefix):
    """Reference outputs on the loaded graph for a battery of queries."""
    out = {}
    q = np.concatenate([ids, ids[::-1], np.array([0, 987654321], np.uint64)])
    out[prefix + "query_ids"] = q
    etl = [[0], [1], [0, 1], [1, 0], [], [T + 3], [0, 0]]
    if T > 2:
        etl += [[2], [0, 2], [2, 1], list(range(T))]
    for n, et in enumerate(etl):
        for count in (1, 5):
            idx, oid, ow, ot = R.sample_neighbor_core(SEED, 11 + n, q, et, count)
            key = "%snb_%d_%d_" % (prefix, n, count)
            out[key + "et"] = np.array(et, np.int32)
            out[key + "idx"] = idx
            out[key + "id"] = oid
            out[key + "w"] = ow
            out[key + "t"] = ot
    for n, et in enumerate([[0], [1], [0, 1], list(range(T))]):
        idx, oid, ow, ot = R.get_full_neighbor(q, et)
        key = "%sfull_%d_" % (prefix, n)
        out[key + "et"] = np.array(et, np.int32)
        out[key + "idx"] = idx
        out[key + "id"] = oid
        out[key + "w"] = ow
        out[key + "t"] = ot
    out[prefix + "node_order"] = R.node_order()
    for n, nt in enumerate([[-1], [0], [1], [0, 1]]):
        out["%ssn_%d_types" % (prefix, n)] = np.array(nt, np.int32)
        out["%ssn_%d" % (prefix, n)] = R.sample_node(SEED, 100 + n, nt, 64)
    for t in (-1, 0, 1):
        i, w, p, a, s = R.alias_table(t)
        out["%salias_%d_ids" % (prefix, t + 1)] = i
        out["%salias_%d_w" % (prefix, t + 1)] = w
        out["%salias_%d_prob" % (prefix, t + 1)] = p
        out["%salias_%d_alias" % (prefix, t + 1)] = a
    starts = q.astype(np.int64)
    L = 6
    et_all = np.tile(np.arange(T, dtype=np.int32), (L, 1))
    out[prefix + "walk_et"] = et_all
    out[prefix + "walk_11"] = R.random_walk(SEED, 200, starts, et_all, L, 1.0, 1.0, -1)
    out[prefix + "walk_n2v"] = R.random_walk(SEED, 300, starts, et_all, L, 0.25, 4.0, -1)
    out[prefix + "walk_n2v_b"] = R.random_walk(SEED, 400, starts, et_all, L, 2.0, 0.5, 777)
    return out


def layer_pack(R, ids, T, prefix, rng):
    """Reference outputs of the layerwise op chain for a battery of queries."""
    out = {}
    etl = [[0], [1], [0, 1], [], [T + 3]] + ([[2, 0], list(range(T))] if T > 2 else [])
    q = np.concatenate([ids, ids[::-1], np.array([0, 987654321], np.uint64)])
    out[prefix + "q"] = q
    for e, et in enumerate(etl):
        out["%set_%d" % (prefix, e)] = np.array(et, np.int32)
        out["%ssumw_%d" % (prefix, e)] = R.get_edge_sum_weight(q, et)
        for c, (call, dn) in enumerate(((3, -1), (4, 4242))):
            lid, lw, lt = R.sample_layer(SEED, call, q, et, dn)
            key = "%slayer_%d_%d_" % (prefix, e, c)
            out[key + "id"], out[key + "w"], out[key + "t"] = lid, lw, lt
    # API_SAMPLE_ROOT on explicit weights (zero rows, single entries, ties)
    for c, (n, m) in enumerate(((1, 4), (3, 10), (17, 9))):
        batch = 6
        roots = rng.choice(ids, (batch, n)).astype(np.uint64)
        w = (rng.random((batch, n)) * 4).astype(np.float32)
        w[rng.random((batch, n)) < 0.3] = 0
        w[1] = 0
        w[2] = 1.0
        key = "%sroot_%d_" % (prefix, c)
        out[key + "roots"], out[key + "w"] = roots, w
        out[key + "m"] = np.int32(m)
        out[key + "out"] = R._sample_root(SEED, 50 + c, roots, w, n, m, -1)
    # the whole TF op + SparseGetAdj on batches of graph nodes
    shapes = ((4, 3, 10), (2, 5, 4), (3, 1, 6))
    for c, (batch, n, count) in enumerate(shapes):
        nodes = rng.choice(ids, (batch, n)).astype(np.uint64)
        if n > 1:
            nodes[0, 1] = nodes[0, 0]
        nodes[-1, -1] = 987654321
        for e, et in enumerate(etl[:3] + etl[-1:]):
            key = "%slw_%d_%d_" % (prefix, c, e)
            nb, ind, val, shape = R.sample_neighbor_layerwise(SEED, 70 + c, nodes, et,
                                                              count, -1)
            out[key + "nodes"], out[key + "et"] = nodes, np.array(et, np.int32)
            out[key + "nb"], out[key + "ind"], out[key + "val"] = nb, ind, val
            out[key + "shape"] = shape
            idx, vals = R.sparse_get_adj(nodes, nb.view(np.uint64), batch, n, count, et)
            out[key + "adj_idx"], out[key + "adj_val"] = idx, vals
    return out


def main_layerwise():
    O.build(ref=True)
    assert O.have_ref(), "oracle/_ref must be built from " + REF
    out = {"seed": np.uint64(SEED)}
    scratch = tempfile.mkdtemp(prefix="euler_golden_")
    try:
        data = convert_fixture(scratch)
        R = O.RefGraph.load_all(data, 2)
        assert R.num_edges() > 0
        ids = np.sort(R.node_order())
        out.update(layer_pack(R, ids, 2, "fx_", np.random.default_rng(11)))
        # the batches of the reference's own test (neighbor_ops_test.py:142-175)
        src = np.array([[1, 2, 3], [1, 2, 3], [2, 3, 4], [2, 2, 4]], np.uint64)
        nb, ind, val, shape = R.sample_neighbor_layerwise(SEED, 90, src, [0, 1], 10, -1)
        out.update(fx_t_nodes=src, fx_t_nb=nb, fx_t_ind=ind, fx_t_val=val,
                   fx_t_shape=shape)
        # sampleLNB WITH a weight function (API_LOCAL_SAMPLE_L), the reference's own
        # test batches (neighbor_ops_test.py:159-175) and a default_node whose low
        # byte shows the op's memset fill
        for c, (wf, dn) in enumerate((("sqrt", -1), ("sqrt", 261), ("none", -1))):
            r = R.sample_neighbor_layerwise_func(SEED, 95 + c, src, [0, 1], 10, wf, dn)
            for k, v in zip(("nb", "w", "t", "ind", "val", "shape"), r):
                out["fx_lf_%d_%s" % (c, k)] = v
        lone = np.array([[6, 6], [987654321, 6], [1, 5]], np.uint64)
        out["fx_lf_lone_nodes"] = lone
        r = R.sample_neighbor_layerwise_func(SEED, 99, lone, [0], 4, "sqrt", 261)
        for k, v in zip(("nb", "w", "t", "ind", "val", "shape"), r):
            out["fx_lf_lone_%s" % k] = v
        # every (src, dst, type) the reference holds an Edge record for
        g = np.load(os.path.join(OUT, "fixture_graph.npz"))
        trip = []
        for s_ in ids:
            for d_ in np.concatenate([ids, [0, 99]]):
                for t_ in (0, 1, 2):
                    if R.edge_exist(s_, d_, t_):
                        trip.append((int(s_), int(d_), t_))
        out["fx_edges"] = np.array(trip, np.int64)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    g = np.load(os.path.join(OUT, "random_graph.npz"))
    n = len(g["row_id"])
    T = int(g["n_types"])
    R = O.RefGraph.build_raw(g["row_id"], g["raw_seg_ptr"], g["raw_nbr"], g["raw_w"], T,
                             g["node_type"], g["node_weight"])
    R.add_edges_from_adjacency()
    out.update(layer_pack(R, g["row_id"][:40], T, "rg_", np.random.default_rng(12)))
    rng = np.random.default_rng(13)
    for c, (batch, n, count, et) in enumerate(((3, 4, 12, [0, 1, 2]), (1, 40, 64, [2]),
                                               (5, 1, 3, [0, 2]), (2, 60, 30, [0, 1, 2]))):
        nodes = rng.choice(g["row_id"], (batch, n)).astype(np.uint64)
        nodes[0, 0] = g["row_id"][0]                  # the hub row: many duplicates
        out["rg_lf_%d_nodes" % c], out["rg_lf_%d_et" % c] = nodes, np.array(et, np.int32)
        r = R.sample_neighbor_layerwise_func(SEED, 120 + c, nodes, et, count, "sqrt", -1)
        for k, v in zip(("nb", "w", "t", "ind", "val", "shape"), r):
            out["rg_lf_%d_%s" % (c, k)] = v
    np.savez_compressed(os.path.join(OUT, "layerwise.npz"), **out)
    print("layerwise golden vectors written")


def main():
    O.build(ref=True)
    assert O.have_ref(), "oracle/_ref must be built from " + REF
    scratch = tempfile.mkdtemp(prefix="euler_golden_")
    try:
        data = convert_fixture(scratch)
        T = 2
        R = O.RefGraph.load(data, T)
        ids = np.sort(R.node_order())
        csr = R.export_csr(ids)
        np.savez(os.path.join(OUT, "fixture_graph.npz"),
                 row_id=csr.row_id, row_ptr=csr.row_ptr, type_end=csr.type_end,
                 nbr=csr.nbr, prefix_w=csr.prefix_w, type_prefix=csr.type_prefix,
                 node_type=csr.node_type, node_weight=csr.node_weight,
                 n_types=np.int32(T))
        pack = sample_pack(R, ids, T, "")
        np.savez(os.path.join(OUT, "fixture_samples.npz"), seed=np.uint64(SEED),
                 **pack)
        # the reference tool's own output files: input of the .dat reader test
        dst = os.path.join(OUT, "fixture_dat")
        shutil.rmtree(dst, ignore_errors=True)
        os.makedirs(os.path.join(dst, "Node"))
        shutil.copy(os.path.join(data, "euler.meta"), dst)
        for fn in sorted(os.listdir(os.path.join(data, "Node"))):
            shutil.copy(os.path.join(data, "Node", fn), os.path.join(dst, "Node"))
        # ... and its Edge partitions (input of euler_gpu_dat_verify_edges)
        os.makedirs(os.path.join(dst, "Edge"))
        for fn in sorted(os.listdir(os.path.join(data, "Edge"))):
            shutil.copy(os.path.join(data, "Edge", fn), os.path.join(dst, "Edge"))
    finally:
        shutil.rmtree(scratch, ignore_errors=True)

    # ---- random heterogeneous graph built through the reference Node::Init
    rng = np.random.default_rng(7)
    n, T = 300, 3
    ids = np.sort(rng.choice(np.arange(1, 5000), n, replace=False)).astype(np.uint64)
    deg = rng.integers(0, 9, size=(n, T))
    deg[rng.random((n, T)) < 0.25] = 0
    deg[0, :] = [40, 0, 300]          # one hub row
    seg = np.zeros(n * T + 1, np.int64)
    seg[1:] = np.cumsum(deg.reshape(-1))
    E = int(seg[-1])
    nbr = rng.choice(ids, E).astype(np.uint64)
    w = (rng.random(E) * 7.5 + 0.5).astype(np.float32)
    w[rng.random(E) < 0.05] = 0
    nt = rng.integers(0, 2, n).astype(np.int32)
    nw = (rng.random(n) * 3 + 0.1).astype(np.float32)
    R = O.RefGraph.build_raw(ids, seg, nbr, w, T, nt, nw)
    csr = R.export_csr(ids)
    pack = sample_pack(R, ids[:40], T, "")
    np.savez_compressed(
        os.path.join(OUT, "random_graph.npz"), seed=np.uint64(SEED),
        raw_seg_ptr=seg, raw_nbr=nbr, raw_w=w,
        row_id=csr.row_id, row_ptr=csr.row_ptr, type_end=csr.type_end,
        nbr=csr.nbr, prefix_w=csr.prefix_w, type_prefix=csr.type_prefix,
        node_type=csr.node_type, node_weight=csr.node_weight,
        n_types=np.int32(T), **pack)

    # ---- exact expectations stated in the reference's own tests
    ref_tests = dict(
        # tf_euler/python/euler_ops/mp_ops_test.py:30-36
        scatter_add_x=np.array([[1., 2.], [3., 4.], [5., 6.]], np.float32),
        scatter_idx=np.array([1, 0, 1], np.int32),
        scatter_add_out=np.array([[3., 4.], [6., 8.]], np.float32),
        # :46-54
        scatter_mean_out=np.array([[3., 4.], [3., 4.]], np.float32),
        # :64-70
        scatter_max_x=np.array([[1., 6.], [3., 4.], [5., 2.]], np.float32),
        scatter_max_out=np.array([[3., 4.], [5., 6.]], np.float32),
        # :80-86
        gather_x=np.array([[1., 2.], [3., 4.], [5., 6.]], np.float32),
        gather_idx=np.array([1, 0, 1, 2], np.int32),
        gather_out=np.array([[3., 4.], [1., 2.], [3., 4.], [5., 6.]], np.float32),
        # tf_euler/python/euler_ops/walk_ops_test.py:49-58
        gen_pair_in=np.array([[1, 2, 3, 4, 5, 6, 7, 8, 9]], np.int64),
        gen_pair_out=np.array(
            [[[1, 2], [1, 3], [2, 1], [2, 3], [2, 4], [3, 2], [3, 1],
              [3, 4], [3, 5], [4, 3], [4, 2], [4, 5], [4, 6], [5, 4],
              [5, 3], [5, 6], [5, 7], [6, 5], [6, 4], [6, 7], [6, 8],
              [7, 6], [7, 5], [7, 8], [7, 9], [8, 7], [8, 6], [8, 9],
              [9, 8], [9, 7]]], np.int64),
    )
    np.savez(os.path.join(OUT, "ref_tests.npz"), **ref_tests)
    print("golden vectors written to", OUT)
    main_layerwise()


def feature_goldens():
    """features.npz: dense float features.  Fixture: what the reference's own
    loader (Node::DeSerialize) holds for tools/test_data/graph.json, exported
    from its Node storage, and the rows the reference's GetFloat32Feature +
    the TF GetDenseFeature copy loop produce.  Random graph: random ragged
    features pushed into the reference nodes, same outputs."""
    O.build(ref=True)
    assert O.have_ref(), "oracle/_ref must be built from " + REF
    out = {}
    scratch = tempfile.mkdtemp(prefix="euler_golden_")
    try:
        data = convert_fixture(scratch)
        R = O.RefGraph.load(data, 2)
        ids = np.sort(R.node_order())
        F = R.export_float_features(ids)
        out.update(fx_ids=ids, fx_n_float=np.int32(F.n_float), fx_feat_ptr=F.feat_ptr,
                   fx_feat_idx=F.feat_idx, fx_feat_val=F.feat_val)
        q = np.concatenate([ids, ids[::-1], [0, 987654321]]).astype(np.int64)
        out["fx_query"] = q
        fids = list(range(F.n_float)) + [F.n_float + 2]
        dims = [int(max(F.feat_idx.reshape(len(ids), -1)[:, f] -
                        (F.feat_idx.reshape(len(ids), -1)[:, f - 1] if f else 0)).max()
                    if False else 0) for f in range(F.n_float)]
        idx2 = F.feat_idx.reshape(len(ids), -1)
        dims = [int((idx2[:, f] - (idx2[:, f - 1] if f else 0)).max()) + (f % 2)
                for f in range(F.n_float)] + [3]
        out["fx_fids"] = np.array(fids, np.int32)
        out["fx_dims"] = np.array(dims, np.int32)
        for k, o in enumerate(R.get_dense_feature(q, fids, dims)):
            out["fx_dense_%d" % k] = o
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    g = np.load(os.path.join(OUT, "random_graph.npz"))
    ids = g["row_id"]
    n = len(ids)
    R = O.RefGraph.build_raw(ids, g["raw_seg_ptr"], g["raw_nbr"], g["raw_w"], 3,
                             g["node_type"], g["node_weight"])
    rng = np.random.default_rng(11)
    per = []
    for i in range(n):
        slots = [list(rng.standard_normal(16).astype(np.float32)),
                 list(rng.standard_normal(int(rng.integers(0, 6))).astype(np.float32)),
                 list(rng.standard_normal(100).astype(np.float32))]
        per.append(slots[:int(rng.integers(1, 4))] if i % 5 == 0 else slots)
    F = O.DenseFeatures.from_lists(per)
    R.set_float_features(ids, F)
    out.update(rg_n_float=np.int32(F.n_float), rg_feat_ptr=F.feat_ptr,
               rg_feat_idx=F.feat_idx, rg_feat_val=F.feat_val)
    q = np.concatenate([rng.choice(ids, 200), [0, 4999999]]).astype(np.int64)
    out["rg_query"] = q
    fids, dims = [0, 1, 2, 5], [16, 5, 100, 4]
    out["rg_fids"] = np.array(fids, np.int32)
    out["rg_dims"] = np.array(dims, np.int32)
    for k, o in enumerate(R.get_dense_feature(q, fids, dims)):
        out["rg_dense_%d" % k] = o
    np.savez_compressed(os.path.join(OUT, "features.npz"), **out)
    print("feature goldens written")


def sparse_feature_goldens():
    O.build(ref=True)
    assert O.have_ref(), "oracle/_ref must be built from " + REF
    out = {}
    scratch = tempfile.mkdtemp(prefix="euler_golden_")
    try:
        data = convert_fixture(scratch)
        R = O.RefGraph.load(data, 2)
        ids = np.sort(R.node_order())
        F = R.export_u64_features(ids)
        out.update(fx_ids=ids, fx_n_u64=np.int32(F.n_u64), fx_feat_ptr=F.feat_ptr,
                   fx_feat_idx=F.feat_idx, fx_feat_val=F.feat_val)
        q = np.concatenate([ids, ids[::-1], [0, 987654321]]).astype(np.uint64)
        out["fx_query"] = q
        fids = list(range(F.n_u64)) + [F.n_u64 + 2, -1]
        dvs = [0, 7, -1, 0, 5, 9][:len(fids)] + [0] * max(0, len(fids) - 6)
        out["fx_fids"], out["fx_defaults"] = np.array(fids, np.int32), np.array(dvs, np.int64)
        for k, (ind, val, shape) in enumerate(R.get_sparse_feature(q, fids, dvs)):
            out["fx_sp_%d_ind" % k], out["fx_sp_%d_val" % k] = ind, val
            out["fx_sp_%d_shape" % k] = shape
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    g = np.load(os.path.join(OUT, "random_graph.npz"))
    ids = g["row_id"]
    n = len(ids)
    R = O.RefGraph.build_raw(ids, g["raw_seg_ptr"], g["raw_nbr"], g["raw_w"], 3,
                             g["node_type"], g["node_weight"])
    rng = np.random.default_rng(21)
    per = []
    for i in range(n):
        slots = [list(rng.integers(0, 2 ** 63, 3, dtype=np.uint64) * 2 + 1),
                 list(rng.integers(0, 1000, int(rng.integers(0, 6)), dtype=np.uint64)),
                 list(rng.integers(0, 2 ** 40, 70 if i % 17 == 0 else 1, dtype=np.uint64))]
        per.append(slots[:int(rng.integers(1, 4))] if i % 5 == 0 else slots)
    F = O.SparseFeatures.from_lists(per)
    R.set_u64_features(ids, F)
    out.update(rg_n_u64=np.int32(F.n_u64), rg_feat_ptr=F.feat_ptr, rg_feat_idx=F.feat_idx,
               rg_feat_val=F.feat_val)
    q = np.concatenate([rng.choice(ids, 200), [0, 4999999]]).astype(np.uint64)
    out["rg_query"] = q
    fids, dvs = [0, 1, 2, 5], [0, 0, 123, 4]
    out["rg_fids"], out["rg_defaults"] = np.array(fids, np.int32), np.array(dvs, np.int64)
    for k, (ind, val, shape) in enumerate(R.get_sparse_feature(q, fids, dvs)):
        out["rg_sp_%d_ind" % k], out["rg_sp_%d_val" % k] = ind, val
        out["rg_sp_%d_shape" % k] = shape
    np.savez_compressed(os.path.join(OUT, "sparse_features.npz"), **out)
    print("sparse feature goldens written")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "features":
        feature_goldens()
    elif len(sys.argv) > 1 and sys.argv[1] == "sparse":
        sparse_feature_goldens()
    elif len(sys.argv) > 1 and sys.argv[1] == "layerwise":
        main_layerwise()
    else:
        main()
        feature_goldens()
        sparse_feature_goldens()
