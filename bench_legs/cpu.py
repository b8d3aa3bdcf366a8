"""cpu_baseline legs: the reference (oracle/_ref) or the oracle timed on the host cores - the checker
as a reported baseline, never the thing measured as the product."""
import argparse
import gc
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

from .common import *          # noqa: F401,F403

__all__ = ['cpu_baseline', '_cpu_node_and_walk_cells', '_n2v_cell', 'cpu_walk_cell', '_walk_cell', '_ref_graph', 'cpu_hetero_cell']


def cpu_baseline(args):
    """SURVEY 8(d) protocol.  The reference sampler (oracle/_ref = the reference's
    own sources + RNG seam) and the GPU run the SAME graph (device generator ==
    host generator, tests/test_gpu_parity.py::test_synthetic_graph_matches_host_
    generator), the SAME roots and the SAME batch sizes (B = 1 024, SURVEY 8's latency
    configuration - the reference's examples default to less still, examples/graphsage/
    run_graphsage.py:35 batch_size 32 - and B = 131 072, the metric's), and the same DAG:
    per hop ID_UNIQUE -> API_SAMPLE_NB -> DATA_GATHER (parser/compiler.cc:76-90;
    oracle/ref_harness.cc: euler_ref_bench_fanout_dag).  Two named CPU numbers
    per batch size:
      as_shipped  USE_OPENMP off: 8 concurrent single-threaded queries (the
                  client pool, client/query_proxy.cc:205-210)
      best        the better of (a) more concurrent queries, (b) -DOPENMP batch
                  loop over `omp_threads` threads, one query at a time
    Rounds: 5 warm-up + 30 timed (median, p10 / p90) - except the cells of
    B = 131 072, which take seconds per round: --cpu-protocol quick (default)
    gives them 1 + 5 rounds, --cpu-protocol full the whole 5 + 30."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    n = args.cpu_nodes
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        avail = 0
    fit_note = "%d nodes / ~%d edges" % (n, 10 * n)
    # the reference's Node objects + unordered_map + our staging arrays: ~1.3 KB / node
    if avail and n * 1400 > avail * 0.8:
        n = max(1_000_000, int(avail * 0.8 / 1400) // 1_000_000 * 1_000_000)
        fit_note = "%d nodes (host has %.0f GB available: %d would not fit)" % (
            n, avail / 2 ** 30, args.cpu_nodes)
    if not O.have_ref():
        return {"value": None, "unit": "sampled edges/s", "cores": cores, "kind": "reference",
                "error": "oracle/_ref/libeuler_ref.so missing"}
    build_threads = min(32, cores)
    t0 = time.time()
    po = O.synth_params(GRAPH_SEED, n, 10 * n, weighted=True)
    csr = O.synth_csr(po, threads=build_threads)
    # per-edge weights as f32 differences of the running sums (timing only: the
    # reference's Node::Init re-accumulates them)
    w = csr.prefix_w.copy()
    w[1:] -= csr.prefix_w[:-1]
    starts = csr.row_ptr[:-1]
    w[starts] = csr.prefix_w[starts]
    n_edges = int(len(csr.nbr))
    R = O.RefGraph.build_raw(csr.row_id, csr.row_ptr, csr.nbr, w, 1, threads=build_threads,
                             build_sampler=True)        # (the global node sampler too: the K2 cell)
    del csr, w
    build_s = time.time() - t0
    side_cells = _cpu_node_and_walk_cells(R, n, cores)
    full = args.cpu_protocol == "full"
    rng = np.random.default_rng(1)
    shipped_threads = min(8, cores)
    # "best case" = OMP_NUM_THREADS = nproc.  Candidates: 32 threads (where the
    # sampler's throughput flattened on the hosts measured in rounds 1-3, tools/cpu_scaling.py)
    # AND every host core; a concurrent query of B = 131072 holds ~1.5 GB of result vectors,
    # so the concurrent-queries cell is also capped by the RAM left beside the graph.
    try:
        import psutil as _ps
        room = int(_ps.virtual_memory().available * 0.5 / 1.5e9)
    except Exception:
        room = 32
    cells = {}
    roots_by_b = {}
    for B in (1024, 131072):
        nb = 64 if B == 1024 else 8
        roots = rng.integers(1, n + 1, B * nb).astype(np.uint64)
        roots_by_b[B] = roots
        big = B > 4096
        wu, timed = (5, 30) if (full or not big) else (1, 5)
        cell = {}
        # the as-shipped cell (8 concurrent single-threaded queries) runs SURVEY 8(d)'s whole
        # protocol at both batch sizes: 5 warm-up + 30 timed rounds (1.4 s a round at B = 131072)
        secs, e = R.bench_fanout_dag(GRAPH_SEED, roots, B, FANOUT, shipped_threads, 0, True, 5, 30)
        cell["as_shipped"] = dict(_stats(secs, e), threads=shipped_threads,
                                  what="%d concurrent single-threaded queries" % shipped_threads)
        cands = []
        # (B = 131072 with EVERY core as a concurrent query - 256 here - is a 9.2 G-edge round of
        # ~30 s and was measured SLOWER than 32: 301-323 M edges/s in profiles/r4_v1_bench.json and
        # r4_v9_bench.json; the default run stops at 64 concurrent queries to stay within minutes,
        # --cpu-protocol full runs every core)
        conc = sorted({min(32, cores), (min(cores, max(32, room)) if full else min(cores, 64, max(32, room)))
                       if big else cores})
        for many in conc:
            if many <= shipped_threads:
                continue
            # (a round of `many` concurrent B = 131072 queries takes ~30 s at 256 threads: two rounds)
            few = big and not full and many > 32
            secs, e = R.bench_fanout_dag(GRAPH_SEED, roots, B, FANOUT, many, 0, True,
                                         (wu if not big else 1) if not few else 0,
                                         (timed if (not big or full) else 3) if not few else 2)
            cands.append(dict(_stats(secs, e), threads=many,
                              what="%d concurrent single-threaded queries" % many))
        for many in sorted({min(32, cores), cores}):
            secs, e = R.bench_fanout_dag(GRAPH_SEED, roots, B, FANOUT, many, 1, True,
                                         wu if not big else max(wu, 2), timed if not big else max(timed, 10))
            cands.append(dict(_stats(secs, e), threads=many,
                              what="-DOPENMP batch loop, %d threads, one query at a time" % many))
        cell["best"] = max(cands + [cell["as_shipped"]], key=lambda c: c["edges_per_s"])
        cell["other"] = [c for c in cands if c is not cell["best"]]
        cells[B] = cell
    del R
    # ---- the GPU on the SAME graph, roots and batch sizes (call ids as the harness':
    # query q, hop h -> call_id 2 q + h), per-step times from HIP events on the stream
    same = {}
    try:
        import euler_amd
        Gs = euler_amd.Graph.synthetic(euler_amd.synth_params(GRAPH_SEED, n, 10 * n, weighted=True))
        Gs.set_seed(GRAPH_SEED)
        for B, roots in roots_by_b.items():
            r = torch.as_tensor(roots.astype(np.int64)).cuda().reshape(-1, B)
            nb = r.shape[0]
            for i in range(5):
                Gs.sample_fanout(r[i % nb], [[0], [0]], FANOUT, n + 1, call_id=2 * i)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(31)]
            torch.cuda.synchronize()
            ev[0].record()
            for i in range(30):
                Gs.sample_fanout(r[i % nb], [[0], [0]], FANOUT, n + 1, call_id=2 * i)
                ev[i + 1].record()
            torch.cuda.synchronize()
            secs = [ev[i].elapsed_time(ev[i + 1]) * 1e-3 for i in range(30)]
            g = _stats(secs, B * (FANOUT[0] + FANOUT[0] * FANOUT[1]))
            g["ratio_to_cpu_as_shipped"] = g["edges_per_s"] / cells[B]["as_shipped"]["edges_per_s"]
            g["ratio_to_cpu_best"] = g["edges_per_s"] / cells[B]["best"]["edges_per_s"]
            same[B] = g
        del Gs
    except Exception as e:                 # the baseline itself must not fail the bench
        same = {"error": str(e)}
    head = cells[131072]["best"]
    return {"value": head["edges_per_s"], "unit": "sampled edges/s", "cores": head["threads"],
            "kind": "reference", "host_cores": cores,
            "graph": "synthetic power-law graph of the metric's family, %s, %d edges built "
                     "(reference Node objects, %.1f s with %d threads)" % (fit_note, n_edges, build_s,
                                                                           build_threads),
            "protocol": "SURVEY 8(d): ID_UNIQUE -> API_SAMPLE_NB -> DATA_GATHER per hop, same graph / "
                        "roots / batch on CPU and GPU; 5 warm-up + 30 timed rounds, median (p10, p90)"
                        + ("" if full else "; B = 131072: as shipped 5 + 30 rounds, the more-threads cells "
                           "1 + 3 (concurrent queries) / 2 + 10 (OpenMP) - pass --cpu-protocol full for "
                           "5 + 30 everywhere"),
            "sample_node": side_cells.get("sample_node"), "deepwalk": side_cells.get("deepwalk"),
            "B1024": {"cpu": cells[1024], "gpu_same_graph": same.get(1024, same)},
            "B131072": {"cpu": cells[131072], "gpu_same_graph": same.get(131072, same)},
            "sample": "value = best CPU configuration at B = 131072 (%s); as shipped (8 query "
                      "threads): %.3g edges/s; B = 1024: as shipped %.3g, best %.3g edges/s.  `cores` is "
                      "where the REFERENCE is fastest on this host, not a handicap: it stops scaling "
                      "beyond that (every configuration tried at B = 131072: %s)"
                      % (head["what"], cells[131072]["as_shipped"]["edges_per_s"],
                         cells[1024]["as_shipped"]["edges_per_s"], cells[1024]["best"]["edges_per_s"],
                         "; ".join("%s: %.3g edges/s" % (c["what"], c["edges_per_s"])
                                   for c in [cells[131072]["best"]] + cells[131072]["other"]))}


def _cpu_node_and_walk_cells(R, n, cores):
    """CPU cells of SampleNode (K2) and DeepWalk on the reference graph `R` (oracle/_ref: the
    reference's own Graph::SampleNode / Node::SampleNeighbor behind the RNG seam), as the
    client runs them: 8 concurrent single-threaded queries (client/query_proxy.cc:205-210) and
    32.  SampleNode: 1M draws per query, type -1 (4 draws per sample, graph.cc:229-236).
    DeepWalk: 16 384 walkers x 40 steps per query (random_walk_op.cc:207-247)."""
    out = {}
    cnt = 1 << 20
    try:
        cells = [_threaded_rate(lambda t_, r_: R.sample_node(GRAPH_SEED, 1000 + 64 * r_ + t_, [-1], cnt),
                                th, cnt) for th in sorted({min(8, cores), min(32, cores)})]
        best = max(cells, key=lambda c: c["per_s"])
        out["sample_node"] = {"value": best["per_s"], "unit": "sampled nodes/s", "cores": best["threads"],
                              "kind": "reference", "as_shipped_8_queries": cells[0]["per_s"],
                              "cells": cells,
                              "sample": "Graph::SampleNode(type -1), %d draws per query, alias tables over "
                                        "%d nodes" % (cnt, n)}
    except Exception as e:
        out["sample_node"] = {"error": repr(e)}
    try:
        out["deepwalk"] = _walk_cell(R, n, cores)
    except Exception as e:
        out["deepwalk"] = {"error": repr(e)}
    return out


def _n2v_cell(R, n, cores):
    """node2vec (p = 0.25, q = 4) on the reference graph R: 8 and 32 concurrent queries of 2 048
    walkers x 10 steps (tf_euler/kernels/random_walk_op.cc:83-168: the client's loop over
    GetFullNeighbor + BuildWeights, reference sources behind the RNG seam)."""
    W, LEN = 2048, 10
    rng = np.random.default_rng(6)
    starts = rng.integers(1, n + 1, (32, W)).astype(np.int64)
    et = [[0]] * LEN
    cells = [_threaded_rate(lambda t_, r_: R.random_walk(GRAPH_SEED, 10 * (64 * r_ + t_), starts[t_], et, LEN,
                                                         0.25, 4.0, n + 1), th, W * LEN, rounds=2)
             for th in sorted({min(8, cores), min(32, cores)})]
    best = max(cells, key=lambda c: c["per_s"])
    return {"value": best["per_s"], "unit": "walker steps/s", "cores": best["threads"], "kind": "reference",
            "as_shipped_8_queries": cells[0]["per_s"], "cells": cells,
            "sample": "random_walk p = 0.25, q = 4 (reference sources behind the RNG seam), %d walkers x %d "
                      "steps per query, %d-node graph" % (W, LEN, n)}


def cpu_walk_cell(args, n2v=False):
    """cpu_baseline of `--workload deepwalk`: the reference's walk (oracle/_ref) on a bounded
    graph of the metric's family (5M nodes / 50M edges: ~10 s to build with 32 threads).
    n2v: also the node2vec cell (key "node2vec")."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    if not O.have_ref():
        return {"value": None, "unit": "walker steps/s", "cores": cores, "kind": "reference",
                "error": "oracle/_ref/libeuler_ref.so missing"}
    n = min(5_000_000, args.nodes)
    R, _ne, build_s = _ref_graph(n, 1, min(32, cores), False)
    cell = _walk_cell(R, n, cores)
    if n2v:
        try:
            cell["node2vec"] = _n2v_cell(R, n, cores)
        except Exception as e:
            cell["node2vec"] = {"error": repr(e)}
    del R
    cell["host_cores"] = cores
    cell["sample"] += "; %d-edge graph of the metric's family built in %.1f s" % (_ne, build_s)
    return cell


def _walk_cell(R, n, cores):
    """DeepWalk on the reference graph R: 8 and 32 concurrent queries of 16 384 walkers x 40
    steps (tf_euler/kernels/random_walk_op.cc:207-247 over the reference's Node::SampleNeighbor)."""
    W, LEN = 16384, 40
    rng = np.random.default_rng(5)
    starts = rng.integers(1, n + 1, (32, W)).astype(np.int64)
    et = [[0]] * LEN
    cells = [_threaded_rate(lambda t_, r_: R.random_walk(GRAPH_SEED, 40 * (64 * r_ + t_), starts[t_], et, LEN,
                                                         1.0, 1.0, n + 1), th, W * LEN)
             for th in sorted({min(8, cores), min(32, cores)})]
    best = max(cells, key=lambda c: c["per_s"])
    return {"value": best["per_s"], "unit": "walker steps/s", "cores": best["threads"], "kind": "reference",
            "as_shipped_8_queries": cells[0]["per_s"], "cells": cells,
            "sample": "random_walk p = q = 1 (reference sources behind the RNG seam), %d walkers x %d steps "
                      "per query, %d-node graph" % (W, LEN, n)}


def _ref_graph(n, n_types, threads, build_sampler):
    """The reference's Graph (oracle/_ref) over the synthetic graph of n nodes / 10 n edges."""
    from oracle import oracle as O
    t0 = time.time()
    po = O.synth_params(GRAPH_SEED, n, 10 * n, n_types=n_types, weighted=True)
    csr = O.synth_csr(po, threads=threads)
    w = csr.prefix_w.copy()
    w[1:] -= csr.prefix_w[:-1]
    starts = csr.row_ptr[:-1]
    w[starts] = csr.prefix_w[starts]
    n_edges = int(len(csr.nbr))
    if n_types == 1:
        seg_ptr = csr.row_ptr
    else:           # build_raw takes one segment per (row, edge type)
        te = csr.type_end.reshape(n, n_types).astype(np.int64)
        seg_ptr = np.concatenate([[0], (csr.row_ptr[:-1, None] + te).reshape(-1)]).astype(np.int64)
    R = O.RefGraph.build_raw(csr.row_id, seg_ptr, csr.nbr, w, n_types, threads=threads,
                             build_sampler=build_sampler)
    return R, n_edges, time.time() - t0


def cpu_hetero_cell(args, type_sets, cnt, D):
    """cpu_baseline of the heterogeneous step: typed SampleNeighbor by the reference
    (oracle/_ref, Node::SampleNeighbor with k = 1 / 3 of 8 / all) + gather + scatter_mean by
    the oracle's restatement of tf_euler/kernels/{gather,scatter}_op.cc, one query = 8 192
    roots through the three type sets, 8 and 32 concurrent queries; a bounded graph (2M
    nodes / 20M edges, 8 edge types)."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    if not O.have_ref():
        return {"value": None, "unit": "sampled edges/s", "cores": cores, "kind": "reference",
                "error": "oracle/_ref/libeuler_ref.so missing"}
    n = min(2_000_000, args.nodes)
    R, n_edges, build_s = _ref_graph(n, 8, min(32, cores), False)
    B = 8192
    rng = np.random.default_rng(9)
    roots = rng.integers(1, n + 1, (32, B)).astype(np.uint64)
    feat = rng.standard_normal((n + 2, D), dtype=np.float32)
    dst = np.repeat(np.arange(B, dtype=np.int32), cnt)

    def query(t_, r_):
        for c, et in enumerate(type_sets):
            idx_, ids, _w, _t = R.sample_neighbor_core(GRAPH_SEED, 3 * (64 * r_ + t_) + c, roots[t_], et, cnt)
            # (core layout: a node without such edges has an empty row)
            lens = (idx_[:, 1] - idx_[:, 0]).astype(np.int64)
            m_ = int(lens.sum())
            d_ = dst if m_ == B * cnt else np.repeat(np.arange(B, dtype=np.int32), lens)
            O.scatter_mean(O.gather(feat, ids[:m_].astype(np.int32)), d_, B)
    cells = [_threaded_rate(query, th, B * cnt * len(type_sets)) for th in sorted({min(8, cores), min(32, cores)})]
    best = max(cells, key=lambda c: c["per_s"])
    del R
    return {"value": best["per_s"], "unit": "sampled edges/s", "cores": best["threads"], "kind": "reference",
            "host_cores": cores, "as_shipped_8_queries": cells[0]["per_s"], "cells": cells,
            "sample": "typed SampleNeighbor (reference sources, k = 1 / 3 of 8 / all, count %d) + gather + "
                      "scatter_mean (oracle's restatement of gather_op.cc / scatter_op.cc, D = %d), %d roots per "
                      "query, %d-node / %d-edge graph with 8 edge types (built in %.1f s)"
                      % (cnt, D, B, n, n_edges, build_s)}
