#!/usr/bin/env python3
"""Benchmark of the sampling hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
                  [--nodes 100000000] [--edges 1000000000]

Metric (BASELINE.json): sampled edges/sec, whole node, 2-hop fanout [25,10] on
a 100M-node / 1B-edge weighted power-law graph resident in HBM.  One "step" =
one minibatch of B roots through SampleFanout([25,10]) = B*275 sampled edges,
roots already in HBM.  With --gpus N > 1 (launched by torch.distributed.run,
one rank per GPU) the graph is hash-partitioned over the ranks (owner(id) =
id % N) and every hop does the id / result all-to-all over RCCL; every rank
samples its own B roots per step (weak scaling).

Rank 0 prints ONE JSON line: metric/value/... plus
  roofline      SampleNeighborPivotKernel (both launches of a step) timed with HIP
                events on its own stream; achieved = algorithmic bytes / time
  cpu_baseline  the reference sampler (oracle/_ref: reference sources + RNG
                seam) timed on the host cores on a bounded sample of the same
                workload family
"""
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

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FANOUT = [25, 10]
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
GRAPH_SEED = 20240521
PREWARM = 16                   # sharded path: extra untimed steps (allocator warm-up)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=131072,
                    help="roots per step per GPU")
    ap.add_argument("--nodes", type=int, default=100_000_000)
    ap.add_argument("--edges", type=int, default=1_000_000_000)
    ap.add_argument("--cpu-nodes", type=int, default=20_000_000,
                    help="graph size of the CPU baseline (SURVEY 8(d): 20M nodes / 200M "
                         "edges; reduced automatically when the host's RAM does not hold it)")
    ap.add_argument("--cpu-protocol", choices=["quick", "full"], default="quick",
                    help="full = 5 warm-up + 30 timed rounds for every cell (minutes)")
    ap.add_argument("--streams", type=int, default=3,
                    help="unsharded path: consecutive minibatches alternate between this many "
                         "HIP streams (each with its own scratch): the latency-bound phases of "
                         "one minibatch (first hop, duplicate detection, sampling of the "
                         "distinct roots) overlap the bandwidth-bound expansion of the other - "
                         "the reference likewise keeps 8 queries in flight "
                         "(client/query_proxy.cc:205-210).  Same box, 20 steps x 5: 1 stream 0.229 ms per "
                         "step, 2: 0.212-0.223, 3: 0.206 (and the steadiest), 4: 0.225")
    ap.add_argument("--workload", choices=["metric", "products", "hetero", "deepwalk"],
                    default="metric",
                    help="metric = BASELINE.json's headline (configs[2]); products = configs[1] "
                         "(ogbn-products-shaped CSR, uniform SampleNeighbor fanout [25,10]); "
                         "hetero = configs[4] on one GPU (8 edge types: typed sampling k = 1 / "
                         "3 of 8 / all + 128-d feature gather + scatter_mean); deepwalk = "
                         "configs[3] on one GPU (random_walk length 40)")
    ap.add_argument("--tuning", default="",
                    help="A/B only: comma-separated key=value pairs for euler_gpu_set_tuning")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the K-step timed loop is repeated this many times; the line "
                         "reports the median repetition (and lists all)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--n2v", action="store_true", help="deepwalk workload: also time node2vec")
    ap.add_argument("--unfused-aggregation", action="store_true",
                    help="hetero workload: gather, then scatter_mean (two passes over the E x D block)")
    ap.add_argument("--hetero-separate", action="store_true",
                    help="hetero workload: one sample_neighbor + one aggregation op per edge-type set "
                         "(round 3's step) instead of the one-enqueue form")
    ap.add_argument("--check-roots", type=int, default=4096,
                    help="roots of the last timed step compared with the CPU oracle after the timed region")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--sustain-steps", type=int, default=4000,
                    help="metric workload, one GPU: after the K timed steps, this many more steps "
                         "(>= 1 s of device time) are timed as ONE region and reported as "
                         "config.sustained - the K-step median is a ~5 ms burst (0 = skip)")
    ap.add_argument("--large-batch", type=int, default=1048576,
                    help="roots of the second roofline launch (SURVEY 8(d) config 3: B >= 1M per "
                         "launch for the HBM-roofline run; 0 = skip)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short products / hetero / deepwalk (+ node2vec) legs the default "
                         "single-GPU metric run appends under config.secondary")
    ap.add_argument("--no-small-batch", action="store_true",
                    help="skip the B = 1 024 latency leg (profiling runs: its 1 000 small "
                         "launches would share kernel names with the step's)")
    ap.add_argument("--pipeline", type=int, default=3,
                    help="sharded path: minibatches in flight, interleaved hop by hop "
                         "from one host thread (each on its own sampler and HIP "
                         "stream): while the host waits for one batch's bucket sizes "
                         "the GPU runs the others' kernels and exchanges (one rank: 0.61 ms "
                         "/ step with 1, 0.47 with 2, 0.46 with 3 or 4)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="functional check only: when fewer GPUs are visible than --gpus, "
                         "let the ranks share them (rank r on GPU r %% visible) with a "
                         "gloo host-staged transport (RCCL refuses two ranks per device); "
                         "the line then says so and is not a scaling number")
    ap.add_argument("--replicas", action="store_true",
                    help="--gpus N: every rank holds the WHOLE graph (68.6 GB of 288) and runs the "
                         "unsharded step on its own roots - no exchange (SURVEY H7); the default "
                         "--gpus N run is hash-sharded and reports this mode beside it "
                         "(config.replicas)")
    ap.add_argument("--no-replicas-leg", action="store_true",
                    help="--gpus N: skip the replicated-graph leg of the sharded run")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU sampler (unique / split / all-to-all / "
                         "merge / gather) even on one rank: measures its overhead")
    return ap.parse_args()


_DEFERRED = None        # a list while a process group is up: the line is printed after its teardown


def _emit(line, now=False):
    """Print the JSON line LAST: RCCL prints its version banner through C stdio (fully buffered
    when stdout is not a terminal, flushed at exit or at the communicator's teardown): runs with
    a process group defer the line until the group is gone, and C stdio is flushed first."""
    if _DEFERRED is not None and not now:
        _DEFERRED.append(line)
        return
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(line), flush=True)


def _teardown(group_up):
    """Barrier + destroy the process group, then the deferred line(s)."""
    global _DEFERRED
    if group_up:
        dist.barrier()
        dist.destroy_process_group()
    pending, _DEFERRED = _DEFERRED or [], None
    for line in pending:
        _emit(line, now=True)


def _stats(secs, edges_per_round):
    """median / p10 / p90 of per-round wall times -> edges/s (p10 of the time is
    the p90 of the rate)."""
    secs = np.sort(np.asarray(secs, np.float64))
    med = float(np.median(secs))
    return {"edges_per_s": edges_per_round / med,
            "p10_edges_per_s": edges_per_round / float(np.percentile(secs, 90)),
            "p90_edges_per_s": edges_per_round / float(np.percentile(secs, 10)),
            "median_ms_per_round": med * 1e3, "rounds": int(len(secs)),
            "edges_per_round": int(edges_per_round)}


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


def _threaded_rate(fn, threads, units_per_call, rounds=3):
    """`threads` host threads each run fn(thread, round) once per round (ctypes releases the
    GIL inside the reference's code): median units/s over the rounds after one warm-up."""
    from concurrent.futures import ThreadPoolExecutor
    secs = []
    with ThreadPoolExecutor(threads) as ex:
        for rnd in range(rounds + 1):
            t0 = time.perf_counter()
            list(ex.map(lambda t_: fn(t_, rnd), range(threads)))
            if rnd:
                secs.append(time.perf_counter() - t0)
    med = float(np.median(secs))
    return {"per_s": threads * units_per_call / med, "threads": threads, "rounds": len(secs),
            "median_ms_per_round": round(med * 1e3, 3), "units_per_round": int(threads * units_per_call)}


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


def latency_small_batch(G, L, _lib, n_nodes, default_node, batch=1024, iters=300, streams=8):
    """B = 1 024 (SURVEY 8's latency configuration): microseconds per minibatch of
    the 2-hop fanout, euler_gpu_sample_fanout called back to back on ONE stream with
    preallocated outputs (no Python allocation in the loop), and the throughput with
    `streams` minibatches in flight (the reference keeps 8 queries in flight,
    client/query_proxy.cc:205-210)."""
    dev = G.device
    layers = len(FANOUT)
    cnt_a = (C.c_int32 * layers)(*FANOUT)
    et_a = (C.c_int32 * layers)(*([0] * layers))
    gen = torch.Generator(device=dev); gen.manual_seed(77)
    roots = torch.randint(1, n_nodes + 1, (64, batch), generator=gen, device=dev, dtype=torch.int64)
    wsz = int(L.euler_gpu_sample_fanout_workspace(batch, cnt_a, layers))

    def buffers():
        o_n, o_w, o_t, m = [], [], [], batch
        for c in FANOUT:
            m *= c
            o_n.append(torch.empty(m, dtype=torch.int64, device=dev))
            o_w.append(torch.empty(m, dtype=torch.float32, device=dev))
            o_t.append(torch.empty(m, dtype=torch.int32, device=dev))
        ws = torch.empty(max(wsz, 16), dtype=torch.uint8, device=dev)
        return (o_n, o_w, o_t, ws, (C.c_void_p * layers)(*[t.data_ptr() for t in o_n]),
                (C.c_void_p * layers)(*[t.data_ptr() for t in o_w]),
                (C.c_void_p * layers)(*[t.data_ptr() for t in o_t]))

    def call(bufs, st, i):
        _lib.check(L.euler_gpu_sample_fanout(
            G._h, st, GRAPH_SEED, 2 * i, C.c_void_p(roots[i % 64].data_ptr()), batch, et_a, 1,
            cnt_a, layers, default_node, bufs[4], bufs[5], bufs[6], C.c_void_p(bufs[3].data_ptr())))

    b0 = buffers()
    st0 = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for i in range(50):
        call(b0, st0, i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        call(b0, st0, i)
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / iters
    side = [torch.cuda.Stream(device=dev) for _ in range(streams)]
    bufs = [buffers() for _ in range(streams)]
    sts = [C.c_void_p(s_.cuda_stream) for s_ in side]
    torch.cuda.synchronize()
    for i in range(4 * streams):
        call(bufs[i % streams], sts[i % streams], i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters * 2):
        call(bufs[i % streams], sts[i % streams], i)
    torch.cuda.synchronize()
    many = (time.perf_counter() - t0) / (iters * 2)
    e = batch * (FANOUT[0] + FANOUT[0] * FANOUT[1])
    # M = 64 such minibatches in ONE enqueue (euler_gpu_sample_fanout_multi), one host thread, one
    # stream: minibatch b draws with call id c0 + 2 b - the results of 64 calls, bit for bit
    M = 64
    multi = {}
    try:
        cnt_m = (C.c_int32 * layers)(*FANOUT)
        wsm = int(L.euler_gpu_sample_fanout_workspace(M * batch, cnt_m, layers))
        mo_n, mo_w, mo_t, m_ = [], [], [], M * batch
        for c in FANOUT:
            m_ *= c
            mo_n.append(torch.empty(m_, dtype=torch.int64, device=dev))
            mo_w.append(torch.empty(m_, dtype=torch.float32, device=dev))
            mo_t.append(torch.empty(m_, dtype=torch.int32, device=dev))
        mws = torch.empty(max(wsm, 16), dtype=torch.uint8, device=dev)
        mpn = (C.c_void_p * layers)(*[t.data_ptr() for t in mo_n])
        mpw = (C.c_void_p * layers)(*[t.data_ptr() for t in mo_w])
        mpt = (C.c_void_p * layers)(*[t.data_ptr() for t in mo_t])
        mroots = roots[:M].contiguous()

        def call_multi(c0):
            _lib.check(L.euler_gpu_sample_fanout_multi(
                G._h, st0, GRAPH_SEED, c0, layers, None, M, C.c_void_p(mroots.data_ptr()), batch, et_a, 1,
                cnt_m, layers, default_node, mpn, mpw, mpt, C.c_void_p(mws.data_ptr())))
        for i in range(5):
            call_multi(0)
        torch.cuda.synchronize()
        it_m = 40
        t0 = time.perf_counter()
        for i in range(it_m):
            call_multi(2 * M * i)
        torch.cuda.synchronize()
        per_call = (time.perf_counter() - t0) / it_m
        # == the separate calls (first, a middle and the last minibatch of the last launch)
        c_last = 2 * M * (it_m - 1)
        for b_ in (0, 31, M - 1):
            call(b0, st0, 0)           # placeholder buffers; the call below rewrites them
            _lib.check(L.euler_gpu_sample_fanout(
                G._h, st0, GRAPH_SEED, c_last + 2 * b_, C.c_void_p(mroots[b_].data_ptr()), batch, et_a, 1,
                cnt_a, layers, default_node, b0[4], b0[5], b0[6], C.c_void_p(b0[3].data_ptr())))
            torch.cuda.synchronize()
            per = batch
            for h, c in enumerate(FANOUT):
                per *= c
                assert torch.equal(mo_n[h][b_ * per:(b_ + 1) * per], b0[0][h]), "multi != separate calls"
                assert torch.equal(mo_w[h][b_ * per:(b_ + 1) * per], b0[1][h])
        multi = {"multi_M": M, "multi_us_per_launch": round(per_call * 1e6, 2),
                 "multi_us_per_minibatch": round(per_call * 1e6 / M, 3),
                 "edges_per_s_multi": e * M / per_call,
                 "multi_checked": "3 of the 64 minibatches == separate euler_gpu_sample_fanout calls"}
        del mo_n, mo_w, mo_t, mws
    except Exception as ex:
        multi = {"multi_error": repr(ex)}
    # the same minibatch through the Python surface (Graph.sample_fanout: allocates its outputs)
    et_l = [[0]] * layers
    for i in range(100):
        G.sample_fanout(roots[i % 64], et_l, FANOUT, default_node, call_id=2 * i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters * 2):
        G.sample_fanout(roots[i % 64], et_l, FANOUT, default_node, call_id=2 * i)
    torch.cuda.synchronize()
    py = (time.perf_counter() - t0) / (iters * 2)
    return {"latency_B1024_us": round(one * 1e6, 2), "edges_per_s_one_stream": e / one,
            "us_per_minibatch_%d_streams" % streams: round(many * 1e6, 2),
            "edges_per_s_%d_streams" % streams: e / many,
            "us_per_minibatch_python_surface": round(py * 1e6, 2), **multi}


def _events(fn, iters):
    """mean milliseconds of fn() over `iters` runs, HIP events on the current stream"""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def _oracle_rows(G, p, need, n_types, spot=64):
    """OracleGraph over the rows of the ids `need`, exported from HBM; `spot` of them are
    first compared with the HOST generator (oracle/eo_synth.c), so that a device generator
    fault cannot hide behind "oracle fed with exported rows"."""
    from oracle import oracle as O
    need = np.unique(np.asarray(need).astype(np.int64).view(np.uint64))
    rp, te, nb, pw, tp = G.export_rows(need)
    po = O.SynthParams()
    for f_, _t in po._fields_:
        setattr(po, f_, getattr(p, f_))
    T = n_types
    for j_ in np.random.default_rng(1).choice(len(need), min(spot, len(need)), replace=False):
        x_ = O.synth_internal_id(po, int(need[j_]))       # the id itself unless hashed_ids
        b_, e_ = int(rp[j_]), int(rp[j_ + 1])
        if not 1 <= x_ <= po.n_nodes:
            assert b_ == e_, "a row for an id outside the graph"
            continue
        h_ = O.synth_csr(po, x_ - 1, x_)
        assert int(h_.row_id[0]) == int(need[j_])
        assert np.array_equal(h_.nbr, nb[b_:e_]) and np.array_equal(h_.prefix_w, pw[b_:e_]), \
            "device generator differs from the host generator at node %d" % int(need[j_])
    return O.OracleGraph(O.CSR(need, rp, te, nb, pw, tp, T))


def _mix64_t(z):
    """oracle/eo_synth.c's sy_mix64 on an int64 tensor (two's-complement wrap = u64 arithmetic):
    the external ids of a hashed_ids graph, computed where the roots live."""
    def lsr(v, sft):
        return (v >> sft) & ((1 << (64 - sft)) - 1)
    z = z ^ lsr(z, 30)
    z = z * (0xbf58476d1ce4e5b9 - (1 << 64))
    z = z ^ lsr(z, 27)
    z = z * (0x94d049bb133111eb - (1 << 64))
    return z ^ lsr(z, 31)


def _oracle_sage_blocks(OG, seed, call, roots, metapath, fanouts, default):
    """SageDataFlow as the reference composes it (dataflow/sage_dataflow.py:35-50 over
    neighbor_dataflow.py:84-110) on the ORACLE: sample_neighbor of the unique frontier per
    hop, tf.unique = first-occurrence ID_UNIQUE, res_n_id / edge_index arithmetic."""
    from oracle import oracle as O

    def uniq(a):
        uq, gi = O.id_unique(a.astype(np.uint64))
        return uq.astype(np.int64), gi.astype(np.int64)
    n_id = roots.copy()
    nbrs, srcs = [], []
    for h, (et, c) in enumerate(zip(metapath, fanouts)):
        nb, _, _ = OG.sample_neighbor(seed, call + h, n_id, et, c, default)
        nbrs.append(nb.reshape(-1))
        srcs.append(np.repeat(np.arange(len(n_id)), c))
        n_id, _ = uniq(np.concatenate([nb.reshape(-1), n_id]))
    n_id = roots.copy()
    last_idx = np.arange(len(n_id))
    want = []
    for i in range(len(fanouts)):
        new_n_id, inv = uniq(np.concatenate([nbrs[i], n_id]))
        res = inv[-len(n_id):]
        src = np.concatenate([srcs[i], last_idx])
        last_idx = np.arange(len(new_n_id))
        want.append((new_n_id, res, np.stack([src, inv])))
        n_id = new_n_id
    return want


def _rank_ctx():
    """(rank, world, wire device) of the sharded secondary workloads; one process: (0, 1, None)"""
    if dist.is_available() and dist.is_initialized():
        on_gpu = dist.get_backend() == "nccl"
        return dist.get_rank(), dist.get_world_size(), (torch.device("cuda", torch.cuda.current_device())
                                                        if on_gpu else torch.device("cpu"))
    return 0, 1, None


def _max_over_ranks(secs, wire):
    if wire is None:
        return secs
    t = torch.tensor(secs, device=wire, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t.tolist()]


def _sync_ranks(wire):
    if wire is not None:
        dist.barrier()
    torch.cuda.synchronize()


def run_hetero(args, quiet=False):
    """configs[4] on one GPU: heterogeneous graph (8 edge types), per-type neighbour
    sampling with one listed type, 3 of 8 (sub-collection draw) and all 8 (type draw
    over all groups), each followed by the 128-d feature gather of the sampled block
    and scatter_mean into the roots (segment reduce, fp32, order-faithful)."""
    import euler_amd
    from euler_amd import ops
    rank, world, wire = _rank_ctx()
    # "same N" as the metric graph - 100M nodes / 1B edges, 8 edge types,
    # D = 128 (51 GB of features + 36 GB of graph in one GPU's 288 GB)
    N, E_h, T, D, CNT = args.nodes, args.edges, 8, 128, 10
    B = args.batch
    t0 = time.time()
    p_h = euler_amd.synth_params(GRAPH_SEED, N, E_h, n_types=T, weighted=True)
    G = euler_amd.Graph.synthetic(p_h, device=torch.cuda.current_device(), partitions=world,
                                  shard_index=rank, shards=world)
    G.set_seed(GRAPH_SEED)
    # N ranks: the graph is hash-sharded (owner = id % world), every typed hop is one id /
    # result exchange (ShardedSampler); the feature table is replicated and the aggregation
    # local - it works on minibatch-local tensors (replicas only)
    S = None
    if wire is not None:
        from euler_amd.distributed import gpu_sharded_sampler
        S = gpu_sharded_sampler(G, partitions=world)

    def sample(r_, et_, call_id):
        if S is None:
            return G.sample_neighbor(r_, et_, CNT, N + 1, call_id=call_id)
        ids_, w_, t_, _m = S.sample_neighbor(r_, et_, CNT, N + 1, call_id)
        return ids_, w_, t_
    feat = torch.randn(N + 2, D, device="cuda", generator=torch.Generator("cuda").manual_seed(7))
    torch.cuda.synchronize()
    build_s = time.time() - t0
    n_steps = args.steps + args.warmup
    gen = torch.Generator(device="cuda"); gen.manual_seed(1234 + rank)
    roots = torch.randint(1, N + 1, (n_steps, B), generator=gen, device="cuda", dtype=torch.int64)
    dst = torch.arange(B, device="cuda", dtype=torch.int32).repeat_interleave(CNT)
    type_sets = ([3], [1, 4, 6], list(range(T)))

    fused = not args.unfused_aggregation

    one_enqueue = S is None and not args.unfused_aggregation and not args.hetero_separate

    def step(i):
        if one_enqueue:
            # the three typed draws of the minibatch as ONE launch and their aggregation as one
            # pass, enqueued by one C call (euler_gpu_sample_aggregate_sets): the results of the
            # three sample_neighbor + gather_segment_reduce pairs below, bit for bit
            return G.sample_neighbor_sets(roots[i], type_sets, CNT, N + 1, call_id=3 * i, feat=feat)[3]
        aggs = []
        if S is not None and not args.hetero_separate:
            # sharded: one front end / host wait / id exchange for the three sets over the same roots
            outs = S.sample_neighbor_sets(roots[i], type_sets, CNT, N + 1, call_id=3 * i)
            return [ops.gather_segment_reduce("mean", feat, o[0].reshape(-1), B, count=CNT) for o in outs]
        for c, et in enumerate(type_sets):
            nb, _w, _t = sample(roots[i], et, 3 * i + c)
            if fused:      # the rows are reduced as they are read, CNT per root (the sampler's int64
                           # ids are the indices: euler_gpu_gather_segment_reduce_ids)
                aggs.append(ops.gather_segment_reduce("mean", feat, nb.reshape(-1), B, count=CNT))
            else:
                src = nb.reshape(-1).to(torch.int32)
                aggs.append(ops.scatter_mean(ops.gather(feat, src), dst, B))
        return aggs

    # consecutive minibatches alternate between --streams HIP streams (default 2), as in the
    # headline workload: the latency-bound sampling of one overlaps the aggregation of another
    n_streams = max(1, args.streams) if S is None else 1     # a sharded hop waits on the host
    side = [torch.cuda.Stream() for _ in range(n_streams)] if n_streams > 1 else None

    def loop(first, last):
        for i in range(first, last):
            if side is None:
                step(i)
            else:
                with torch.cuda.stream(side[i % n_streams]):
                    step(i)

    _sync_ranks(wire)
    loop(0, max(args.warmup, 2 * n_streams))
    _sync_ranks(wire)
    reps = []
    for _rep in range(max(1, args.repeats)):
        _sync_ranks(wire)
        t0 = time.perf_counter()
        loop(args.warmup, n_steps)
        _sync_ranks(wire)
        reps.append(time.perf_counter() - t0)
    reps = _max_over_ranks(reps, wire)           # slowest rank, per repetition
    elapsed = float(np.median(reps))
    if S is not None:
        edges = B * CNT * len(type_sets) * world
        # the sharded step's own launches: the owners' pass of each typed hop
        # (euler_gpu_sample_neighbor_packed over the distinct ids asked for), timed alone
        from euler_amd import _lib as _lb
        st_s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        x_own = roots[n_steps - 1]
        if world > 1:
            x_own = torch.clamp((x_own // world) * world + (rank if rank else world), max=N - world)
        x_own = torch.unique(x_own).contiguous()
        k1s = []
        for c, et in enumerate(type_sets):
            ms_ = _events(lambda: G.sample_neighbor_packed(x_own, et, CNT, N + 1, call_id=c), 10)
            b_ = C.c_double(0)
            eta = (C.c_int32 * len(et))(*et)
            _lb.check(_lb.lib().euler_gpu_sample_neighbor_algo_bytes(
                G._h, st_s, C.c_void_p(x_own.data_ptr()), x_own.numel(), eta, len(et), CNT, C.byref(b_)))
            k1s.append({"listed_types": len(et), "ms": round(ms_, 4), "algorithmic_bytes": b_.value,
                        "frac": round(b_.value / ms_ / 1e6 / HBM_PEAK_GBS, 4)})
        tb = sum(x_["algorithmic_bytes"] for x_ in k1s)
        tm = sum(x_["ms"] for x_ in k1s)
        roof_s = {"kernel": "SampleNeighborKernel / SampleNeighborTypedPivotKernel (owners' pass of a typed "
                            "sharded hop, packed wire rows)", "bound": "hbm",
                  "achieved": round(tb / tm / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": round(tb / tm / 1e6 / HBM_PEAK_GBS, 4), "traffic": None,
                  "algorithmic_bytes_per_launch": tb / len(k1s), "avg_launch_ms": round(tm / len(k1s), 4),
                  "launches": k1s,
                  "note": "rank 0's three typed launches of one step, each timed alone with HIP events over "
                          "the distinct roots; the aggregation is the unsharded path's (replicas only)"}
        cpu_s = None
        if rank == 0 and not args.no_cpu_baseline and not quiet:
            try:
                cpu_s = cpu_hetero_cell(args, type_sets, CNT, D)
            except Exception as e:
                cpu_s = {"error": repr(e)}
        line = {
            "metric": "sampled + aggregated edges/sec, typed SampleNeighbor (k = 1, 3 of 8, all) + 128-d "
                      "gather + scatter_mean, heterogeneous graph (BASELINE configs[4])",
            "value": edges * args.steps / elapsed, "unit": "sampled edges/s",
            "n_gpus": min(world, torch.cuda.device_count()),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 / f32",
            "data": "synthetic",
            "config": {"workload": "hetero, sharded: %d nodes, %d edge types, hash owner(id) = id %% %d, one "
                                   "exchange per typed hop (3 per step), %d roots per step per rank, "
                                   "features [%d, %d] f32 replicated, aggregation local"
                                   % (N, T, world, B, N + 2, D),
                       "ranks": world, "graph_build_s": round(build_s, 2), "repeats": len(reps),
                       "repeat_ms_per_step": [round(x_ / args.steps * 1e3, 4) for x_ in reps],
                       "transport": "%s, %d ranks in the communicator" % (dist.get_backend(),
                                                                           dist.get_world_size())},
            "roofline": roof_s, "cpu_baseline": cpu_s,
        }
        if rank == 0 and not quiet:
            _emit(line)
        return line
    one_stream = None
    if side is not None:                       # the same steps on ONE stream, for the record
        for i in range(args.warmup, n_steps):  # (untimed first: this stream's allocator pool is empty)
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.warmup, n_steps):
            step(i)
        torch.cuda.synchronize()
        one_stream = (time.perf_counter() - t0) / args.steps * 1e3
    edges = B * CNT * len(type_sets)
    # phase split of one step + the dominant kernel's roofline (the gather: E rows of
    # D floats read at random and written in order: 8 E D + 4 E bytes)
    r = roots[n_steps - 1]
    ph = {}
    host_us = {}
    from euler_amd import _lib as _lib0
    L0 = _lib0.lib()
    o_n = torch.empty((B, CNT), dtype=torch.int64, device="cuda")
    o_w = torch.empty((B, CNT), dtype=torch.float32, device="cuda")
    o_t = torch.empty((B, CNT), dtype=torch.int32, device="cuda")
    for c, et in enumerate(type_sets):
        # the launch alone, enqueued back to back from C between two HIP events on its stream
        # (a Python call of the op costs the host more than the kernel runs: see host_us_per_call)
        ms_c = C.c_float(0)
        eta0 = (C.c_int32 * len(et))(*et)
        _lib0.check(L0.euler_gpu_time_sample_neighbor(
            G._h, C.c_void_p(torch.cuda.current_stream().cuda_stream), GRAPH_SEED,
            C.c_void_p(r.data_ptr()), B, eta0, len(et), CNT, _lib0.LAYOUT_TF,
            C.c_void_p(o_n.data_ptr()), C.c_void_p(o_w.data_ptr()), C.c_void_p(o_t.data_ptr()), 20,
            C.byref(ms_c)))
        ph["sample k=%d" % len(et)] = float(ms_c.value)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            G.sample_neighbor(r, et, CNT, N + 1, call_id=c)
        host_us["sample k=%d" % len(et)] = (time.perf_counter() - t0) / 50 * 1e6     # enqueue only
        torch.cuda.synchronize()
    nb = G.sample_neighbor(r, [3], CNT, N + 1, call_id=0)[0].reshape(-1).to(torch.int32)
    g_ms = _events(lambda: ops.gather(feat, nb), 10)
    x = ops.gather(feat, nb)
    s_ms = _events(lambda: ops.scatter_mean(x, dst, B), 10)
    f_ms = _events(lambda: ops.gather_segment_reduce("mean", feat, nb, B, count=CNT), 10)
    assert torch.equal(ops.gather_segment_reduce("mean", feat, nb, B, count=CNT), ops.scatter_mean(x, dst, B))
    assert torch.equal(ops.gather_scatter("mean", feat, nb, dst, B), ops.scatter_mean(x, dst, B))
    # parity at bench scale: 64 roots of the last step, every type set, against the oracle
    # fed with the rows exported from HBM; their aggregated features against an fp64 mean
    sel = np.random.default_rng(0).choice(B, 256, replace=False)
    r_sel = r.cpu().numpy()[sel]
    need_ids = r_sel[(r_sel >= 1) & (r_sel <= N)]
    OGh = _oracle_rows(G, p_h, need_ids, T)
    checked = 0
    for c, et in enumerate(type_sets):
        nb_b, w_b, t_b = G.sample_neighbor(r, et, CNT, N + 1, call_id=900 + c)
        on, ow, ot = OGh.sample_neighbor(GRAPH_SEED, 900 + c, r_sel, et, CNT, N + 1)
        got = nb_b.reshape(B, CNT).cpu().numpy()[sel]
        assert np.array_equal(got, on.reshape(-1, CNT)), "hetero: sampled ids differ from the oracle"
        assert np.array_equal(w_b.reshape(B, CNT).cpu().numpy()[sel], ow.reshape(-1, CNT))
        assert np.array_equal(t_b.reshape(B, CNT).cpu().numpy()[sel], ot.reshape(-1, CNT))
        agg = ops.gather_segment_reduce("mean", feat, nb_b.reshape(-1).to(torch.int32), B, count=CNT)
        ref = feat[torch.as_tensor(got.reshape(-1)).cuda()].double().reshape(len(sel), CNT, D).mean(1)
        a_sel = agg[torch.as_tensor(sel).cuda()].double()
        assert torch.all((a_sel - ref).abs() <= 1e-5 * (1.0 + ref.abs())), "hetero: aggregation off"
        checked += int(got.size)
    # the one-enqueue step == the three separate ops, on the whole batch
    sn, sw, st_, sagg = G.sample_neighbor_sets(r, type_sets, CNT, N + 1, call_id=900, feat=feat)
    for c, et in enumerate(type_sets):
        nb_b, w_b, t_b = G.sample_neighbor(r, et, CNT, N + 1, call_id=900 + c)
        assert torch.equal(sn[c], nb_b) and torch.equal(sw[c], w_b) and torch.equal(st_[c], t_b), \
            "hetero: one launch over the type sets != the separate launches"
        assert torch.equal(sagg[c], ops.gather_segment_reduce("mean", feat, nb_b.reshape(-1), B, count=CNT))
    # ... and its launches alone (HIP events around 20 steps enqueued back to back)
    set_ms = _events(lambda: G.sample_neighbor_sets(r, type_sets, CNT, N + 1, call_id=5), 20)
    step_ms = _events(lambda: G.sample_neighbor_sets(r, type_sets, CNT, N + 1, call_id=5, feat=feat), 20)
    E = B * CNT
    g_bytes = 8.0 * E * D + 4.0 * E
    s_bytes = 4.0 * E * D + 4.0 * E + 4.0 * B * D
    f_bytes = 4.0 * E * D + 4.0 * E + 4.0 * B * D      # rows read once, their numbers, means written
    # the typed sampler is the largest share of the step once the aggregation is one pass:
    # its launches against SURVEY 8(d)'s byte count (euler_gpu_sample_neighbor_algo_bytes)
    from euler_amd import _lib
    Lb = _lib.lib()
    st_ = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    k1 = []
    for et in type_sets:
        b_ = C.c_double(0)
        eta = (C.c_int32 * len(et))(*et)
        _lib.check(Lb.euler_gpu_sample_neighbor_algo_bytes(
            G._h, st_, C.c_void_p(r.data_ptr()), r.numel(), eta, len(et), CNT, C.byref(b_)))
        ms_ = ph["sample k=%d" % len(et)]
        k1.append({"listed_types": len(et), "ms": round(ms_, 4), "algorithmic_bytes": b_.value,
                   "GBps": round(b_.value / ms_ / 1e6, 1)})
    k1_bytes = sum(x_["algorithmic_bytes"] for x_ in k1) / len(k1)
    k1_ms = sum(x_["ms"] for x_ in k1) / len(k1)
    sets_bytes = sum(x_["algorithmic_bytes"] for x_ in k1)
    roof = {"kernel": "SampleNeighborSetsKernel: the three typed draws of a minibatch in one launch (one "
                      "listed type; 3 of 8 and all 8: type draw + search)",
            "bound": "hbm", "achieved": round(sets_bytes / set_ms / 1e6, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(sets_bytes / set_ms / 1e6 / HBM_PEAK_GBS, 4), "traffic": None,
            "algorithmic_bytes_per_launch": sets_bytes, "avg_launch_ms": round(set_ms, 4),
            "step_kernels_ms": round(step_ms, 4),
            "separate_launches": {"avg_launch_ms": round(k1_ms, 4), "algorithmic_bytes_per_launch": k1_bytes,
                                  "frac": round(k1_bytes / k1_ms / 1e6 / HBM_PEAK_GBS, 4)},
            "launches": k1,
            "aggregation": {
                "one_pass": {"ms": round(f_ms, 4), "algorithmic_bytes": f_bytes,
                             "GBps": round(f_bytes / f_ms / 1e6, 1),
                             "note": "rows read once per EDGE by the formula; the sampled neighbours of a "
                                     "power-law graph repeat, so most of those reads are L2 / MALL hits "
                                     "and the rate can exceed the HBM peak"},
                "gather": {"ms": round(g_ms, 4), "algorithmic_bytes": g_bytes,
                           "GBps": round(g_bytes / g_ms / 1e6, 1),
                           "frac": round(g_bytes / g_ms / 1e6 / HBM_PEAK_GBS, 4)},
                "scatter_mean": {"ms": round(s_ms, 4), "algorithmic_bytes": s_bytes,
                                 "GBps": round(s_bytes / s_ms / 1e6, 1),
                                 "frac": round(s_bytes / s_ms / 1e6 / HBM_PEAK_GBS, 4)}}}
    line = {
        "metric": "sampled + aggregated edges/sec, typed SampleNeighbor (k = 1, 3 of 8, all) + 128-d "
                  "gather + scatter_mean, heterogeneous graph (BASELINE configs[4], 1 GPU)",
        "value": edges * args.steps / elapsed, "unit": "sampled edges/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 / f32",
        "data": "synthetic",
        "config": {"workload": "hetero: %d nodes / %d edges, %d edge types, weighted; %d roots per step, "
                               "3 typed hops of %d neighbours, features [%d, %d] f32"
                               % (N, G.num_edges, T, B, CNT, N + 2, D),
                   "graph_build_s": round(build_s, 2), "repeats": len(reps),
                   "repeat_ms_per_step": [round(x_ / args.steps * 1e3, 4) for x_ in reps],
                   "streams": n_streams,
                   "parity_checked_edges": checked,
                   "one_stream_ms_per_step": None if one_stream is None else round(one_stream, 4),
                   "aggregation": ("ops.gather_segment_reduce (one pass, %d rows per root)" % CNT if fused
                                   else "ops.gather + ops.scatter_mean"),
                   "step": ("one enqueue: Graph.sample_neighbor_sets(feat=...) = euler_gpu_sample_aggregate_sets"
                            if one_enqueue else "3 x (sample_neighbor + aggregation) ops"),
                   "phases_ms": dict({k_: round(v_, 4) for k_, v_ in ph.items()},
                                     gather=round(g_ms, 4), scatter_mean=round(s_ms, 4),
                                     gather_scatter_mean=round(f_ms, 4)),
                   "host_us_per_call": {k_: round(v_, 1) for k_, v_ in host_us.items()},
                   "host_note": "phases_ms of the sampling launches are kernel times (enqueued from C); "
                                "host_us_per_call = what one Python call of the op costs the host to enqueue "
                                "(a step issues 9 ops: ~0.16 ms of host time against ~0.32 ms of kernels)"},
        "roofline": roof,
        "cpu_baseline": None,
    }
    if not quiet and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_hetero_cell(args, type_sets, CNT, D)
        except Exception as e:
            line["cpu_baseline"] = {"error": repr(e)}
    if not quiet:
        _emit(line)
    del G, feat
    torch.cuda.empty_cache()
    return line


def run_deepwalk(args, G=None, p_g=None, quiet=False):
    """configs[3] on one GPU: DeepWalk, random_walk length 40 (p = q = 1) from 1M
    start nodes of the metric graph; value = walker steps / s.  --n2v also times
    node2vec (p = 0.25, q = 4) on 100 000 walkers x 10 steps."""
    import euler_amd
    from euler_amd import _lib
    L = _lib.lib()
    rank, world, wire = _rank_ctx()
    N = args.nodes
    t0 = time.time()
    if G is None:
        p_g = euler_amd.synth_params(GRAPH_SEED, N, args.edges, weighted=True)
        G = euler_amd.Graph.synthetic(p_g, device=torch.cuda.current_device(), partitions=world,
                                      shard_index=rank, shards=world)
    G.set_seed(GRAPH_SEED)
    torch.cuda.synchronize()
    build_s = time.time() - t0
    W, LEN = (1_000_000 if N >= 100_000_000 else max(1000, N // 100)), 40
    gen = torch.Generator(device="cuda"); gen.manual_seed(1234 + rank)
    n_steps = args.steps + args.warmup
    starts = torch.randint(1, N + 1, (n_steps, W), generator=gen, device="cuda", dtype=torch.int64)
    et = [[0]] * LEN
    # N ranks: the graph is hash-sharded (owner = id % world); every walk step is one id /
    # result exchange (ShardedSampler.random_walk), the node2vec run fetches the rows of the
    # walkers' nodes from their owners step by step (random_walk_op.cc:83-168)
    S = None
    if wire is not None:
        from euler_amd.distributed import gpu_sharded_sampler
        S = gpu_sharded_sampler(G, partitions=world)

    def walk(st_, et_, p_, q_, call_id):
        if S is None:
            return G.random_walk(st_, et_, p_, q_, N + 1, call_id=call_id)
        return S.random_walk(st_, et_, p_, q_, default_node=N + 1, call_id=call_id)

    for i in range(args.warmup):
        walk(starts[i], et, 1.0, 1.0, LEN * i)
    _sync_ranks(wire)
    reps = []
    for _rep in range(max(1, args.repeats)):
        _sync_ranks(wire)
        t0 = time.perf_counter()
        for i in range(args.warmup, n_steps):
            walks = walk(starts[i], et, 1.0, 1.0, LEN * i)
        _sync_ranks(wire)
        reps.append(time.perf_counter() - t0)
    reps = _max_over_ranks(reps, wire)           # slowest rank, per repetition
    elapsed = float(np.median(reps))
    if S is not None:
        # the call's own figures: host waits / level sizes (C orchestration), SURVEY 8(d)'s bytes
        # of the walk over the nodes whose rows THIS rank holds (x ranks: every rank's walkers
        # visit every shard alike), and - one rank - 64 walkers against the oracle
        walk_stats, roof_s, cpu_s, checked_s = None, None, None, None
        try:
            from euler_amd.distributed import c_sharded_random_walk
            last = n_steps - 1
            if getattr(S, "c_walk_fn", None) is not None:
                _w, walk_stats = c_sharded_random_walk(G, S.c_transport, starts[last], et, N + 1, LEN * last,
                                                       S.partitions, S.walk_cohorts, S.dense_table,
                                                       return_stats=True)
            ms_c = elapsed / args.steps * 1e3
            b_ = C.c_double(0)
            et_a = (C.c_int32 * LEN)(*([0] * LEN))
            _lib.check(L.euler_gpu_random_walk_algo_bytes(
                G._h, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(walks.data_ptr()),
                W, et_a, 1, LEN, 1.0, 1.0, C.byref(b_)))
            wb_ = b_.value * world
            roof_s = {"kernel": "WalkOwnedKernel + front end + ShWalkPathKernel (the whole "
                                "euler_gpu_sharded_random_walk call; per-step launches are microseconds)",
                      "bound": "hbm", "achieved": round(wb_ / world / (ms_c * 1e-3) / 1e9, 1),
                      "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": round(wb_ / world / (ms_c * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                      "algorithmic_bytes_per_launch": wb_ / world, "avg_launch_ms": round(ms_c, 4),
                      "note": "bytes = SURVEY 8(d)'s walk formula (K1 with count 1 per walker step) over the "
                              "walkers of one rank; time = the call, wall clock (it contains the host waits)"}
            if world == 1 and not args.no_check:
                sel = np.random.default_rng(0).choice(W, 64, replace=False)
                w_sel = walks.cpu().numpy()[sel]
                need_ids = w_sel[(w_sel >= 1) & (w_sel <= N)]
                OGw = _oracle_rows(G, p_g, need_ids, 1)
                ow_ = OGw.random_walk(GRAPH_SEED, LEN * last, starts[last].cpu().numpy()[sel], et, LEN,
                                      1.0, 1.0, N + 1)
                assert np.array_equal(ow_, w_sel), "sharded deepwalk: walks differ from the oracle"
                checked_s = int(64 * LEN)
            if rank == 0 and not args.no_cpu_baseline and not quiet:
                cpu_s = cpu_walk_cell(args, n2v=args.n2v)
        except AssertionError:
            raise
        except Exception as e:
            roof_s = {"error": repr(e)}
        n2v = None
        if args.n2v:
            # SURVEY 8(d) config 4's "one node2vec run p = 0.25, q = 4", sharded: the C entry
            # euler_gpu_sharded_node2vec_walk (per step: rows of the walkers' nodes from their owners,
            # the draw on the requester), its own roofline / CPU cell / oracle check
            W2, L2 = min(100_000, W), 10
            s2 = starts[0][:W2].contiguous()
            et2 = [[0]] * L2
            w2 = walk(s2, et2, 0.25, 4.0, 3)
            secs2 = []
            for _r in range(3):
                _sync_ranks(wire)
                t0 = time.perf_counter()
                w2 = walk(s2, et2, 0.25, 4.0, 3)
                _sync_ranks(wire)
                secs2.append(time.perf_counter() - t0)
            sec2 = float(np.median(_max_over_ranks(secs2, wire)))
            n2v = {"walkers_per_rank": W2, "walk_len": L2, "p": 0.25, "q": 4.0,
                   "ms": round(sec2 * 1e3, 3), "steps_per_s": world * W2 * L2 / sec2,
                   "orchestration": "euler_gpu_sharded_node2vec_walk (C)" if getattr(S, "c_n2v_fn", None) is not None
                                    else "ShardedSampler.random_walk (Python loop)"}
            try:
                from euler_amd.distributed import c_sharded_node2vec_walk
                if getattr(S, "c_n2v_fn", None) is not None:
                    _w2, st2 = c_sharded_node2vec_walk(G, S.c_transport, s2, et2, 0.25, 4.0, N + 1, 3, S.partitions,
                                                       S.dense_table, return_stats=True)
                    assert torch.equal(_w2, w2)
                    n2v["walk_stats"] = st2
                b2_ = C.c_double(0)
                et_b = (C.c_int32 * L2)(*([0] * L2))
                if world == 1:
                    _lib.check(L.euler_gpu_random_walk_algo_bytes(
                        G._h, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(w2.data_ptr()),
                        W2, et_b, 1, L2, 0.25, 4.0, C.byref(b2_)))
                    n2v["roofline"] = {
                        "kernel": "Node2VecListWaveKernel + FullNb* + front end (the whole call)", "bound": "hbm",
                        "achieved": round(b2_.value / sec2 / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(b2_.value / sec2 / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                        "algorithmic_bytes_per_launch": b2_.value, "avg_launch_ms": round(sec2 * 1e3, 3),
                        "note": "bytes = SURVEY 8(d): (deg(cur) + deg(prev)) x 12 per walker step; time = the "
                                "call, wall clock (three host waits per step)"}
                    if not args.no_check:
                        w2_sel = w2.cpu().numpy()[:64]       # the draw is keyed by the walker's INDEX
                        need2 = w2_sel[(w2_sel >= 1) & (w2_sel <= N)]
                        OG2 = _oracle_rows(G, p_g, need2, 1)
                        o2 = OG2.random_walk(GRAPH_SEED, 3, s2.cpu().numpy()[:64], et2, L2, 0.25, 4.0, N + 1)
                        assert np.array_equal(o2, w2_sel), "sharded node2vec: walks differ from the oracle"
                        n2v["parity_checked_steps"] = int(64 * L2)
                if cpu_s is not None and isinstance(cpu_s.get("node2vec"), dict):
                    n2v["cpu_baseline"] = cpu_s.pop("node2vec")
            except AssertionError:
                raise
            except Exception as e:
                n2v["error"] = repr(e)
        line = {
            "metric": "walker steps/sec, DeepWalk random_walk length 40 (p = q = 1) on the power-law "
                      "graph hash-sharded over the ranks (BASELINE configs[3])",
            "value": world * W * LEN * args.steps / elapsed, "unit": "walker steps/s",
            "n_gpus": min(world, torch.cuda.device_count()),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "deepwalk, sharded: %d walkers x %d steps per rank per step of the "
                                   "bench, graph %d nodes / %d edges (all shards), owner(id) = id %% %d, "
                                   "one exchange per walk step" % (W, LEN, N, args.edges, world),
                       "ranks": world, "graph_build_s": round(build_s, 2), "repeats": len(reps),
                       "repeat_ms_per_step": [round(x_ / args.steps * 1e3, 4) for x_ in reps],
                       "transport": "%s, %d ranks in the communicator" % (dist.get_backend(),
                                                                           dist.get_world_size()),
                       "orchestration": ("euler_gpu_sharded_random_walk (C): levels of merged walkers, "
                                         "%d cohorts" % getattr(S, "walk_cohorts", 0))
                                        if getattr(S, "c_walk_fn", None) is not None else
                                        "ShardedSampler.random_walk (Python): one sample_neighbor per step",
                       "walk_stats": walk_stats, "parity_checked_steps": checked_s,
                       "node2vec": n2v},
            "roofline": roof_s, "cpu_baseline": cpu_s,
        }
        if rank == 0 and not quiet:
            _emit(line)
        return line

    def walk_bytes(walks_, n, L_, p, q):
        b = C.c_double(0)
        et_a = (C.c_int32 * L_)(*([0] * L_))
        _lib.check(L.euler_gpu_random_walk_algo_bytes(
            G._h, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(walks_.data_ptr()),
            n, et_a, 1, L_, p, q, C.byref(b)))
        return b.value

    ms = _events(lambda: G.random_walk(starts[n_steps - 1], et, 1.0, 1.0, N + 1, call_id=7), 5)
    wb = walk_bytes(walks, W, LEN, 1.0, 1.0)
    # parity at bench scale: 64 walkers of the last step against the oracle fed with the rows
    # (exported from HBM) of every node they visit
    last = n_steps - 1
    sel = np.random.default_rng(0).choice(W, 64, replace=False)
    w_sel = walks.cpu().numpy()[sel]
    need_ids = w_sel[(w_sel >= 1) & (w_sel <= N)]
    OGw = _oracle_rows(G, p_g, need_ids, 1)
    ow_ = OGw.random_walk(GRAPH_SEED, LEN * last, starts[last].cpu().numpy()[sel], et, LEN, 1.0, 1.0, N + 1)
    assert np.array_equal(ow_, w_sel), "deepwalk: walks differ from the oracle"
    checked = int(w_sel.shape[0] * LEN)
    n2v = None
    if args.n2v:
        W2, L2 = 100_000, 10
        s2 = starts[0][:W2].contiguous()
        et2 = [[0]] * L2
        w2 = G.random_walk(s2, et2, 0.25, 4.0, N + 1, call_id=3)
        ms2 = _events(lambda: G.random_walk(s2, et2, 0.25, 4.0, N + 1, call_id=3), 2)
        b2 = walk_bytes(w2, W2, L2, 0.25, 4.0)
        # the biased draw is keyed by the walker's INDEX: the first 64 walkers, as walkers 0..63
        w2_sel = w2.cpu().numpy()[:64]
        need2 = w2_sel[(w2_sel >= 1) & (w2_sel <= N)]
        OG2 = _oracle_rows(G, p_g, need2, 1)
        o2 = OG2.random_walk(GRAPH_SEED, 3, s2.cpu().numpy()[:64], et2, L2, 0.25, 4.0, N + 1)
        assert np.array_equal(o2, w2_sel), "node2vec: walks differ from the oracle"
        n2v = {"walkers": W2, "walk_len": L2, "p": 0.25, "q": 4.0, "ms": round(ms2, 3),
               "steps_per_s": W2 * L2 / (ms2 * 1e-3), "algorithmic_bytes": b2,
               "GBps": round(b2 / (ms2 * 1e-3) / 1e9, 1),
               "frac": round(b2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
               "parity_checked_steps": int(64 * L2)}
    line = {
        "metric": "walker steps/sec, DeepWalk random_walk length 40 (p = q = 1) on the 100M-node "
                  "power-law graph (BASELINE configs[3], 1 GPU)",
        "value": W * LEN * args.steps / elapsed, "unit": "walker steps/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "deepwalk: %d walkers x %d steps per step of the bench, graph %d nodes / "
                               "%d edges, weighted" % (W, LEN, N, G.num_edges),
                   "graph_build_s": round(build_s, 2), "repeats": len(reps),
                   "repeat_ms_per_step": [round(x_ / args.steps * 1e3, 4) for x_ in reps],
                   "parity_checked_steps": checked,
                   "node2vec": n2v},
        "roofline": {"kernel": "RandomWalkKernel", "bound": "hbm",
                     "achieved": round(wb / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(wb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                     "traffic": None, "algorithmic_bytes_per_launch": wb, "avg_launch_ms": round(ms, 4)},
        "cpu_baseline": None,
    }
    if not quiet and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_walk_cell(args)
        except Exception as e:
            line["cpu_baseline"] = {"error": repr(e)}
    if not quiet:
        _emit(line)
    return line


def run_products_leg(args):
    """configs[1] as a short leg of the default run: the products-shaped uniform graph, the
    2-hop fanout on two alternating streams, 64 roots checked against the oracle."""
    import copy
    import euler_amd
    N, E = 2_449_029, 123_718_280
    p = euler_amd.synth_params(GRAPH_SEED, N, E, weighted=False)
    G = euler_amd.Graph.synthetic(p)
    G.set_seed(GRAPH_SEED)
    B = args.batch
    steps, warm = 10, 3
    gen = torch.Generator(device="cuda"); gen.manual_seed(4321)
    roots = torch.randint(1, N + 1, (steps + warm, B), generator=gen, device="cuda", dtype=torch.int64)
    et = [[0], [0]]
    side = [torch.cuda.Stream(), torch.cuda.Stream()]

    def loop(first, last):
        res = None
        for i in range(first, last):
            with torch.cuda.stream(side[i % 2]):
                res = G.sample_fanout(roots[i], et, FANOUT, N + 1, call_id=2 * i)
        return res
    torch.cuda.synchronize()
    loop(0, warm + 1)
    torch.cuda.synchronize()
    reps = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = loop(warm, warm + steps)
        torch.cuda.synchronize()
        reps.append(time.perf_counter() - t0)
    elapsed = float(np.median(reps))
    last = warm + steps - 1
    sel = np.random.default_rng(0).choice(B, 64, replace=False)
    r0 = roots[last].cpu().numpy()[sel]
    hop1 = out[0][1].reshape(B, FANOUT[0]).cpu().numpy()[sel]
    hop2 = out[0][2].reshape(B, FANOUT[0], FANOUT[1]).cpu().numpy()[sel]
    need = np.concatenate([r0, hop1.reshape(-1)])
    OG = _oracle_rows(G, p, need[(need >= 1) & (need <= N)], 1)
    on, _, _ = OG.sample_fanout(GRAPH_SEED, 2 * last, r0, et, FANOUT, N + 1)
    assert np.array_equal(on[0], hop1.reshape(-1)) and np.array_equal(on[1], hop2.reshape(-1)), \
        "products: sampled ids differ from the oracle"
    ms_alone = _events(lambda: G.sample_fanout(roots[last], et, FANOUT, N + 1, call_id=5), 10)
    edges = B * (FANOUT[0] + FANOUT[0] * FANOUT[1])
    # uniform weights: no search, no sums read - per sampled edge 16 (out) + 8 (id), per root
    # the record; the expansion's 16 per output edge is the out above
    algo = 36.0 * B + 24.0 * edges + 36.0 * B * FANOUT[0]
    res = {"value": edges * steps / elapsed, "unit": "sampled edges/s",
           "ms_per_step": elapsed / steps * 1e3, "one_stream_ms_per_step": round(ms_alone, 4),
           "roofline_frac": round(algo / (ms_alone * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "parity_checked": int(64 * 275),
           "workload": "ogbn-products-shaped uniform graph (%d nodes / %d edges), fanout [25,10], %d roots "
                       "per step, two streams" % (N, G.num_edges, B)}
    del G
    torch.cuda.empty_cache()
    return res


def run_hashed_leg(args, weighted=True):
    """The metric step on a graph shaped like a converted dataset: the same 100M nodes / 1B
    weighted edges, but every node known by an arbitrary u64 id (hash id map instead of
    row = id - 1) and two edge-type groups per node (Cora's train / train_removed,
    tf_euler/python/dataset/cora.py:36-52); the fanout lists one type per hop, as GraphSAGE
    does.  This is the general form of the one-kernel step (fanout_local.h: WbSamplePairG - hash
    id map, segment limits out of the row's records), which the headline's plain graph never
    reaches."""
    import euler_amd
    from euler_amd import _lib
    L = _lib.lib()
    N, E = args.nodes, args.edges
    t0 = time.time()
    p = euler_amd.synth_params(GRAPH_SEED, N, E, n_types=2, weighted=weighted, hashed_ids=True)
    G = euler_amd.Graph.synthetic(p)
    G.set_seed(GRAPH_SEED)
    torch.cuda.synchronize()
    build_s = time.time() - t0
    B = args.batch
    steps, warm = 10, 3
    gen = torch.Generator(device="cuda"); gen.manual_seed(2468)
    roots = _mix64_t(torch.randint(1, N + 1, (steps + warm, B), generator=gen, device="cuda",
                                   dtype=torch.int64))
    et = [[0], [0]]
    default = -1
    side = [torch.cuda.Stream(), torch.cuda.Stream()]

    def loop(first, last):
        res = None
        for i in range(first, last):
            with torch.cuda.stream(side[i % 2]):
                res = G.sample_fanout(roots[i], et, FANOUT, default, call_id=2 * i)
        return res
    torch.cuda.synchronize()
    loop(0, warm + 1)
    torch.cuda.synchronize()
    reps = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = loop(warm, warm + steps)
        torch.cuda.synchronize()
        reps.append(time.perf_counter() - t0)
    elapsed = float(np.median(reps))
    last = warm + steps - 1
    sel = np.random.default_rng(0).choice(B, 64, replace=False)
    r0 = roots[last].cpu().numpy()[sel]
    hop1 = out[0][1].reshape(B, FANOUT[0]).cpu().numpy()[sel]
    hop2 = out[0][2].reshape(B, FANOUT[0], FANOUT[1]).cpu().numpy()[sel]
    w2 = out[1][1].reshape(B, -1).cpu().numpy()[sel]
    need = np.concatenate([r0, hop1.reshape(-1)])
    OG = _oracle_rows(G, p, need[need != default], 2)
    on, ow, _ot = OG.sample_fanout(GRAPH_SEED, 2 * last, r0, et, FANOUT, default)
    assert np.array_equal(on[0], hop1.reshape(-1)) and np.array_equal(on[1], hop2.reshape(-1)), \
        "hashed ids / 2 types: sampled ids differ from the oracle"
    assert np.array_equal(ow[1], w2.reshape(-1)), "hashed ids / 2 types: weights differ from the oracle"
    r = roots[last].contiguous()
    ms_alone = _events(lambda: G.sample_fanout(r, et, FANOUT, default, call_id=5), 10)
    # SURVEY 8(d) bytes of the step, as for the headline: K1 over the batch + K1 over the
    # globally distinct hop-2 roots + 12 per hop-2 input id + 16 per expanded edge
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    et1 = (C.c_int32 * 1)(0)

    def algo_bytes(x, cnt):
        b = C.c_double(0)
        _lib.check(L.euler_gpu_sample_neighbor_algo_bytes(
            G._h, st, C.c_void_p(x.data_ptr()), x.numel(), et1, 1, cnt, C.byref(b)))
        return b.value
    hop2_roots = out[0][1].reshape(-1)
    uniq2 = torch.unique(hop2_roots).contiguous()
    n2 = hop2_roots.numel()
    algo = algo_bytes(r, FANOUT[0]) + algo_bytes(uniq2, FANOUT[1]) + 12.0 * n2 + 16.0 * n2 * FANOUT[1]
    edges = B * (FANOUT[0] + FANOUT[0] * FANOUT[1])
    res = {"value": edges * steps / elapsed, "unit": "sampled edges/s",
           "ms_per_step": elapsed / steps * 1e3, "one_stream_ms_per_step": round(ms_alone, 4),
           "roofline_frac": round(algo / (ms_alone * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "algorithmic_bytes_per_launch": algo, "parity_checked": int(64 * 275),
           "kernel": "SampleFanoutLeanKernel<.., WB = 2> (the one-kernel step's general form: hash id map, "
                     "edge-type groups, weight-bucket index)" if weighted else
                     "SampleFanoutLeanKernel<.., WB = 6> (general form on uniform weights: the draw is an index "
                     "computation, PivotSample's H1)",
           "graph_build_s": round(build_s, 2), "graph_bytes": G.device_bytes,
           "workload": "the metric step on %d nodes / %d edges with hashed u64 ids and 2 edge-type "
                       "groups per node%s, one listed type per hop, %d roots per step, two streams"
                       % (N, G.num_edges, "" if weighted else ", all weights 1.0 (what the reference's dataset "
                          "converters write)", B)}
    # the same step listing BOTH type groups per hop - what the reference's evaluation does
    # (metapath = [all_edge_type] * layers, examples/graphsage/run_graphsage.py:57): a type draw
    # per sample, then the neighbour draw (fanout_local.h, WB == 3); checked against the oracle
    et_all = [[0, 1], [0, 1]]

    def loop_all(first, last):
        res_ = None
        for i in range(first, last):
            with torch.cuda.stream(side[i % 2]):
                res_ = G.sample_fanout(roots[i], et_all, FANOUT, default, call_id=2 * i)
        return res_
    torch.cuda.synchronize()
    loop_all(0, warm + 1)
    reps_all = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out_all = loop_all(warm, warm + steps)
        torch.cuda.synchronize()
        reps_all.append(time.perf_counter() - t0)
    hop1a = out_all[0][1].reshape(B, FANOUT[0]).cpu().numpy()[sel]
    hop2a = out_all[0][2].reshape(B, FANOUT[0], FANOUT[1]).cpu().numpy()[sel]
    t2a = out_all[2][1].reshape(B, -1).cpu().numpy()[sel]
    need = np.concatenate([r0, hop1a.reshape(-1)])
    OGa = _oracle_rows(G, p, need[need != default], 2)
    on, _ow, ot = OGa.sample_fanout(GRAPH_SEED, 2 * last, r0, et_all, FANOUT, default)
    assert np.array_equal(on[0], hop1a.reshape(-1)) and np.array_equal(on[1], hop2a.reshape(-1)), \
        "hashed ids / all types: sampled ids differ from the oracle"
    assert np.array_equal(ot[1], t2a.reshape(-1)), "hashed ids / all types: types differ from the oracle"
    el_all = float(np.median(reps_all))
    ms_all_alone = _events(lambda: G.sample_fanout(r, et_all, FANOUT, default, call_id=5), 10)
    et2 = (C.c_int32 * 2)(0, 1)

    def algo_bytes2(x, cnt):          # the K1 formula with its type-draw term (both groups listed)
        b = C.c_double(0)
        _lib.check(L.euler_gpu_sample_neighbor_algo_bytes(
            G._h, st, C.c_void_p(x.data_ptr()), x.numel(), et2, 2, cnt, C.byref(b)))
        return b.value
    h2a = out_all[0][1].reshape(-1)
    algo_all = (algo_bytes2(r, FANOUT[0]) + algo_bytes2(torch.unique(h2a).contiguous(), FANOUT[1])
                + 12.0 * h2a.numel() + 16.0 * h2a.numel() * FANOUT[1])
    res["all_types_per_hop"] = {"value": edges * steps / el_all, "unit": "sampled edges/s",
                                "ms_per_step": round(el_all / steps * 1e3, 4), "edge_types": et_all,
                                "one_stream_ms_per_step": round(ms_all_alone, 4),
                                "roofline_frac": round(algo_all / (ms_all_alone * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                "algorithmic_bytes_per_launch": algo_all,
                                "parity_checked": int(64 * 275),
                                "kernel": "SampleFanoutLeanKernel<.., WB = %d> (a type draw per sample)"
                                          % (4 if weighted else 5)}
    del G, out, out_all
    torch.cuda.empty_cache()
    return res


def run_unique_leg(args, G, p_g):
    """The metric step in the (unique rows, index) form (euler_gpu_sample_fanout_unique): the
    GQL result before DATA_GATHER - hop 2 as distinct rows + the row of every hop-1 sample;
    the whole result is compared with the dense form on the device."""
    N, B = args.nodes, args.batch
    gen = torch.Generator(device="cuda"); gen.manual_seed(99)
    r = torch.randint(1, N + 1, (B,), generator=gen, device="cuda", dtype=torch.int64)
    et = [[0], [0]]
    id1, w1, t1, idx, rid, rw, rt = G.sample_fanout_unique(r, et, FANOUT, N + 1, call_id=8)
    dn, dw, dt = G.sample_fanout(r, et, FANOUT, N + 1, call_id=8)
    assert torch.equal(id1.reshape(-1), dn[1]) and torch.equal(rid[idx].reshape(-1), dn[2])
    assert torch.equal(rw[idx].reshape(-1), dw[1]) and torch.equal(rt[idx].reshape(-1), dt[1])
    rows = int(torch.unique(idx).numel())
    del dn, dw, dt
    # ... and 64 roots of it against the ORACLE (rows exported from HBM, host generator spot check)
    sel = np.random.default_rng(0).choice(B, 64, replace=False)
    r0 = r.cpu().numpy()[sel]
    sel_t = torch.as_tensor(sel).cuda()
    hop1 = id1.reshape(B, FANOUT[0])[sel_t].cpu().numpy()
    idx_sel = idx.reshape(B, FANOUT[0])[sel_t].reshape(-1)
    hop2 = rid[idx_sel].reshape(64, -1).cpu().numpy()
    w2 = rw[idx_sel].reshape(64, -1).cpu().numpy()
    need = np.concatenate([r0, hop1.reshape(-1)])
    OG = _oracle_rows(G, p_g, need[(need >= 1) & (need <= N)], 1)
    on, ow, _ot = OG.sample_fanout(GRAPH_SEED, 8, r0, et, FANOUT, N + 1)
    assert np.array_equal(on[0], hop1.reshape(-1)) and np.array_equal(on[1], hop2.reshape(-1)), \
        "unique rows: ids differ from the oracle"
    assert np.array_equal(ow[1], w2.reshape(-1)), "unique rows: weights differ from the oracle"
    ms = _events(lambda: G.sample_fanout_unique(r, et, FANOUT, N + 1, call_id=8), 10)
    edges = B * (FANOUT[0] + FANOUT[0] * FANOUT[1])
    # SURVEY 8(d) bytes of this contract: K1 over the batch + K1 over the globally distinct
    # hop-2 roots (their 16 output bytes per sampled edge are the rows) + 8 + 4 per hop-2 input
    # id (duplicate detection) + 4 per row-index entry written; no expansion
    from euler_amd import _lib
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    et1 = (C.c_int32 * 1)(0)

    def algo_bytes(x, cnt):
        b = C.c_double(0)
        _lib.check(_lib.lib().euler_gpu_sample_neighbor_algo_bytes(
            G._h, st, C.c_void_p(x.data_ptr()), x.numel(), et1, 1, cnt, C.byref(b)))
        return b.value
    hop2_roots = id1.reshape(-1).contiguous()
    algo = (algo_bytes(r, FANOUT[0]) + algo_bytes(torch.unique(hop2_roots).contiguous(), FANOUT[1])
            + 12.0 * hop2_roots.numel() + 4.0 * idx.numel())
    return {"value": edges / (ms * 1e-3), "unit": "sampled edges/s (as rows + index)",
            "ms_per_step": round(ms, 4),
            "roofline_frac": round(algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "algorithmic_bytes_per_launch": algo, "parity_checked": int(edges),
            "parity_checked_vs_oracle": int(64 * 275),
            "distinct_rows": rows, "positions": int(idx.numel()),
            "workload": "the metric step, hop 2 left as %d distinct rows + a row index per hop-1 sample "
                        "(one stream, output buffers allocated per call)" % rows}


def run_sage_leg(args, G, p_g):
    """SageDataFlow block construction (euler_gpu_sage_blocks: sampler + first-occurrence
    unique + res_n_id + edge_index per hop, one enqueue) on the metric graph: blocks/s; the
    blocks are compared with the op-by-op composition of the base class."""
    from euler_amd.dataflow import SageDataFlow
    N = args.nodes
    B = 16384
    gen = torch.Generator(device="cuda"); gen.manual_seed(77)
    r = torch.randint(1, N + 1, (B,), generator=gen, device="cuda", dtype=torch.int64)
    flow = SageDataFlow(G, FANOUT, [[0], [0]], add_self_loops=True, max_id=N)
    G.set_seed(GRAPH_SEED, 4000)
    df = flow(r)
    flow.fused = False
    G.set_seed(GRAPH_SEED, 4000)
    df2 = flow(r)
    flow.fused = True
    n_edges = 0
    for b1, b2 in zip(df, df2):
        assert torch.equal(b1.n_id, b2.n_id) and torch.equal(b1.res_n_id, b2.res_n_id)
        assert torch.equal(b1.edge_index, b2.edge_index)
        n_edges += int(b1.edge_index.shape[1])
    # ... and against the ORACLE's composition of the reference's flow (sample_neighbor of the
    # unique frontier, first-occurrence unique, edge_index arithmetic) on this graph for a batch
    # whose frontier rows can be exported: 128 roots -> ~3 K frontier rows of the 100M-node graph
    Bo = 128
    ro = r[:Bo].contiguous()
    G.set_seed(GRAPH_SEED, 5000)
    dfo = flow(ro)
    ro_np = ro.cpu().numpy()
    G.set_seed(GRAPH_SEED, 5000)
    nb1 = G.sample_neighbor(ro, [0], FANOUT[0], N + 1, call_id=5000)[0].reshape(-1).cpu().numpy()
    need = np.concatenate([ro_np, nb1])
    OG = _oracle_rows(G, p_g, need[(need >= 1) & (need <= N)], 1)
    want = _oracle_sage_blocks(OG, GRAPH_SEED, 5000, ro_np, [[0], [0]], FANOUT, N + 1)
    o_edges = 0
    for blk, (wn, wr, we) in zip(dfo.blocks, want):
        assert np.array_equal(blk.n_id.cpu().numpy(), wn), "sage blocks: n_id differs from the oracle"
        assert np.array_equal(blk.res_n_id.cpu().numpy(), wr), "sage blocks: res_n_id differs"
        assert np.array_equal(blk.edge_index.cpu().numpy(), we), "sage blocks: edge_index differs"
        o_edges += int(we.shape[1])
    G.set_seed(GRAPH_SEED)
    ms = _events(lambda: flow(r), 10)
    # the same enqueue without the host's read of the layer sizes (padded tensors + counts on the
    # device: a consumer that masks never waits)
    ms_nosync = _events(lambda: G.sage_blocks(r, [[0], [0]], FANOUT, default_node=N + 1, sync=False), 10)
    # SURVEY 8(d) bytes of the flow, hop by hop over the sizes this minibatch really has: K1
    # over the layer's nodes + 8 + 4 per id that goes through the first-occurrence unique
    # ([neighbours | nodes]) + 8 per distinct id written (n_id) + 8 per res_n_id entry + 2 x 8
    # per edge_index column
    from euler_amd import _lib
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    et1 = (C.c_int32 * 1)(0)
    algo = 0.0
    # (df is ordered from the outermost hop inwards: blocks[-1] is hop 0)
    layer = r
    for h, blk in enumerate(reversed(list(df))):
        b_ = C.c_double(0)
        x = layer.contiguous()
        _lib.check(_lib.lib().euler_gpu_sample_neighbor_algo_bytes(
            G._h, st, C.c_void_p(x.data_ptr()), x.numel(), et1, 1, FANOUT[h], C.byref(b_)))
        m_in = x.numel() * (FANOUT[h] + 1)
        algo += b_.value + 12.0 * m_in + 8.0 * blk.n_id.numel() + 8.0 * x.numel() \
            + 16.0 * blk.edge_index.shape[1]
        layer = blk.n_id
    # GraphSAGE callers at small batch: B = 1 024 roots per minibatch, one flow per call against M = 64
    # minibatches' flows in ONE enqueue (euler_gpu_sage_blocks_multi); sampled edges = the samples the
    # hops draw (counts[h] x fanout[h]), read once from the counts of a checked run
    Bs, Ms = 1024, 64
    rs = torch.randint(1, N + 1, (Ms, Bs), generator=gen, device="cuda", dtype=torch.int64)
    G.set_seed(GRAPH_SEED, 6000)
    per_mb = G.sage_blocks_multi(rs, [[0], [0]], FANOUT, default_node=N + 1)
    multi_checked = 0
    for b_ in (0, 31, 63):
        G.set_seed(GRAPH_SEED)
        one = G.sage_blocks(rs[b_], [[0], [0]], FANOUT, default_node=N + 1, call_id=6000 + 2 * b_)
        assert list(one[1]) == list(per_mb[b_][1])
        for x_, y_ in zip(one[0], per_mb[b_][0]):
            for u_, v_ in zip(x_, y_):
                assert torch.equal(u_, v_), "sage_blocks_multi differs from the separate call"
        multi_checked += 1
    drawn = sum(c_[h_] * FANOUT[h_] for _blk, c_ in per_mb for h_ in range(2))
    G.set_seed(GRAPH_SEED)
    ms_multi = _events(lambda: G.sage_blocks_multi(rs, [[0], [0]], FANOUT, default_node=N + 1, sync=False), 10)
    ms_small = _events(lambda: G.sage_blocks(rs[0], [[0], [0]], FANOUT, default_node=N + 1, sync=False), 20)
    small = {"B": Bs, "M": Ms, "us_per_minibatch_single_calls": round(ms_small * 1e3, 2),
             "multi_ms_per_launch": round(ms_multi, 4),
             "multi_us_per_minibatch": round(ms_multi * 1e3 / Ms, 2),
             "multi_sampled_edges_per_s": drawn / (ms_multi * 1e-3),
             "single_sampled_edges_per_s": (drawn / Ms) / (ms_small * 1e-3),
             "multi_checked": "%d of the %d minibatches == separate euler_gpu_sage_blocks calls" % (multi_checked, Ms)}
    return {"value": 1e3 / ms, "unit": "minibatches (2 blocks each)/s", "ms_per_step": round(ms, 4),
            "ms_per_step_without_host_read": round(ms_nosync, 4), "small_batch": small,
            "roofline_frac": round(algo / (ms_nosync * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "roofline_frac_with_host_read": round(algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "algorithmic_bytes_per_minibatch": algo,
            "parity_checked": n_edges, "parity_checked_vs_oracle": o_edges,
            "block_edges_per_s": n_edges / (ms * 1e-3),
            "workload": "SageDataFlow, %d roots, fanouts %s, self loops: %d block edges per minibatch; "
                        "one host read (the layer sizes) per minibatch" % (B, FANOUT, n_edges)}


def run_node_legs(args, G, p_g):
    """SampleNode (K2: Graph::SampleNode over alias tables, core/graph/graph.cc:221-245,
    common/alias_method.cc:66-78) on tables of the metric graph's N nodes - 4 node types, f32
    weights in [0.5, 4.5), the global sampler built by Graph.set_node_sampler - and the
    DeepWalk minibatch of the reference's example (examples/deepwalk/deepwalk.py:47-63:
    random_walk -> gen_pair -> sample_node(batch x pairs x num_negs)), both checked against the
    oracle's restatement on the same arrays.  SURVEY 8(d) bytes of a draw: 8 (id) + 4 (prob) +
    8 (alias id, only when the coin misses) + 8 out; the legs count 20 per draw, the lower
    bound."""
    from oracle import oracle as O
    from euler_amd import _lib, euler_ops
    L = _lib.lib()
    N = args.nodes
    out = {}
    t0 = time.time()
    ids = np.arange(1, N + 1, dtype=np.uint64)
    types = (((ids * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(61)) & np.uint64(3)).astype(np.int32)
    weights = (0.5 + 4.0 * np.random.default_rng(11).random(N, dtype=np.float32)).astype(np.float32)
    G.set_node_sampler(None, types, weights, 4)
    build_s = time.time() - t0
    t0 = time.time()
    osamp = None
    if not args.no_check:
        osamp = O.lib().eo_node_sampler_create(N, O._p(ids, O._u64p), O._p(types, O._i32p),
                                               O._p(weights, O._f32p), 4)
    oracle_s = time.time() - t0

    def oracle_nodes(call_id, node_type, count):
        nt = np.asarray([node_type], np.int32)
        o = np.zeros(count, np.uint64)
        got = O.lib().eo_sample_node(osamp, GRAPH_SEED, call_id, O._p(nt, O._i32p), 1, count, O._p(o, O._u64p))
        assert got == count
        return o
    count = 32 * 1024 * 1024
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    buf = torch.empty(count, dtype=torch.int64, device="cuda")
    legs = {}
    for name_, nt in (("all_types", -1), ("one_type", 1)):
        nt_a = (C.c_int32 * 1)(nt)

        def call(cid=7):
            _lib.check(L.euler_gpu_sample_node(G._h, st, GRAPH_SEED, cid, nt_a, 1, count,
                                               C.c_void_p(buf.data_ptr())))
        ms = _events(call, 10)
        chk = 0
        if osamp is not None:
            call(7)
            torch.cuda.synchronize()
            head = buf[:1 << 18].cpu().numpy().view(np.uint64)
            assert np.array_equal(head, oracle_nodes(7, nt, 1 << 18)), "sample_node differs from the oracle"
            chk = 1 << 18
        # 8 id + 4 prob + 8 out per draw, + 8 for the alias id of the draws that take it
        # (counted for none of them: the lower bound of SURVEY 8(d)'s 20 .. 28 bytes)
        algo = 20.0 * count
        legs[name_] = {"ms": round(ms, 4), "nodes_per_s": count / (ms * 1e-3),
                       "algorithmic_bytes": algo, "GBps": round(algo / (ms * 1e-3) / 1e9, 1),
                       "roofline_frac": round(algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                       "table_lines_per_s": count / (ms * 1e-3), "parity_checked": chk}
    head_ = legs["all_types"]
    out["sample_node"] = {
        "value": head_["nodes_per_s"], "unit": "sampled nodes/s", "ms_per_step": head_["ms"],
        "roofline_frac": head_["roofline_frac"], "parity_checked": head_["parity_checked"],
        "count": count, "one_type": legs["one_type"], "all_types": head_,
        "sampler_build_s": round(build_s, 2), "oracle_build_s": round(oracle_s, 2),
        "bound": "one random 32-byte table entry per draw: a draw moves one 128-byte line for 20-28 "
                 "algorithmic bytes, so the line rate of random reads beyond the L2 (54 G lines/s measured, "
                 "tools/ubench_gather.hip) caps K2 at ~0.19 of the byte roofline; table_lines_per_s against "
                 "that rate is the figure of merit (profiles/r5_sample_node_pmc.json: read requests per draw)",
        "workload": "SampleNode count = %d over %d nodes in 4 node types (type -1: type draw + node draw, "
                    "4 uniforms per sample; one type: 2), weights f32 in [0.5, 4.5)" % (count, N)}
    del buf
    # ---- the DeepWalk minibatch (examples/deepwalk/deepwalk.py:47-63)
    sys.path.insert(0, os.path.join(ROOT, "examples", "python"))
    import deepwalk_minibatch as dm
    prev = None
    try:
        prev = euler_ops.get_default_graph()
    except Exception:
        prev = None
    euler_ops.set_default_graph(G)
    try:
        Bd, WL, NEG = 131072, 3, 5            # run_deepwalk.py's walk_len / windows / num_negs, a big batch
        gen = torch.Generator(device="cuda"); gen.manual_seed(31)
        inputs = torch.randint(1, N + 1, (8, Bd), generator=gen, device="cuda", dtype=torch.int64)

        def mb(i, call=None):
            if call is not None:
                G.set_seed(GRAPH_SEED, call)
            return dm.to_sample(inputs[i % 8], 1, [0], N, WL, 1.0, 1.0, 1, 1, NEG)
        src, pos, negs = mb(0, 600)
        if osamp is not None:
            # 64 inputs: their walks (rows exported from HBM), pairs and the call's first negatives
            sel = np.random.default_rng(2).choice(Bd, 64, replace=False)
            inp = inputs[0].cpu().numpy()
            pairs = src.numel() // Bd
            walk = G.random_walk(inputs[0], [[0]] * WL, 1.0, 1.0, N + 1, call_id=600).cpu().numpy()[sel]
            OGw = _oracle_rows(G, p_g, walk[(walk >= 1) & (walk <= N)], 1)
            opath = OGw.random_walk(GRAPH_SEED, 600, inp[sel], [[0]] * WL, WL, 1.0, 1.0, N + 1)
            opair = O.gen_pair(opath, 1, 1)
            assert np.array_equal(src.reshape(Bd, pairs).cpu().numpy()[sel], opair[..., 0])
            assert np.array_equal(pos.reshape(Bd, pairs).cpu().numpy()[sel], opair[..., 1])
            want = oracle_nodes(600 + WL, 1, 1 << 16)
            assert np.array_equal(negs.reshape(-1)[:1 << 16].cpu().numpy().view(np.uint64), want), \
                "deepwalk minibatch: negatives differ from the oracle"
        G.set_seed(GRAPH_SEED)
        ms = _events(lambda: mb(1), 10)
        pairs_n = int(src.shape[0])
        # bytes: the walk's K1 terms (count 1) + 16 per pair written + 20 per negative
        wb = C.c_double(0)
        et_a = (C.c_int32 * WL)(*([0] * WL))
        walk_all = G.random_walk(inputs[1], [[0]] * WL, 1.0, 1.0, N + 1, call_id=5)
        _lib.check(L.euler_gpu_random_walk_algo_bytes(G._h, st, C.c_void_p(walk_all.data_ptr()), Bd, et_a, 1,
                                                      WL, 1.0, 1.0, C.byref(wb)))
        algo = wb.value + 8.0 * Bd * (WL + 1) + 16.0 * pairs_n + 20.0 * negs.numel()
        out["deepwalk_minibatch"] = {
            "value": 1e3 / ms, "unit": "minibatches/s", "ms_per_step": round(ms, 4),
            "pairs_per_s": pairs_n / (ms * 1e-3), "negatives_per_s": negs.numel() / (ms * 1e-3),
            "roofline_frac": round(algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "algorithmic_bytes": algo,
            "parity_checked": (64 * pairs_n // Bd * 2 + (1 << 16)) if osamp is not None else 0,
            "workload": "examples/python/deepwalk_minibatch.py to_sample (deepwalk.py:47-63): batch %d, "
                        "walk_len %d, windows 1 / 1, %d negatives per pair: %d pairs, %d negatives per "
                        "minibatch, through the euler_ops surface on one stream" % (Bd, WL, NEG, pairs_n,
                                                                                    negs.numel())}
    except Exception as e:
        out["deepwalk_minibatch"] = {"error": repr(e)}
    finally:
        if prev is not None:
            euler_ops.set_default_graph(prev)
        if osamp is not None:
            O.lib().eo_node_sampler_destroy(osamp)
    return out


def secondary_legs(args, G, p_g):
    """BASELINE configs[1], [3], [4] as short legs of the default run (rank 0, one GPU), each
    with a spot check against the oracle at bench scale: {value, ms_per_step, roofline_frac,
    parity_checked}."""
    import copy
    sec = {}
    t0 = time.time()
    try:
        a = copy.copy(args)
        a.steps, a.warmup, a.repeats, a.n2v = 3, 1, 3, True
        d = run_deepwalk(a, G, p_g, quiet=True)
        sec["deepwalk"] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                           "roofline_frac": d["roofline"]["frac"],
                           "parity_checked": d["config"]["parity_checked_steps"],
                           "workload": d["config"]["workload"]}
        n2 = d["config"]["node2vec"]
        sec["node2vec"] = {"value": n2["steps_per_s"], "unit": "walker steps/s", "ms_per_step": n2["ms"],
                           "roofline_frac": n2["frac"], "parity_checked": n2["parity_checked_steps"],
                           "workload": "p = 0.25, q = 4: %d walkers x %d steps on the metric graph"
                                       % (n2["walkers"], n2["walk_len"])}
    except Exception as e:          # a side measurement must not fail the bench
        sec["deepwalk"] = {"error": repr(e)}
    for name_, fn_ in (("fanout_unique_rows", run_unique_leg), ("sage_blocks", run_sage_leg)):
        try:
            sec[name_] = fn_(args, G, p_g)
        except Exception as e:
            sec[name_] = {"error": repr(e)}
    try:
        sec.update(run_node_legs(args, G, p_g))
    except Exception as e:
        sec["sample_node"] = {"error": repr(e)}
    try:
        sec["products"] = run_products_leg(args)
    except Exception as e:
        sec["products"] = {"error": repr(e)}
    try:
        sec["metric_hashed_T2"] = run_hashed_leg(args)
    except Exception as e:
        sec["metric_hashed_T2"] = {"error": repr(e)}
    try:      # ... with all weights 1.0: the shape of every dataset the reference ships
        sec["metric_hashed_T2_unweighted"] = run_hashed_leg(args, weighted=False)
    except Exception as e:
        sec["metric_hashed_T2_unweighted"] = {"error": repr(e)}
    try:
        a = copy.copy(args)
        a.steps, a.warmup, a.repeats = 10, 3, 3
        h = run_hetero(a, quiet=True)
        sec["hetero"] = {"value": h["value"], "unit": h["unit"], "ms_per_step": h["ms_per_step"],
                         "roofline_frac": h["roofline"]["frac"],
                         "parity_checked": h["config"]["parity_checked_edges"],
                         "workload": h["config"]["workload"]}
    except Exception as e:
        sec["hetero"] = {"error": repr(e)}
    sec["seconds"] = round(time.time() - t0, 1)
    return sec


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks
    ourselves (torch.distributed.run, one process per GPU, rendezvous on
    127.0.0.1) and pass their output through.  Returns the exit code."""
    import socket
    import subprocess
    visible = torch.cuda.device_count()
    if visible < args.gpus and not args.oversubscribe:
        print("bench.py: --gpus %d but only %d GPU(s) visible (use --oversubscribe for a "
              "functional check on fewer devices)" % (args.gpus, visible), file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    visible = torch.cuda.device_count()
    shared_gpus = world > visible           # ranks share devices: functional check only
    if shared_gpus and not args.oversubscribe:
        raise SystemExit("bench.py: %d ranks but %d GPU(s) visible (pass --oversubscribe)"
                         % (world, visible))
    local_rank %= visible
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    replicas = bool(args.replicas) and world > 1
    sharded = (world > 1 and not replicas) or args.force_sharded
    backend = "gloo" if shared_gpus else "nccl"
    if sharded or world > 1:
        global _DEFERRED
        _DEFERRED = []
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import euler_amd
    from euler_amd import _lib
    L = _lib.lib()
    for kv in filter(None, args.tuning.split(",")):
        k_, v_ = kv.split("=")
        _lib.check(L.euler_gpu_set_tuning(int(k_), int(v_)))
    if args.workload in ("hetero", "deepwalk"):
        (run_hetero if args.workload == "hetero" else run_deepwalk)(args)
        _teardown(sharded or world > 1)
        return
    weighted = args.workload != "products"
    if args.workload == "products":
        # configs[1]: ogbn-products' public shape (2 449 029 nodes, 61 859 140 undirected =
        # 123 718 280 directed edges), all weights 1.0; `ogb` is not installed and there is
        # no network: a degree-sequence-matched synthetic graph of that size
        args.nodes, args.edges = 2_449_029, 123_718_280
        args.no_cpu_baseline = True

    t0 = time.time()
    p = euler_amd.synth_params(GRAPH_SEED, args.nodes, args.edges, weighted=weighted)
    g_parts = 1 if replicas else world
    G = euler_amd.Graph.synthetic(p, device=local_rank, partitions=g_parts,
                                  shard_index=0 if replicas else rank, shards=g_parts)
    G.set_seed(GRAPH_SEED)
    torch.cuda.synchronize()
    build_s = time.time() - t0

    B = args.batch
    n_steps = args.steps + args.warmup
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    roots = torch.randint(1, args.nodes + 1, (n_steps, B), generator=gen,
                          device=dev, dtype=torch.int64)
    et = [[0], [0]]
    default_node = args.nodes + 1

    def sync():
        if sharded or world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sustained = None
    if sharded:
        # A sharded hop makes the host wait once (the bucket sizes of its front
        # end).  K minibatches are kept in flight from ONE host thread
        # (euler_amd.distributed.run_interleaved): each has its own sampler
        # (front-end handle, id-indexed table, counts mailbox) and HIP stream, the
        # hops of the batches alternate, and every rank issues the same sequence
        # of collectives because the schedule does not depend on the data.
        from euler_amd.distributed import gpu_sharded_sampler, run_interleaved
        K = max(1, min(args.pipeline, args.steps))
        samplers = [gpu_sharded_sampler(G, partitions=world) for _ in range(K)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(K)]

        trace = [] if os.environ.get("EULER_BENCH_TRACE") else None

        def run(first, last):
            def make(j):
                if trace is not None:
                    trace.append((first + j, time.perf_counter(),
                                  torch.cuda.memory_stats()["num_device_alloc"]))
                return samplers[j % K].sample_fanout_steps(
                    roots[first + j], et, FANOUT, default_node, call_id=2 * (first + j))
            kept = [None]                    # only the last minibatch's outputs stay alive

            def consume(job, value):
                kept[0] = value
            run_interleaved(make, last - first, K,
                            enter=lambda k: torch.cuda.stream(streams[k]), on_result=consume)
            res = kept
            for st_ in streams:
                st_.synchronize()
            return res[-1] if res else None

        torch.cuda.synchronize()          # roots were produced on the default stream
        # The distinct-id counts differ from batch to batch, so the caching allocator
        # keeps meeting new tensor sizes for a while (130 device allocations over the
        # first 20 steps, none afterwards): PREWARM extra untimed steps on top of the W
        # the caller asked for keep those out of the timed region.
        for _ in range(PREWARM // max(n_steps, 1) + 1):
            run(0, min(n_steps, PREWARM))
        run(0, args.warmup)
        sync()
        gc.collect(); gc.freeze(); gc.disable()      # a gen-2 collection costs ~40 ms
        wire0 = sum(s_.bytes_sent for s_ in samplers)
        rep_secs = []
        for _rep in range(max(1, args.repeats)):
            sync()
            t0 = time.perf_counter()
            out = run(args.warmup, n_steps)
            sync()
            rep_secs.append(time.perf_counter() - t0)
        gc.enable()
        wire_bytes = (sum(s_.bytes_sent for s_ in samplers) - wire0) / len(rep_secs)
        if trace is not None and rank == 0:
            tail = trace[-(n_steps - args.warmup):]
            print("trace (step, ms since previous job started, device allocs so far):",
                  [(a[0], round((a[1] - b[1]) * 1e3, 2), a[2]) for a, b in zip(tail[1:], tail[:-1])],
                  file=sys.stderr)
    def unsharded_run(Gx, extras=True):
        """The K steps on the unsharded graph Gx (this rank's whole graph): --streams callers'
        streams alternating; returns (rep_secs, one_stream secs, last outputs, sustained)."""
        def step(i):
            return Gx.sample_fanout(roots[i % n_steps], et, FANOUT, default_node, call_id=2 * (i % n_steps))

        n_streams = max(1, args.streams)
        side = [torch.cuda.Stream(device=dev) for _ in range(n_streams)] if n_streams > 1 else None

        def loop(first, last, streams):
            res = None
            if streams is None:
                for i in range(first, last):
                    res = step(i)
            else:
                for i in range(first, last):
                    with torch.cuda.stream(streams[i % len(streams)]):
                        res = step(i)
            return res

        torch.cuda.synchronize()              # roots were produced on the default stream
        out_ = loop(0, max(args.warmup, 2 * n_streams), side)
        sync()
        gc.collect(); gc.freeze(); gc.disable()      # a gen-2 collection costs ~40 ms
        one_stream_ = []
        if side is not None and extras:       # the same K steps on ONE stream, for the record
            loop(0, args.warmup, None)
            for _rep in range(3):
                sync()
                t0 = time.perf_counter()
                loop(args.warmup, n_steps, None)
                sync()
                one_stream_.append(time.perf_counter() - t0)
            loop(0, 2 * n_streams, side)
        rep_secs_ = []
        for _rep in range(max(1, args.repeats)):
            sync()
            t0 = time.perf_counter()
            out_ = loop(args.warmup, n_steps, side)
            sync()
            rep_secs_.append(time.perf_counter() - t0)
        # the same loop as ONE long region (>= 1 s of device time): what the K-step bursts above
        # are extrapolated to (roots cycle through the n_steps batches; call ids keep counting)
        sustained_ = None
        if world == 1 and extras and args.sustain_steps > 0 and side is not None:
            ns_ = args.sustain_steps
            sync()
            t0 = time.perf_counter()
            for i in range(ns_):
                with torch.cuda.stream(side[i % n_streams]):
                    Gx.sample_fanout(roots[i % n_steps], et, FANOUT, default_node, call_id=2 * (n_steps + i))
            sync()
            dt_ = time.perf_counter() - t0
            sustained_ = {"steps": ns_, "seconds": round(dt_, 4), "ms_per_step": round(dt_ / ns_ * 1e3, 4),
                          "edges_per_s": B * (FANOUT[0] + FANOUT[0] * FANOUT[1]) * ns_ / dt_,
                          "streams": n_streams}
        gc.enable()
        return rep_secs_, one_stream_, out_, sustained_

    def reduce_secs(secs):
        if world <= 1:
            return secs
        t = torch.tensor(secs, device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)          # slowest rank, per repetition
        return [float(x) for x in t.tolist()]

    replicas_leg = None
    one_stream = []
    if not sharded:
        rep_secs, one_stream, out, sustained = unsharded_run(G)
    elif world > 1 and not args.no_replicas_leg and args.workload == "metric":
        # beside the hash-sharded headline: every rank with the WHOLE graph in its own HBM
        # (68.6 GB of 288), the unsharded step on its own roots - what sharding costs
        try:
            Gr = euler_amd.Graph.synthetic(p, device=local_rank)
            Gr.set_seed(GRAPH_SEED)
            r_secs, _one, _out, _sus = unsharded_run(Gr, extras=False)
            r_secs = reduce_secs(r_secs)
            r_el = float(np.median(r_secs))
            replicas_leg = {"value": B * 275 * world * args.steps / r_el, "unit": "sampled edges/s",
                            "ms_per_step": r_el / args.steps * 1e3, "graph_bytes_per_gpu": Gr.device_bytes,
                            "repeat_ms_per_step": [round(x / args.steps * 1e3, 4) for x in r_secs],
                            "what": "graph replicated on every GPU, no exchange (SURVEY H7): the same "
                                    "roots, steps and timing protocol as the sharded headline"}
            del Gr, _out
            torch.cuda.empty_cache()
        except Exception as e:
            replicas_leg = {"error": repr(e)}
    exchanged = None
    wire_dev = dev if backend == "nccl" else "cpu"
    rep_secs = reduce_secs(rep_secs)
    # EXACTLY K steps were timed, `repeats` times over; the line reports the median
    elapsed = float(np.median(rep_secs))
    if world > 1 and sharded:
        t = torch.tensor([float(wire_bytes)], device=wire_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        exchanged = float(t.item()) / args.steps          # bytes / step, all ranks
    elif sharded:
        exchanged = 0.0                                   # one rank: self exchanges only
    edges_per_step = B * (FANOUT[0] + FANOUT[0] * FANOUT[1]) * world
    value = edges_per_step * args.steps / elapsed

    # ---- parity check at full size, after the timed region: --check-roots roots of the last step
    # (default 4 096 = 1.1 M sampled edges) against the CPU oracle fed with rows exported from HBM
    # (oracle/step_check.py: all of their hop-1 samples, every hop-2 position against the first
    # position of its node, one row per distinct hop-1 child against the oracle); single GPU only
    checked = None
    if (world == 1 or replicas) and rank == 0 and not args.no_check:
        from oracle import oracle as O
        from oracle.step_check import check_fanout_step
        last = n_steps - 1
        n_chk = max(1, min(B, args.check_roots))
        sel = torch.as_tensor(np.sort(np.random.default_rng(0).choice(B, n_chk, replace=False))).to(dev)
        r_sel = roots[last][sel]
        c1_, c2_ = FANOUT
        sub_n = [r_sel, out[0][1].reshape(B, c1_)[sel].reshape(-1), out[0][2].reshape(B, c1_ * c2_)[sel].reshape(-1)]
        sub_w = [out[1][0].reshape(B, c1_)[sel].reshape(-1), out[1][1].reshape(B, c1_ * c2_)[sel].reshape(-1)]
        sub_t = [out[2][0].reshape(B, c1_)[sel].reshape(-1), out[2][1].reshape(B, c1_ * c2_)[sel].reshape(-1)]
        # the rows in HBM are the rows the HOST generator (oracle/eo_synth.c) produces for
        # these ids - a generator fault at full size cannot hide behind "oracle fed with
        # exported rows"
        need = np.unique(np.concatenate([r_sel.cpu().numpy()[:64], sub_n[1].cpu().numpy()[:1600]])).astype(np.uint64)
        need = need[need <= args.nodes]
        rp, te, nb, pw, tp = G.export_rows(need)
        po = O.SynthParams()
        for f_, _t in po._fields_:
            setattr(po, f_, getattr(p, f_))
        for j_ in np.random.default_rng(1).choice(len(need), min(256, len(need)), replace=False):
            h_ = O.synth_csr(po, int(need[j_]) - 1, int(need[j_]))
            b_, e_ = int(rp[j_]), int(rp[j_ + 1])
            assert np.array_equal(h_.nbr, nb[b_:e_]) and np.array_equal(h_.prefix_w, pw[b_:e_]), \
                "device generator differs from the host generator at node %d" % int(need[j_])
        checked, _distinct = check_fanout_step(G, O.OracleGraph, O.CSR, GRAPH_SEED, 2 * last, r_sel, sub_n, sub_w,
                                               sub_t, FANOUT, default_node, args.nodes)
        del sub_n, sub_w, sub_t

    # ---- roofline leg: the launches of a step, phase by phase, HIP events on
    # the stream the kernels run on.  Hop 1 samples the caller's roots directly;
    # hop 2 counts duplicate roots on device and (when they repeat) samples the
    # distinct ones once and expands.  The dominant kernel is K1
    # (SampleNeighborPivotKernel): its algorithmic bytes are SURVEY 8(d)'s
    # per-root / per-edge figure summed over the roots it actually processes.
    roofline = None
    fused = not sharded and "27=0" not in args.tuning.split(",")
    if not sharded and not fused:
        raise SystemExit("bench.py: --tuning 27=0 (hop-by-hop fanout) has no roofline leg any more; "
                         "use tools/ab_key.py for that A/B")
    if rank == 0 and fused:
        # The step is ONE kernel (fanout_local.h: hop 1, the duplicate children found inside
        # the wave, hop 2 once per distinct child, the rows streamed out): the roofline object
        # is that launch.  Algorithmic bytes = SURVEY 8(d), evaluated on the roots the step
        # really has: K1 over the batch (hop 1) + K1 over the DISTINCT hop-2 roots (counted
        # globally, as the reference's ID_UNIQUE would - the kernel itself samples ~1.3x as
        # many, duplicates across waves) + 8 + 4 per hop-2 input id (dedup) + 16 per
        # expanded output edge (gather).
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        et1 = (C.c_int32 * 1)(0)
        layers = len(FANOUT)
        cnt_a = (C.c_int32 * layers)(*FANOUT)
        et_a = (C.c_int32 * layers)(*([0] * layers))
        r = roots[n_steps - 1].contiguous()
        o_n, o_w, o_t, m = [], [], [], r.numel()
        for c in FANOUT:
            m *= c
            o_n.append(torch.empty(m, dtype=torch.int64, device=dev))
            o_w.append(torch.empty(m, dtype=torch.float32, device=dev))
            o_t.append(torch.empty(m, dtype=torch.int32, device=dev))
        wsz = int(L.euler_gpu_sample_fanout_workspace(r.numel(), cnt_a, layers))
        fws = torch.empty(max(wsz, 16), dtype=torch.uint8, device=dev)
        pn = (C.c_void_p * layers)(*[t.data_ptr() for t in o_n])
        pw_ = (C.c_void_p * layers)(*[t.data_ptr() for t in o_w])
        pt = (C.c_void_p * layers)(*[t.data_ptr() for t in o_t])
        torch.cuda.synchronize()
        ms = C.c_float(0)
        _lib.check(L.euler_gpu_time_sample_fanout(
            G._h, st, GRAPH_SEED, C.c_void_p(r.data_ptr()), r.numel(), et_a, 1, cnt_a, layers,
            default_node, pn, pw_, pt, C.c_void_p(fws.data_ptr()), 20, C.byref(ms)))

        kernel_name = (L.euler_gpu_last_fanout_kernel() or b"").decode() or "SampleFanoutLeanKernel"

        def algo_bytes(x, cnt):
            b = C.c_double(0)
            _lib.check(L.euler_gpu_sample_neighbor_algo_bytes(
                G._h, st, C.c_void_p(x.data_ptr()), x.numel(), et1, 1, cnt, C.byref(b)))
            return b.value
        hop2_roots = o_n[0]
        uniq2 = torch.unique(hop2_roots).contiguous()
        b1 = algo_bytes(r, FANOUT[0])
        b2 = algo_bytes(uniq2, FANOUT[1])
        n2 = hop2_roots.numel()
        b_dedup = 12.0 * n2
        b_gather = 16.0 * n2 * FANOUT[1]
        total_b = b1 + b2 + b_dedup + b_gather
        achieved = total_b / (ms.value * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        pmc_rec = {}
        if os.path.exists(pmc):
            try:
                pmc_rec = json.load(open(pmc))
                if pmc_rec.get("batch") == B and pmc_rec.get("nodes") == args.nodes and \
                        pmc_rec.get("kernel") == kernel_name:
                    traffic = pmc_rec.get("hbm_bytes_per_launch")
                else:
                    pmc_rec = {}
            except Exception:
                traffic = None
                pmc_rec = {}
        one_ms = (float(np.median(one_stream)) / args.steps * 1e3) if one_stream else None
        rd_req = pmc_rec.get("read_requests_per_launch")
        roofline = {
            "kernel": kernel_name, "bound": "hbm",
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
            "traffic_note": "profiles/pmc_latest.json: TCC_EA0_RDREQ x 128 B + TCC_EA0_WRREQ x 64 B per "
                            "launch (separate rocprofv3 --pmc pass; a read request moves a 128-byte line, "
                            "FETCH_SIZE tallies it at 64 B - calibration in profiles/r2_pmc_summary.json)",
            "algorithmic_bytes_per_launch": round(total_b, 1),
            "avg_launch_ms": round(ms.value, 4),
            # what the three-stream headline must not be mistaken for: the same
            # bytes over ONE caller's step time (enqueue gaps included), and over the headline's
            "one_stream_frac": (round(total_b / (one_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if one_ms else None),
            "headline_streams_frac": round(total_b / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
            # 128-byte lines the launch reads from the fabric (TCC_EA0_RDREQ of the same --pmc pass as
            # `traffic`) per second of the launch: the chip completes ~48 G random lines/s when every
            # line is ONE request and 19.5 G when a lane makes four requests for its line, as a
            # draw here does (tools/ubench_block.hip, profiles/r6_ubench_block.txt)
            "read_lines_per_s": (round(rd_req / (ms.value * 1e-3), 1) if rd_req else None),
            "launches_per_step": [{
                "roots": int(r.numel()), "fanout": FANOUT, "hop2_roots": int(n2),
                "hop2_distinct_roots": int(uniq2.numel()),
                "k1_hop1_algorithmic_bytes": b1, "k1_hop2_distinct_algorithmic_bytes": b2,
                "dedup_algorithmic_bytes": b_dedup, "gather_algorithmic_bytes": b_gather}],
            "note": "the whole step is this one launch, timed alone on one stream with HIP events "
                    "(euler_gpu_time_sample_fanout, 20 calls); bytes = SURVEY 8(d) summed over the "
                    "step: K1 of hop 1 over the batch, K1 of hop 2 over the globally distinct hop-2 "
                    "roots, 8 + 4 per hop-2 input id, 16 per expanded output edge",
        }
        if args.large_batch > 0 and args.large_batch != B:
            # "B >= 1M per launch for the HBM-roofline run": the same launch
            # over 1 048 576 roots (8 rounds of the chip instead of 6.4: the tail and the launch
            # are a smaller share), bytes counted the same way
            try:
                BL = args.large_batch
                gl = torch.Generator(device=dev); gl.manual_seed(555)
                rl = torch.randint(1, args.nodes + 1, (BL,), generator=gl, device=dev, dtype=torch.int64)
                ol_n, ol_w, ol_t, m = [], [], [], BL
                for c in FANOUT:
                    m *= c
                    ol_n.append(torch.empty(m, dtype=torch.int64, device=dev))
                    ol_w.append(torch.empty(m, dtype=torch.float32, device=dev))
                    ol_t.append(torch.empty(m, dtype=torch.int32, device=dev))
                wl = torch.empty(max(int(L.euler_gpu_sample_fanout_workspace(BL, cnt_a, layers)), 16),
                                 dtype=torch.uint8, device=dev)
                pln = (C.c_void_p * layers)(*[t.data_ptr() for t in ol_n])
                plw = (C.c_void_p * layers)(*[t.data_ptr() for t in ol_w])
                plt_ = (C.c_void_p * layers)(*[t.data_ptr() for t in ol_t])
                msl = C.c_float(0)
                torch.cuda.synchronize()
                _lib.check(L.euler_gpu_time_sample_fanout(
                    G._h, st, GRAPH_SEED, C.c_void_p(rl.data_ptr()), BL, et_a, 1, cnt_a, layers,
                    default_node, pln, plw, plt_, C.c_void_p(wl.data_ptr()), 5, C.byref(msl)))
                u2 = torch.unique(ol_n[0]).contiguous()
                nl2 = ol_n[0].numel()
                tb = algo_bytes(rl, FANOUT[0]) + algo_bytes(u2, FANOUT[1]) + 12.0 * nl2 + 16.0 * nl2 * FANOUT[1]
                # the same 64-root oracle check on this launch's output
                chk = None
                if not args.no_check:
                    sel = np.random.default_rng(3).choice(BL, 64, replace=False)
                    st_t = torch.as_tensor(sel).to(dev)
                    r0 = rl[st_t].cpu().numpy()
                    h1 = ol_n[0].reshape(BL, FANOUT[0])[st_t].cpu().numpy()
                    h2 = ol_n[1].reshape(BL, -1)[st_t].cpu().numpy()
                    need_l = np.concatenate([r0, h1.reshape(-1)])
                    OGl = _oracle_rows(G, p, need_l[(need_l >= 1) & (need_l <= args.nodes)], 1)
                    # euler_gpu_time_sample_fanout draws iteration it with call ids it * layers + h
                    onl, _, _ = OGl.sample_fanout(GRAPH_SEED, 4 * layers, r0, et, FANOUT, default_node)
                    assert np.array_equal(onl[0], h1.reshape(-1)) and np.array_equal(onl[1], h2.reshape(-1)), \
                        "B = 1M launch: sampled ids differ from the oracle"
                    chk = int(64 * 275)
                roofline["large_batch"] = {
                    "roots": BL, "avg_launch_ms": round(msl.value, 4),
                    "algorithmic_bytes_per_launch": round(tb, 1),
                    "achieved": round(tb / (msl.value * 1e-3) / 1e9, 2), "unit": "GB/s",
                    "frac": round(tb / (msl.value * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "edges_per_s": BL * 275 / (msl.value * 1e-3),
                    "hop2_distinct_roots": int(u2.numel()), "parity_checked_edges": chk}
                del ol_n, ol_w, ol_t, wl, rl, u2
                torch.cuda.empty_cache()
            except Exception as e:          # a side measurement must not fail the bench
                roofline["large_batch"] = {"error": repr(e)}
    if not fused:
        # The sharded step's own launches (every rank): the owners' pass of a hop is
        # euler_gpu_sample_neighbor_packed over the DISTINCT ids a rank is asked for - K1
        # (SampleNeighborPivotKernel, one sample per lane, wire rows written directly).  Timed
        # alone with HIP events on its stream, over the ids this rank's own batch asks for
        # mapped onto ids this rank owns (a rank is asked for as many as it asks, in expectation):
        # hop 1 = the distinct roots, hop 2 = the distinct hop-1 samples.  Bytes = SURVEY 8(d)'s
        # K1 formula over exactly those ids.
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        et1 = (C.c_int32 * 1)(0)

        def own(x):
            if world > 1 and sharded:      # a shard owns ids == rank (mod world)
                x = (x // world) * world + (rank if rank else world)
                x = torch.clamp(x, max=args.nodes - world)
            return torch.unique(x).contiguous()

        def algo_bytes(x, cnt):
            b = C.c_double(0)
            _lib.check(L.euler_gpu_sample_neighbor_algo_bytes(
                G._h, st, C.c_void_p(x.data_ptr()), x.numel(), et1, 1, cnt, C.byref(b)))
            return b.value
        hop_ids = [own(roots[n_steps - 1]), own(out[0][1].reshape(-1))]
        k1_ms, k1_bytes, phases = [], [], []
        for h, x in enumerate(hop_ids):
            cnt = FANOUT[h]
            ms_h = _events(lambda: G.sample_neighbor_packed(x, [0], cnt, default_node, call_id=h), 10)
            kb = algo_bytes(x, cnt)
            k1_ms.append(ms_h); k1_bytes.append(kb)
            phases.append({"hop": h, "distinct_ids_sampled": int(x.numel()), "count": cnt,
                           "k1_ms": round(ms_h, 4), "k1_algorithmic_bytes": kb,
                           "frac": round(kb / (ms_h * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
        # the requester's side of hop 2, for the record: expansion of the answers to positions
        try:
            from euler_amd import ops as _ops
            n2 = out[0][1].numel()
            rows2 = G.sample_neighbor_packed(hop_ids[1], [0], FANOUT[1], default_node, call_id=1)
            pos2 = torch.randint(0, hop_ids[1].numel(), (n2,), device=dev, dtype=torch.int32)
            ms_x = _events(lambda: _ops.expand_packed(pos2, rows2, FANOUT[1], 0), 10)
            xb = 16.0 * n2 * FANOUT[1] + 4.0 * n2 + rows2.numel() * 4.0
            phases.append({"hop": 1, "expand_ms": round(ms_x, 4), "expand_algorithmic_bytes": xb,
                           "expand_GBps": round(xb / (ms_x * 1e-3) / 1e9, 1)})
            del rows2, pos2
        except Exception as e:
            phases.append({"expand_error": repr(e)})
        achieved = sum(k1_bytes) / (sum(k1_ms) * 1e-3) / 1e9
        per_rank = None
        if world > 1:
            t = torch.tensor([sum(k1_bytes), sum(k1_ms)], device=wire_dev, dtype=torch.float64)
            allt = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            per_rank = [{"rank": r_, "frac": round(float(x_[0]) / (float(x_[1]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "k1_ms": round(float(x_[1]), 4)} for r_, x_ in enumerate(allt)]
            achieved = float(np.mean([x_["frac"] for x_ in per_rank])) * HBM_PEAK_GBS
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_sharded_latest.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("batch") == B and rec.get("nodes") == args.nodes and \
                        "SampleNeighborPivotKernel" in rec.get("kernel", ""):
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        if rank == 0:
            roofline = {
                "kernel": "SampleNeighborPivotKernel (owners' pass of a sharded hop, packed wire rows)",
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "traffic_note": "profiles/pmc_sharded_latest.json: TCC_EA0_RDREQ x 128 B + WRITE_SIZE per "
                                "launch of this kernel in the sharded step (separate rocprofv3 --pmc "
                                "passes over tools/sharded_one.py), mean of the two hops' launches; null when "
                                "no such file matches this batch / graph",
                "algorithmic_bytes_per_launch": round(sum(k1_bytes) / len(k1_bytes), 1),
                "avg_launch_ms": round(sum(k1_ms) / len(k1_ms), 4),
                "launch_ms": [round(x, 4) for x in k1_ms],
                "launches_per_step": phases,
                "per_rank": per_rank,
                "note": "the K1 launches of one sharded step, each timed alone on one stream with HIP "
                        "events (10 calls): hop 1 over the distinct roots, hop 2 over the distinct hop-1 "
                        "samples; bytes = SURVEY 8(d)'s K1 formula over those ids; N ranks: the mean of the "
                        "ranks' fractions, per_rank lists them",
            }

    small = None
    if rank == 0 and world == 1 and not sharded and not args.no_small_batch:
        try:
            small = latency_small_batch(G, L, _lib, args.nodes, default_node)
        except Exception as e:              # a side measurement must not fail the bench
            small = {"error": str(e)}
    secondary = None
    if rank == 0 and world == 1 and not sharded and args.workload == "metric" and not args.no_secondary:
        secondary = secondary_legs(args, G, p)
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        # (N ranks: rank 0 alone, after the timed region - the other ranks wait in the closing
        # barrier; the reference's CPU sampler does not get faster with more GPUs)
        cpu = cpu_baseline(args)

    if rank == 0:
        line = {
            "metric": ("sampled edges/sec (whole node), 2-hop fanout=[25,10], "
                       "100M-node power-law graph") if args.workload == "metric" else
                      "sampled edges/sec, uniform SampleNeighbor fanout=[25,10], ogbn-products-shaped "
                      "CSR in one MI355X's HBM (BASELINE configs[1])",
            "value": value, "unit": "sampled edges/s",
            "n_gpus": min(world, visible),
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": ("synthetic RMAT-marginal power-law graph, %d nodes / "
                             "%d edges (min degree 1), f32 weights uniform [0.5,8), "
                             "weighted CDF-inversion SampleNeighbor, fanout [25,10], "
                             "%d uniform-random roots per step per GPU, TF dense "
                             "layout" if weighted else
                             "products: synthetic power-law graph with ogbn-products' node / edge "
                             "counts, %d nodes / %d edges, all weights 1.0 (the draw is edge "
                             "floor(u * deg): no search), fanout [25,10], %d uniform-random roots "
                             "per step, TF dense layout") % (args.nodes, G.num_edges * 1 if world == 1
                                                            else args.edges, B),
                "roots_per_step_per_gpu": B, "fanout": FANOUT,
                "graph_bytes_per_gpu": G.device_bytes,
                "graph_build_s": round(build_s, 2),
                "partitioning": ("none" if not sharded else
                                 "sharded sampler on 1 rank, %d minibatches in flight" % args.pipeline)
                                if world == 1 else
                                ("replicas: every rank holds the whole graph, no exchange" if replicas else
                                 "hash owner(id)=id%%%d, all-to-all per hop, %d minibatches "
                                 "in flight" % (world, args.pipeline)),
                "replicas": replicas_leg,
                "parity_checked_edges": checked,
                "small_batch": small,
                "sustained": sustained,
                "secondary": secondary,
                "latency_B1024_us": (small or {}).get("latency_B1024_us"),
                "streams": 1 if sharded else max(1, args.streams),
                "tuning": args.tuning or None,
                "one_stream_ms_per_step": (round(float(np.median(one_stream)) / args.steps * 1e3, 4)
                                           if (not sharded and one_stream) else None),
                "repeats": len(rep_secs),
                "repeat_ms_per_step": [round(x / args.steps * 1e3, 4) for x in rep_secs],
                "ranks": world,
                "transport": (None if not (sharded or world > 1) else
                              "RCCL (torch.distributed nccl backend), %d ranks in the communicator%s"
                              % (dist.get_world_size(), "" if sharded else "; barriers and the timing "
                                 "reduction only (replicas)")
                              if backend == "nccl" else
                              "gloo, host-staged: %d ranks share %d GPU(s) - functional check, "
                              "NOT a scaling number" % (world, visible)),
                "exchanged_bytes_per_step": exchanged,
            },
            "roofline": roofline, "cpu_baseline": cpu,
            # (top level, where a parser that drops `config` still sees them: VERDICT r5 #9)
            "sustained": sustained,
            "repeat_ms_per_step": {"min": round(min(rep_secs) / args.steps * 1e3, 4),
                                   "median": round(float(np.median(rep_secs)) / args.steps * 1e3, 4),
                                   "max": round(max(rep_secs) / args.steps * 1e3, 4), "repeats": len(rep_secs)},
            "parity_checked_edges": checked,
        }
        _emit(line)
    _teardown(sharded or world > 1)


if __name__ == "__main__":
    main()
