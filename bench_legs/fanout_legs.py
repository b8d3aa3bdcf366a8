"""The metric step on other graph shapes (products-shaped uniform, hashed ids + type groups) and
its (unique rows, index) form."""
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

__all__ = ['run_products_leg', 'run_hashed_leg', 'run_unique_leg']


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
