"""BASELINE configs[4]: heterogeneous typed sampling + aggregation."""
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
from .cpu import cpu_hetero_cell

__all__ = ['run_hetero']


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
    n_streams = max(1, args.streams) if S is None else 1
    side = [torch.cuda.Stream() for _ in range(n_streams)] if n_streams > 1 else None
    # sharded: a hop waits on the host once (the front end's bucket sizes) - K minibatches in flight
    # from one host thread, each on its own sampler and stream, advanced hop by hop in a fixed
    # order (run_interleaved, as the sharded metric step does): while the host waits for one, the
    # GPU runs the others' owners' passes, expansions and aggregations
    # (one rank: 0.386-0.394 ms per step with 1, 0.342-0.42 with 2, 0.341-0.344 with 3, 0.348-0.354 with 4)
    K_fl = max(1, min(args.pipeline, 3, args.steps)) if S is not None and not args.hetero_separate else 1
    if K_fl > 1:
        from euler_amd.distributed import gpu_sharded_sampler, run_interleaved
        samplers = [S] + [gpu_sharded_sampler(G, partitions=world) for _ in range(K_fl - 1)]
        fl_streams = [torch.cuda.Stream() for _ in range(K_fl)]

        def job(i):
            def gen():
                outs = yield from samplers[i % K_fl].sample_neighbor_sets_steps(
                    roots[i], type_sets, CNT, N + 1, call_id=3 * i)
                return [ops.gather_segment_reduce("mean", feat, o[0].reshape(-1), B, count=CNT) for o in outs]
            return gen()

    def loop(first, last):
        if K_fl > 1:
            torch.cuda.synchronize()
            kept = [None]
            run_interleaved(lambda j: job(first + j), last - first, K_fl,
                            enter=lambda k: torch.cuda.stream(fl_streams[k]),
                            on_result=lambda j, v: kept.__setitem__(0, v))
            for st_ in fl_streams:
                st_.synchronize()
            return
        for i in range(first, last):
            if side is None:
                step(i)
            else:
                with torch.cuda.stream(side[i % n_streams]):
                    step(i)

    _sync_ranks(wire)
    loop(0, max(args.warmup, 2 * n_streams))
    _sync_ranks(wire)
    import gc
    gc.collect(); gc.freeze(); gc.disable()      # (a gen-2 collection inside a repeat costs ~40 ms)
    reps = []
    for _rep in range(max(1, args.repeats)):
        _sync_ranks(wire)
        t0 = time.perf_counter()
        loop(args.warmup, n_steps)
        _sync_ranks(wire)
        reps.append(time.perf_counter() - t0)
    gc.enable()
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
        # the step's owners' pass: the three sets as ONE launch writing wire rows
        # (euler_gpu_sample_neighbor_sets_packed); the separate launches stay beside it
        tm = _events(lambda: G.sample_neighbor_sets_packed(x_own, type_sets, CNT, N + 1, call_id=0), 10)
        roof_s = {"kernel": "SampleNeighborSetsLdsKernel (owners' pass of the step's typed sharded hops, the "
                            "three sets in one launch, packed wire rows)", "bound": "hbm",
                  "achieved": round(tb / tm / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": round(tb / tm / 1e6 / HBM_PEAK_GBS, 4), "traffic": None,
                  "algorithmic_bytes_per_launch": tb, "avg_launch_ms": round(tm, 4),
                  "separate_launches": k1s,
                  "note": "rank 0's owners' pass of one step, timed alone with HIP events over the distinct "
                          "roots (separate_launches: one sample_neighbor_packed call per set, what the step "
                          "made before); the aggregation is the unsharded path's (replicas only)"}
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
                                                                           dist.get_world_size()),
                       "minibatches_in_flight": K_fl},
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
