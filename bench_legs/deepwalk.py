"""BASELINE configs[3]: DeepWalk / node2vec walks, unsharded and sharded."""
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
from .cpu import cpu_walk_cell

__all__ = ['run_deepwalk']


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
    import gc
    gc.collect(); gc.freeze(); gc.disable()      # a gen-2 collection costs ~40 ms: one repeat in five at 2 x the others
    reps = []
    for _rep in range(max(1, args.repeats)):
        _sync_ranks(wire)
        t0 = time.perf_counter()
        for i in range(args.warmup, n_steps):
            walks = walk(starts[i], et, 1.0, 1.0, LEN * i)
        _sync_ranks(wire)
        reps.append(time.perf_counter() - t0)
    gc.enable()
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
                              "walkers of one rank; time = the call, wall clock"}
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
                        "kernel": "N2vListMergedKernel + FullNb* + front end (the whole call)", "bound": "hbm",
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
                       "orchestration": ("euler_gpu_sharded_random_walk (C): levels of merged walkers in slab "
                                         "layout, the whole walk enqueued - host waits per call in walk_stats "
                                         "(tuning key 63 = 0: a wait per step); %d cohorts"
                                         % getattr(S, "walk_cohorts", 0))
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
