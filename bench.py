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
    ap.add_argument("--cpu-nodes", type=int, default=4_000_000,
                    help="graph size of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--pipeline", type=int, default=3,
                    help="sharded path: minibatches in flight, interleaved hop by hop "
                         "from one host thread (each on its own sampler and HIP "
                         "stream): while the host waits for one batch's bucket sizes "
                         "the GPU runs the others' kernels and exchanges (one rank: 0.61 ms "
                         "/ step with 1, 0.47 with 2, 0.46 with 3 or 4)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU sampler (unique / split / all-to-all / "
                         "merge / gather) even on one rank: measures its overhead")
    return ap.parse_args()


def cpu_baseline(args):
    """Reference sampler on the host: same synthetic family, smaller graph
    (the reference's per-node objects cannot hold 100M nodes: SURVEY F8).  Two
    thread counts are timed - 8, the reference's client pool
    (client/query_proxy.cc:209), and 32, where its throughput peaks on this
    host (tools/cpu_scaling.py: 8 -> 133, 32 -> 370, 64 -> 321, 256 -> 223 M
    edges/s on the 1M-node graph) - and the better one is the baseline."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    n = args.cpu_nodes
    po = O.synth_params(GRAPH_SEED, n, 10 * n, weighted=True)
    csr = O.synth_csr(po)
    rng = np.random.default_rng(1)
    batch = 1024
    kind = "port"
    if O.have_ref():
        T = 1
        seg = csr.row_ptr.copy()
        # per-edge weights as f32 differences of the running sums (timing
        # only: the reference's Node::Init re-accumulates them)
        w = csr.prefix_w.copy()
        w[1:] -= csr.prefix_w[:-1]
        starts = csr.row_ptr[:-1]
        w[starts] = csr.prefix_w[starts]
        R = O.RefGraph.build_raw(csr.row_id, seg, csr.nbr, w, T)
        bench = R.bench_fanout
        kind = "reference"
    else:
        bench = O.OracleGraph(csr).bench_fanout
    runs = []
    for threads, iters in ((min(8, cores), 1024), (min(32, cores), 8192)):
        if runs and threads == runs[0][0]:
            continue
        roots = rng.integers(1, n + 1, batch * iters).astype(np.uint64)
        bench(GRAPH_SEED, roots[:batch * 4], batch, 4, FANOUT, threads)   # warm-up
        secs, edges = bench(GRAPH_SEED, roots, batch, iters, FANOUT, threads)
        runs.append((threads, iters, secs, edges / secs))
    best = max(runs, key=lambda x: x[3])
    # the GPU on the SAME graph (the >= 10x claim is made on an identical
    # graph; the device generator builds the graph the host generator built)
    same = None
    try:
        import euler_amd
        Gs = euler_amd.Graph.synthetic(euler_amd.synth_params(GRAPH_SEED, n, 10 * n, weighted=True))
        Gs.set_seed(GRAPH_SEED)
        gen = torch.Generator(device="cuda")
        gen.manual_seed(99)
        B = args.batch
        r = torch.randint(1, n + 1, (24, B), generator=gen, device="cuda", dtype=torch.int64)
        for i in range(4):
            Gs.sample_fanout(r[i], [[0], [0]], FANOUT, n + 1, call_id=2 * i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(4, 24):
            Gs.sample_fanout(r[i], [[0], [0]], FANOUT, n + 1, call_id=2 * i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        same = {"value": B * (FANOUT[0] + FANOUT[0] * FANOUT[1]) / dt, "ms_per_step": dt * 1e3,
                "roots_per_step": B, "ratio_to_cpu": B * (FANOUT[0] + FANOUT[0] * FANOUT[1]) / dt / best[3]}
        del Gs
    except Exception as e:                 # the baseline itself must not fail the bench
        same = {"error": str(e)}
    return {"value": best[3], "unit": "sampled edges/s", "cores": best[0],
            "kind": kind, "gpu_same_graph": same,
            "sample": "%d minibatches x %d roots, fanout [25,10] (%.1f s of wall time on %d "
                      "threads), synthetic power-law graph of the same family with %d nodes / "
                      "%d edges (the reference's Node objects for it build in seconds; the "
                      "100M-node graph does not fit them); other thread counts: %s; host has "
                      "%d cores"
                      % (best[1], batch, best[2], best[0], n, len(csr.nbr),
                         ", ".join("%d threads -> %.0f M edges/s" % (r[0], r[3] / 1e6)
                                   for r in runs if r is not best) or "none", cores)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=dev)

    import euler_amd
    from euler_amd import _lib
    L = _lib.lib()

    t0 = time.time()
    p = euler_amd.synth_params(GRAPH_SEED, args.nodes, args.edges, weighted=True)
    G = euler_amd.Graph.synthetic(p, device=local_rank, partitions=world,
                                  shard_index=rank, shards=world)
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
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

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
        t0 = time.perf_counter()
        out = run(args.warmup, n_steps)
        sync()
        elapsed = time.perf_counter() - t0
        gc.enable()
        if trace is not None and rank == 0:
            tail = trace[-(n_steps - args.warmup):]
            print("trace (step, ms since previous job started, device allocs so far):",
                  [(a[0], round((a[1] - b[1]) * 1e3, 2), a[2]) for a, b in zip(tail[1:], tail[:-1])],
                  file=sys.stderr)
    else:
        def step(i):
            return G.sample_fanout(roots[i], et, FANOUT, default_node, call_id=2 * i)

        for i in range(args.warmup):
            out = step(i)
        sync()
        gc.collect(); gc.freeze(); gc.disable()      # a gen-2 collection costs ~40 ms
        t0 = time.perf_counter()
        for i in range(args.warmup, n_steps):
            out = step(i)
        sync()
        elapsed = time.perf_counter() - t0
        gc.enable()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    edges_per_step = B * (FANOUT[0] + FANOUT[0] * FANOUT[1]) * world
    value = edges_per_step * args.steps / elapsed

    # ---- parity spot check at full size: 64 roots of the last step against
    # the oracle fed with the rows exported from HBM (single GPU only)
    checked = None
    if world == 1 and not args.no_check:
        from oracle import oracle as O
        last = n_steps - 1
        sel = np.random.default_rng(0).choice(B, 64, replace=False)
        r0 = roots[last].cpu().numpy()[sel]
        hop1 = out[0][1].reshape(B, FANOUT[0]).cpu().numpy()[sel]
        hop2 = out[0][2].reshape(B, FANOUT[0], FANOUT[1]).cpu().numpy()[sel]
        need = np.unique(np.concatenate([r0, hop1.reshape(-1)])).astype(np.uint64)
        need = need[need <= args.nodes]
        rp, te, nb, pw, tp = G.export_rows(need)
        OG = O.OracleGraph(O.CSR(need, rp, te, nb, pw, tp, 1))
        on, _, _ = OG.sample_fanout(GRAPH_SEED, 2 * last, r0, et, FANOUT, default_node)
        assert np.array_equal(on[0], hop1.reshape(-1)), "hop-1 ids differ from oracle"
        assert np.array_equal(on[1], hop2.reshape(-1)), "hop-2 ids differ from oracle"
        checked = int(len(r0) * 275)

    # ---- roofline leg: the launches of a step, phase by phase, HIP events on
    # the stream the kernels run on.  Hop 1 samples the caller's roots directly;
    # hop 2 counts duplicate roots on device and (when they repeat) samples the
    # distinct ones once and expands.  The dominant kernel is K1
    # (SampleNeighborPivotKernel): its algorithmic bytes are SURVEY 8(d)'s
    # per-root / per-edge figure summed over the roots it actually processes.
    roofline = None
    if rank == 0:
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        et1 = (C.c_int32 * 1)(0)
        iters = 5
        shapes = [(roots[n_steps - 1].contiguous(), FANOUT[0], 0)]
        if world == 1:
            shapes.append((out[0][1].contiguous(), FANOUT[1], 1))
        else:
            shapes.append((torch.randint(1, args.nodes + 1, (B * FANOUT[0],),
                                         generator=gen, device=dev,
                                         dtype=torch.int64), FANOUT[1], 1))

        def algo_bytes(r, cnt):
            b = C.c_double(0)
            _lib.check(L.euler_gpu_sample_neighbor_algo_bytes(
                G._h, st, C.c_void_p(r.data_ptr()), r.numel(), et1, 1, cnt, C.byref(b)))
            return b.value

        k1_ms, k1_bytes, phases = [], [], []
        in_place = None
        if world == 1:
            # the fanout timed in place (hop 1's kernel also enters its ids into
            # hop 2's owner table, so the hops are not independent launches)
            layers = len(FANOUT)
            cnt_a = (C.c_int32 * layers)(*FANOUT)
            et_a = (C.c_int32 * layers)(*([0] * layers))
            r = shapes[0][0]
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
            ms_all = (C.c_float * (3 * layers))()
            nu_all = (C.c_int64 * layers)()
            _lib.check(L.euler_gpu_time_sample_fanout_phases(
                G._h, st, GRAPH_SEED, C.c_void_p(r.data_ptr()), r.numel(), et_a, 1, cnt_a,
                layers, default_node, pn, pw_, pt, C.c_void_p(fws.data_ptr()), iters,
                ms_all, nu_all))
            in_place = ([list(ms_all[3 * h:3 * h + 3]) for h in range(layers)],
                        list(nu_all), [r] + o_n)
        for h, (r, cnt, dedup) in enumerate(shapes):
            if world > 1:      # a shard only owns ids == rank (mod world)
                r = (r // world) * world + (rank if rank else world)
                r = torch.clamp(r, max=args.nodes - world)
            n = r.numel()
            ms3 = (C.c_float * 3)()
            nu = C.c_int64(-1)
            if in_place is None:
                oid = torch.empty(n * cnt, dtype=torch.int64, device=dev)
                ow = torch.empty(n * cnt, dtype=torch.float32, device=dev)
                ot = torch.empty(n * cnt, dtype=torch.int32, device=dev)
            if in_place is not None:
                r = in_place[2][h].contiguous()     # this hop's roots in the timed fanout
                ms3[0], ms3[1], ms3[2] = in_place[0][h]
                nu = C.c_int64(in_place[1][h])
            else:
                _lib.check(L.euler_gpu_time_sample_neighbor_phases(
                    G._h, st, GRAPH_SEED, C.c_void_p(r.data_ptr()), n, et1, 1, cnt,
                    _lib.LAYOUT_TF, dedup, C.c_void_p(oid.data_ptr()),
                    C.c_void_p(ow.data_ptr()), C.c_void_p(ot.data_ptr()), iters, ms3,
                    C.byref(nu)))
            unique_path = dedup == 1 and nu.value >= 0 and nu.value * 4 <= n * 3
            sampled = torch.unique(r) if unique_path else r
            kb = algo_bytes(sampled.contiguous(), cnt)
            # hop chaining: this hop's kernel also does the next hop's 4-byte
            # owner store per id (SURVEY 8(d)'s "dedup: 8 + 4 per input id" -
            # the 4 move here, the next hop's dedup keeps the 8)
            marks_next = in_place is not None and h + 1 < len(shapes) and not unique_path
            premarked = in_place is not None and h > 0
            if marks_next:
                kb += 4.0 * n * cnt
            k1_ms.append(ms3[1])
            k1_bytes.append(kb)
            ph = {"roots": n, "count": cnt, "roots_sampled": int(sampled.numel()),
                  "k1_ms": round(ms3[1], 4), "k1_algorithmic_bytes": kb,
                  "marks_next_hop": bool(marks_next)}
            if unique_path:
                # dedup adds 8 + 4 bytes per input id, the gather 16
                # per expanded output edge (+ the unique rows it reads once)
                eb = 16.0 * n * cnt + 16.0 * nu.value * cnt + 4.0 * n
                ph.update({"dedup_ms": round(ms3[0], 4),
                           "dedup_algorithmic_bytes": (8.0 if premarked else 12.0) * n,
                           "expand_ms": round(ms3[2], 4), "expand_algorithmic_bytes": eb,
                           "expand_GBps": round(eb / (ms3[2] * 1e-3) / 1e9, 1)})
            phases.append(ph)
        achieved = sum(k1_bytes) / (sum(k1_ms) * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("batch") == B and rec.get("nodes") == args.nodes:
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {
            "kernel": "SampleNeighborPivotKernel",
            "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic,
            "algorithmic_bytes_per_launch": round(sum(k1_bytes) / len(k1_bytes), 1),
            "avg_launch_ms": round(sum(k1_ms) / len(k1_ms), 4),
            "launch_ms": [round(x, 4) for x in k1_ms],
            "launches_per_step": phases,
            "note": "K1 launches that do work in one step: hop 1 over the batch, hop 2 "
                    "over the distinct hop-2 roots (duplicates are counted on device and "
                    "their rows expanded by DedupExpandKernel); bytes = SURVEY 8(d) formula "
                    "over the roots each launch processes",
        }

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)

    if rank == 0:
        line = {
            "metric": "sampled edges/sec (whole node), 2-hop fanout=[25,10], "
                      "100M-node power-law graph",
            "value": value, "unit": "sampled edges/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "synthetic RMAT-marginal power-law graph, %d nodes / "
                            "%d edges (min degree 1), f32 weights uniform [0.5,8), "
                            "weighted CDF-inversion SampleNeighbor, fanout [25,10], "
                            "%d uniform-random roots per step per GPU, TF dense "
                            "layout" % (args.nodes, G.num_edges * 1 if world == 1
                                        else args.edges, B),
                "roots_per_step_per_gpu": B, "fanout": FANOUT,
                "graph_bytes_per_gpu": G.device_bytes,
                "graph_build_s": round(build_s, 2),
                "partitioning": ("none" if not sharded else
                                 "sharded sampler on 1 rank, %d minibatches in flight" % args.pipeline)
                                if world == 1 else
                                "hash owner(id)=id%%%d, all-to-all per hop, %d minibatches "
                                "in flight" % (world, args.pipeline),
                "parity_checked_edges": checked,
            },
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
