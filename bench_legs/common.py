"""What every leg of bench.py shares: constants, the deferred JSON line, timing / statistics helpers,
the oracle spot-check helpers, the rank context of a torch.distributed run."""
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

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FANOUT = [25, 10]
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
GRAPH_SEED = 20240521
PREWARM = 16                   # sharded path: extra untimed steps (allocator warm-up)

__all__ = ['_emit', '_teardown', '_stats', '_threaded_rate', '_events', '_oracle_rows', '_mix64_t', '_oracle_sage_blocks', '_rank_ctx', '_max_over_ranks', '_sync_ranks', 'ROOT', 'FANOUT', 'HBM_PEAK_GBS', 'GRAPH_SEED', 'PREWARM', '_defer_lines']


_DEFERRED = None        # a list while a process group is up: the line is printed after its teardown


def _defer_lines():
    """A process group is coming up: keep the JSON line(s) until _teardown."""
    global _DEFERRED
    _DEFERRED = []


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
