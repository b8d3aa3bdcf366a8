"""SampleNode and the DeepWalk minibatch legs."""
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

__all__ = ['run_node_legs']


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
