"""SageDataFlow block construction leg."""
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

__all__ = ['run_sage_leg']


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
