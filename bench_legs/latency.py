"""Small-batch latency leg (B = 1 024 from C, several minibatches per launch, streams)."""
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

__all__ = ['latency_small_batch']


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
