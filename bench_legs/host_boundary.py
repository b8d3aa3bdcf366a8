"""Secondary leg `host_boundary`: the metric step with HOST buffers on both sides - the PCIe-inclusive
rates DESIGN 4.3 quotes next to the headline (never the headline `value`).  (1) the client boundary
(include/euler_query.h: euler::Query / QueryProxy::RunAsyncGremlin with the chain
tf_euler/kernels/sample_fanout_op.cc:37-42 builds; roots in a host tensor, every result in a host
Tensor, FillNeighbor layout) and (2) the C ABI's device step followed by the copy of its outputs into
pinned host memory on the same stream."""
import ctypes as C
import time

import numpy as np
import torch

K_INT32, K_UINT64 = 2, 7          # euler::DataType (core/framework/types.h:26-39)


def run_host_boundary_leg(args, G, p_g):
    from euler_amd import _lib
    L = _lib.lib()
    B, fanout = int(args.batch), [25, 10]
    edges = B * (fanout[0] + fanout[0] * fanout[1])
    rng = np.random.default_rng(5)
    roots = rng.integers(1, int(args.nodes), B).astype(np.uint64)
    L.euler_query_set_graph(G._h)
    try:
        gremlin = "v(nodes)"
        for i in range(2):
            gremlin += ".sampleNB(et_%d,nb_count_%d,-1).as(nb_%d)" % (i, i, i)
        names = ["nodes", "et_0", "nb_count_0", "et_1", "nb_count_1"]
        arrs = [roots, np.array([0], np.int32), np.array([fanout[0]], np.int32),
                np.array([0], np.int32), np.array([fanout[1]], np.int32)]
        c_names = (C.c_char_p * 5)(*[x.encode() for x in names])
        c_dts = (C.c_int32 * 5)(K_UINT64, K_INT32, K_INT32, K_INT32, K_INT32)
        c_cnt = (C.c_int64 * 5)(B, 1, -1, 1, -1)
        c_ptr = (C.c_void_p * 5)(*[a.ctypes.data for a in arrs])
        # every result tensor of the query lands in host memory; the harness copies only the
        # small one it asks for (a C++ host reads the tensors in place)
        res = np.zeros(B * 2, np.int32)
        L.euler_query_run.restype = C.c_int64

        def q():
            rc = L.euler_query_run(gremlin.encode(), 5, c_names, c_dts, c_cnt, c_ptr, b"nb_0:0",
                                   res.ctypes.data_as(C.c_void_p), C.c_int64(res.nbytes))
            if rc != res.nbytes:
                raise RuntimeError("euler_query_run: %d" % rc)
        for _ in range(16):               # 8 proxy threads, each with its own stream and arena
            q()
        ts = []
        for _ in range(11):
            t0 = time.perf_counter(); q(); ts.append(time.perf_counter() - t0)
        ts.sort()
        q_ms = ts[len(ts) // 2] * 1e3
    finally:
        L.euler_query_set_graph(None)
    dev_roots = torch.as_tensor(roots.astype(np.int64)).to(G.device)
    host = []

    def step():
        nb, w, t = G.sample_fanout(dev_roots, [[0], [0]], fanout, default_node=-1)
        flat = [x.reshape(-1) for x in list(nb[1:]) + list(w) + list(t)]
        if not host:
            host.extend(torch.empty(x.shape, dtype=x.dtype).pin_memory() for x in flat)
        for h, x in zip(host, flat):
            h.copy_(x, non_blocking=True)
        torch.cuda.synchronize()
    for _ in range(3):
        step()
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
    ts.sort()
    s_ms = ts[len(ts) // 2] * 1e3
    nbytes = sum(h.numel() * h.element_size() for h in host)
    return {"value": edges / (q_ms * 1e-3), "unit": "sampled edges/s", "ms_per_step": round(q_ms, 3),
            "workload": "the metric step through euler::Query (host tensors in and out, %d roots, "
                        "fanout [25, 10], %.0f MB of results per step)" % (B, nbytes / 1e6),
            "device_step_plus_copy_to_pinned": {"ms_per_step": round(s_ms, 3),
                                                "value": edges / (s_ms * 1e-3),
                                                "d2h_GBps": round(nbytes / (s_ms * 1e-3) / 1e9, 1)},
            "note": "PCIe-inclusive: bounded by the copy of the results to the host (16 B per sampled "
                    "edge); not comparable with the headline, whose buffers stay in HBM"}
