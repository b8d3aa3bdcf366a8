"""The metric step with HOST buffers on both sides (the PCIe-inclusive rate DESIGN 4.3 quotes; never the
headline `value`): (1) through the client boundary - euler::Query / QueryProxy::RunAsyncGremlin fed with the
chain tf_euler/kernels/sample_fanout_op.cc:37-42 builds (roots in a host tensor, every result tensor in
malloc'ed host memory, FillNeighbor layout) - and (2) the C ABI's device step followed by the copy of its
outputs into pinned host memory on the same stream.  100M / 1B metric graph, fanout [25, 10]."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import euler_amd                      # noqa: E402
from euler_amd import _lib            # noqa: E402

K_INT32, K_UINT64 = 2, 7
N, E = int(os.environ.get("HB_NODES", 100_000_000)), int(os.environ.get("HB_EDGES", 1_000_000_000))
FANOUT = [25, 10]


def main():
    L = _lib.lib()
    p = euler_amd.synth_params(20240607, N, E, weighted=True)
    G = euler_amd.Graph.synthetic(p, device=0)
    L.euler_query_set_graph(G._h)
    rng = np.random.default_rng(5)
    out = {"graph": "%d nodes / %d edges" % (N, G.num_edges), "fanout": FANOUT}
    gremlin = "v(nodes)"
    for i in range(2):
        gremlin += ".sampleNB(et_%d,nb_count_%d,-1).as(nb_%d)" % (i, i, i)
    for B in (1024, 131072):
        roots = rng.integers(1, N, B).astype(np.uint64)
        edges = B * (FANOUT[0] + FANOUT[0] * FANOUT[1])
        # ---- (1) the client boundary
        names = ["nodes", "et_0", "nb_count_0", "et_1", "nb_count_1"]
        arrs = [roots, np.array([0], np.int32), np.array([FANOUT[0]], np.int32),
                np.array([0], np.int32), np.array([FANOUT[1]], np.int32)]
        cnts = [B, 1, -1, 1, -1]
        dts = [K_UINT64, K_INT32, K_INT32, K_INT32, K_INT32]
        c_names = (C.c_char_p * 5)(*[x.encode() for x in names])
        c_dts = (C.c_int32 * 5)(*dts)
        c_cnt = (C.c_int64 * 5)(*cnts)
        c_ptr = (C.c_void_p * 5)(*[a.ctypes.data for a in arrs])
        # every result tensor of the query lands in host memory; the harness copies only the small
        # one it asks for (a C++ host reads the tensors in place)
        res = np.zeros(B * 2, np.int32)
        L.euler_query_run.restype = C.c_int64

        def q():
            rc = L.euler_query_run(gremlin.encode(), 5, c_names, c_dts, c_cnt, c_ptr, b"nb_0:0",
                                   res.ctypes.data_as(C.c_void_p), C.c_int64(res.nbytes))
            assert rc == res.nbytes, rc
        for _ in range(16):           # 8 proxy threads, each with its own stream and staging arena
            q()
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); q(); ts.append(time.perf_counter() - t0)
        ts.sort()
        out["query_shim_B%d" % B] = {"ms": round(ts[len(ts) // 2] * 1e3, 3), "min_ms": round(ts[0] * 1e3, 3),
                                     "edges_per_s": edges / ts[len(ts) // 2]}
        # ---- (1b) eight caller threads keep queries in flight (the reference's client pool,
        # client/query_proxy.cc:205-210): aggregate rate of the boundary
        import threading
        per = 24 if B <= 4096 else 6
        bufs = [np.zeros(B * 2, np.int32) for _ in range(8)]

        def worker(k):
            for _ in range(per):
                rc = L.euler_query_run(gremlin.encode(), 5, c_names, c_dts, c_cnt, c_ptr, b"nb_0:0",
                                       bufs[k].ctypes.data_as(C.c_void_p), C.c_int64(bufs[k].nbytes))
                assert rc == bufs[k].nbytes, rc
        t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        el = time.perf_counter() - t0
        out["query_shim_8_in_flight_B%d" % B] = {"queries": 8 * per, "ms_per_query": round(el / (8 * per) * 1e3, 4),
                                                 "edges_per_s": edges * 8 * per / el}
        # ---- (2) device step + outputs copied to pinned host memory
        dev_roots = torch.as_tensor(roots.astype(np.int64)).cuda()
        host = None

        def step():
            nonlocal host
            nb, w, t = G.sample_fanout(dev_roots, [[0], [0]], FANOUT, default_node=-1)
            flat = [x.reshape(-1) for x in list(nb[1:]) + list(w) + list(t)]
            if host is None:
                host = [torch.empty(x.shape, dtype=x.dtype).pin_memory() for x in flat]
            for h, x in zip(host, flat):
                h.copy_(x, non_blocking=True)
            torch.cuda.synchronize()
        for _ in range(3):
            step()
        ts = []
        for _ in range(9):
            t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
        ts.sort()
        nbytes = sum(h.numel() * h.element_size() for h in host)
        out["device_step_plus_d2h_B%d" % B] = {"ms": round(ts[len(ts) // 2] * 1e3, 3), "d2h_bytes": nbytes,
                                               "edges_per_s": edges / ts[len(ts) // 2],
                                               "d2h_GBps_if_all_copy": nbytes / ts[len(ts) // 2] / 1e9}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
