"""Inline rows (tuning key 26) on the metric's first hop: the lane-per-root kernel
(key 19 = 2) and the lane-per-sample kernel (key 19 = 0), alone, HIP-event timed
inside the library; then the whole 2-hop step on one and on two streams."""
import os, sys, json, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, euler_amd
from euler_amd import _lib
L = _lib.lib()
N = 100_000_000
L.euler_gpu_set_tuning(26, 1)      # build the inline lines
G = euler_amd.Graph.synthetic(euler_amd.synth_params(20240521, N, 10 * N, weighted=True))
G.set_seed(20240521)
B = 131072
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
roots = torch.randint(1, N + 1, (8, B), generator=gen, device='cuda')
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
et1 = (C.c_int32 * 1)(0)


def k1_us(r, cnt, iters=20):
    n = r.numel()
    oid = torch.empty(n * cnt, dtype=torch.int64, device='cuda')
    ow = torch.empty(n * cnt, dtype=torch.float32, device='cuda')
    ot = torch.empty(n * cnt, dtype=torch.int32, device='cuda')
    ms3 = (C.c_float * 3)()
    nu = C.c_int64(-1)
    _lib.check(L.euler_gpu_time_sample_neighbor_phases(
        G._h, st, 20240521, C.c_void_p(r.data_ptr()), n, et1, 1, cnt, _lib.LAYOUT_TF,
        0, C.c_void_p(oid.data_ptr()), C.c_void_p(ow.data_ptr()),
        C.c_void_p(ot.data_ptr()), iters, ms3, C.byref(nu)))
    return round(ms3[1] * 1e3, 1)


def step_ms(n_streams, steps=24):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    def loop(k):
        for i in range(k):
            with torch.cuda.stream(streams[i % n_streams]):
                G.sample_fanout(roots[i % 8], [[0], [0]], [25, 10], N + 1, call_id=2 * i)
    loop(8); torch.cuda.synchronize()
    best = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        loop(steps)
        for s in streams: s.synchronize()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / steps)
    best.sort()
    return round(best[2], 4)


res = {}
for inline in (1, 0, 1, 0):
    L.euler_gpu_set_tuning(26, inline)
    for row in (2, 0):
        L.euler_gpu_set_tuning(19, row)
        k1_us(roots[0], 25, 3)
        res.setdefault("hop1 inline=%d row=%d us" % (inline, row), []).append(k1_us(roots[0], 25))
    L.euler_gpu_set_tuning(19, 1)
    for ns in (1, 2):
        res.setdefault("step inline=%d streams=%d ms" % (inline, ns), []).append(step_ms(ns))
L.euler_gpu_set_tuning(26, 0)
print(json.dumps(res))
