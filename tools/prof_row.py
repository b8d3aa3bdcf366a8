"""Row-kernel decomposition on the metric's first hop (131 072 roots x 25), timed
with HIP events inside the library (euler_gpu_time_sample_neighbor_phases), under
the ablations of tuning key 2 (1 = no draws, 2 = no write phase, 4 = no Philox);
run it under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os, sys, json, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, euler_amd
from euler_amd import _lib
L = _lib.lib()
N = 100_000_000
p = euler_amd.synth_params(20240521, N, 10 * N, weighted=True)
G = euler_amd.Graph.synthetic(p)
G.set_seed(20240521)
B = 131072
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
roots = torch.randint(1, N + 1, (B,), generator=gen, device='cuda')
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


res = {}
for row in (1, 0):
    L.euler_gpu_set_tuning(19, row)
    for ab in ((0, 1, 2, 3, 4, 7) if row else (0,)):
        L.euler_gpu_set_tuning(2, ab)
        k1_us(roots, 25, 3)
        res["count25 row=%d ablate=%d" % (row, ab)] = k1_us(roots, 25)
    L.euler_gpu_set_tuning(2, 0)
    res["count10 row=%d" % row] = k1_us(roots, 10)
    res["count25 B=1024 row=%d" % row] = k1_us(roots[:1024].contiguous(), 25, 50)
L.euler_gpu_set_tuning(2, 0); L.euler_gpu_set_tuning(19, 1)
print(json.dumps(res))
