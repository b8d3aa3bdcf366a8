"""Where the duplicate-root path starts to pay: hop-2 roots of batches of
different sizes, sampled directly vs through the duplicate path.

  python tools/ab_dedup_threshold.py"""
import sys, json, ctypes as C
sys.path.insert(0, '.')
import torch, euler_amd
from euler_amd import _lib
L = _lib.lib()
N = 100_000_000
G = euler_amd.Graph.synthetic(euler_amd.synth_params(20240521, N, 1_000_000_000, weighted=True))
G.set_seed(20240521)
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
et1 = (C.c_int32 * 1)(0)
L.euler_gpu_set_tuning(5, 2)           # the duplicate path whatever the size
res = {}
for B in (256, 512, 1024, 2048, 4096, 8192, 16384, 32768):
    roots = torch.randint(1, N + 1, (B,), generator=gen, device='cuda')
    hop1 = G.sample_neighbor(roots, [0], 25, N + 1, call_id=0, dedup=False)[0].reshape(-1).contiguous()
    n, cnt = hop1.numel(), 10
    oid = torch.empty(n * cnt, dtype=torch.int64, device='cuda')
    ow = torch.empty(n * cnt, dtype=torch.float32, device='cuda')
    ot = torch.empty(n * cnt, dtype=torch.int32, device='cuda')
    row = {}
    for dedup in (0, 1):
        ms3 = (C.c_float * 3)(); nu = C.c_int64(-1)
        for rep in range(2):
            _lib.check(L.euler_gpu_time_sample_neighbor_phases(
                G._h, st, 20240521, C.c_void_p(hop1.data_ptr()), n, et1, 1, cnt, _lib.LAYOUT_TF,
                dedup, C.c_void_p(oid.data_ptr()), C.c_void_p(ow.data_ptr()),
                C.c_void_p(ot.data_ptr()), 30, ms3, C.byref(nu)))
        row['dedup=%d' % dedup] = round(sum(ms3), 4)
        if dedup:
            row['distinct_frac'] = round(nu.value / n, 3)
    res['hop-2 roots %d' % n] = row
L.euler_gpu_set_tuning(5, 1)
print(json.dumps(res, indent=1))
