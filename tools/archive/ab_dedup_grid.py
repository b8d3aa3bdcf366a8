"""Cost of the gated (empty) launches of the duplicate-root path and of the
workgroup cap of K1 on the pass over the distinct roots.

  python tools/ab_dedup_grid.py

hop 1 (131 072 distinct roots) with dedup on vs off: the difference in the K1
phase is the all-empty gated launch, the expand phase is an all-empty launch.
hop 2 with dedup on under several workgroup caps (tuning key 3)."""
import sys, json, ctypes as C
sys.path.insert(0, '.')
import torch, euler_amd
from euler_amd import _lib
L = _lib.lib()
p = euler_amd.synth_params(20240521, 100_000_000, 1_000_000_000, weighted=True)
G = euler_amd.Graph.synthetic(p)
G.set_seed(20240521)
B = 131072
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
roots = torch.randint(1, 100_000_001, (B,), generator=gen, device='cuda')
out = G.sample_fanout(roots, [[0], [0]], [25, 10], 100_000_001, call_id=0)
hop2 = out[0][1].contiguous()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
et1 = (C.c_int32 * 1)(0)


def phases(r, cnt, dedup, iters=20):
    n = r.numel()
    oid = torch.empty(n * cnt, dtype=torch.int64, device='cuda')
    ow = torch.empty(n * cnt, dtype=torch.float32, device='cuda')
    ot = torch.empty(n * cnt, dtype=torch.int32, device='cuda')
    ms3 = (C.c_float * 3)()
    nu = C.c_int64(-1)
    _lib.check(L.euler_gpu_time_sample_neighbor_phases(
        G._h, st, 20240521, C.c_void_p(r.data_ptr()), n, et1, 1, cnt, _lib.LAYOUT_TF,
        dedup, C.c_void_p(oid.data_ptr()), C.c_void_p(ow.data_ptr()),
        C.c_void_p(ot.data_ptr()), iters, ms3, C.byref(nu)))
    return [round(x, 4) for x in ms3], nu.value


res = {}
L.euler_gpu_set_tuning(5, 2)          # dedup path regardless of n
for rep in range(2):
    res.setdefault('hop1 dedup=0', []).append(phases(roots, 25, 0))
    res.setdefault('hop1 dedup=1 (gated launches empty)', []).append(phases(roots, 25, 1))
L.euler_gpu_set_tuning(5, 1)
for cap in (0, 16384, 12288, 8192, 4096, 2048, 0):
    L.euler_gpu_set_tuning(3, cap)
    res.setdefault('hop2 cap=%d' % cap, []).append(phases(hop2, 10, 1))
L.euler_gpu_set_tuning(3, 0)
print(json.dumps(res, indent=1))
