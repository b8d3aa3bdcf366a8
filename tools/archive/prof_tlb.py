"""Does page locality bound K1?  The same roots sampled in random order and in
ascending row order (sorted ids): hop 1 (131 072 uniform roots x 25) and the
distinct roots of hop 2 (x 10), lane-per-sample kernels, HIP-event timing."""
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
pass


def k1_us(r, cnt, iters=20):
    n = r.numel()
    oid = torch.empty(n * cnt, dtype=torch.int64, device='cuda')
    ow = torch.empty(n * cnt, dtype=torch.float32, device='cuda')
    ot = torch.empty(n * cnt, dtype=torch.int32, device='cuda')
    ms3 = (C.c_float * 3)()
    nu = C.c_int64(-1)
    for it in (3, iters):
        _lib.check(L.euler_gpu_time_sample_neighbor_phases(
            G._h, st, 20240521, C.c_void_p(r.data_ptr()), n, et1, 1, cnt, _lib.LAYOUT_TF,
            0, C.c_void_p(oid.data_ptr()), C.c_void_p(ow.data_ptr()),
            C.c_void_p(ot.data_ptr()), it, ms3, C.byref(nu)))
    return round(ms3[1] * 1e3, 1)


res = {}
out = G.sample_fanout(roots, [[0], [0]], [25, 10], N + 1, call_id=0)
uq = torch.unique(out[0][1])
uq = uq[uq <= N]
perm = torch.randperm(uq.numel(), device='cuda')
res["hop1 random"] = k1_us(roots, 25)
res["hop1 sorted"] = k1_us(torch.sort(roots)[0].contiguous(), 25)
L.euler_gpu_set_tuning(4, 0)          # one sample per lane, as the pass over the distinct roots
res["hop2 distinct (%d) random" % uq.numel()] = k1_us(uq[perm].contiguous(), 10)
res["hop2 distinct sorted"] = k1_us(uq.contiguous(), 10)
# bucketed by the top 8 bits of the row only
b = (uq * 256 // (N + 1))
shuf = uq[perm]
key = (shuf * 256 // (N + 1))
res["hop2 distinct 256 buckets"] = k1_us(shuf[torch.sort(key, stable=True)[1]].contiguous(), 10)
key = (shuf * 4096 // (N + 1))
res["hop2 distinct 4096 buckets"] = k1_us(shuf[torch.sort(key, stable=True)[1]].contiguous(), 10)
L.euler_gpu_set_tuning(4, 1)
print(json.dumps(res))
