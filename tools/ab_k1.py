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
out = G.sample_fanout(roots, [[0],[0]], [25,10], 100_000_001, call_id=0)
hop2_roots = out[0][1].contiguous()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
et1 = (C.c_int32*1)(0)
res = {}
for variant in ((6,0),):
    L.euler_gpu_set_tuning(0, variant[0]); L.euler_gpu_set_tuning(3, variant[1])
    L.euler_gpu_set_tuning(5, 0)
    for pair in (0, 1, 0, 1):
     L.euler_gpu_set_tuning(6, pair)
     for name, r, cnt in (('hop1', roots, 25), ('hop2', hop2_roots, 10)):
        n = r.numel()
        oid = torch.empty(n*cnt, dtype=torch.int64, device='cuda'); ow = torch.empty(n*cnt, dtype=torch.float32, device='cuda'); ot = torch.empty(n*cnt, dtype=torch.int32, device='cuda')
        ms = C.c_float(0)
        _lib.check(L.euler_gpu_time_sample_neighbor(G._h, st, 20240521, C.c_void_p(r.data_ptr()), n, et1, 1, cnt, 1, C.c_void_p(oid.data_ptr()), C.c_void_p(ow.data_ptr()), C.c_void_p(ot.data_ptr()), 10, C.byref(ms)))
        res.setdefault((variant,pair,name), []).append(round(ms.value,4))
print({str(k): v for k, v in res.items()})
