import sys, json, ctypes as C
sys.path.insert(0, '.')
import torch, euler_amd
from euler_amd import _lib
L = _lib.lib()
L.euler_gpu_debug_k1_phases.restype = C.c_int
p = euler_amd.synth_params(20240521, 100_000_000, 1_000_000_000, weighted=True)
G = euler_amd.Graph.synthetic(p)
G.set_seed(20240521)
B = 131072
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
roots = torch.randint(1, 100_000_001, (B,), generator=gen, device='cuda')
out = G.sample_fanout(roots, [[0],[0]], [25,10], 100_000_001, call_id=0)
hop2 = out[0][1].contiguous()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
names = ['root', 'meta', 'limit', 'philox', 'search', 'final', 'store']
for label, r, cnt in (('hop2', hop2, 10), ('hop1', roots, 25)):
    n = r.numel()
    oid = torch.empty(n*cnt, dtype=torch.int64, device='cuda'); ow = torch.empty(n*cnt, dtype=torch.float32, device='cuda')
    for grid in (4096,):
        acc = (C.c_uint64 * 16)()
        for rep in range(2):
            ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
            ev0.record()
            rc = L.euler_gpu_debug_k1_phases(G._h, st, C.c_uint64(20240521), C.c_void_p(r.data_ptr()), C.c_int64(n), C.c_int32(cnt), C.c_void_p(oid.data_ptr()), C.c_void_p(ow.data_ptr()), C.c_int32(grid), acc)
            ev1.record(); torch.cuda.synchronize()
        assert rc == 0, _lib.last_error() if hasattr(_lib, 'last_error') else rc
        iters = acc[8] // 1
        waves = grid * 4
        res = {names[i]: round(acc[i] / max(acc[8], 1), 1) for i in range(7)}
        print(label, 'grid', grid, 'ms', round(ev0.elapsed_time(ev1), 3), 'ticks per wave-iteration:', json.dumps(res), 'probe steps per wave-iter', round(acc[9] / acc[8], 2), 'wave-iters', acc[8])
