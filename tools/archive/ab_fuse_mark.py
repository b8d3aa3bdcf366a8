"""Fanout phases with and without the hop chaining (tuning key 9): hop h's
kernels enter their ids into hop h+1's owner table.

  python tools/ab_fuse_mark.py [fanout ...]     default 25 10"""
import sys, json, time, ctypes as C
sys.path.insert(0, '.')
import torch, euler_amd
from euler_amd import _lib
L = _lib.lib()
fan = [int(x) for x in sys.argv[1:]] or [25, 10]
layers = len(fan)
N = 100_000_000
p = euler_amd.synth_params(20240521, N, 1_000_000_000, weighted=True)
G = euler_amd.Graph.synthetic(p)
G.set_seed(20240521)
B = 131072 if layers == 2 else 8192
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
roots = torch.randint(1, N + 1, (B,), generator=gen, device='cuda')
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
et = (C.c_int32 * layers)(*([0] * layers))
cnt = (C.c_int32 * layers)(*fan)
outs_n, outs_w, outs_t = [], [], []
m = B
for c in fan:
    m *= c
    outs_n.append(torch.empty(m, dtype=torch.int64, device='cuda'))
    outs_w.append(torch.empty(m, dtype=torch.float32, device='cuda'))
    outs_t.append(torch.empty(m, dtype=torch.int32, device='cuda'))
ws = torch.empty(max(int(L.euler_gpu_sample_fanout_workspace(B, cnt, layers)), 16),
                 dtype=torch.uint8, device='cuda')
pn = (C.c_void_p * layers)(*[t.data_ptr() for t in outs_n])
pw = (C.c_void_p * layers)(*[t.data_ptr() for t in outs_w])
pt = (C.c_void_p * layers)(*[t.data_ptr() for t in outs_t])
res = {}
ref = None
for fuse in (1, 0, 1, 0):
    L.euler_gpu_set_tuning(9, fuse)
    ms = (C.c_float * (3 * layers))()
    nu = (C.c_int64 * layers)()
    _lib.check(L.euler_gpu_time_sample_fanout_phases(
        G._h, st, 20240521, C.c_void_p(roots.data_ptr()), B, et, 1, cnt, layers, N + 1,
        pn, pw, pt, C.c_void_p(ws.data_ptr()), 20, ms, nu))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        G.sample_fanout(roots, [[0]] * layers, fan, N + 1, call_id=0)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 20 * 1e3
    out = G.sample_fanout(roots, [[0]] * layers, fan, N + 1, call_id=0)
    sig = [int(x.sum().item()) for x in out[0]] + [float(x.double().sum().item()) for x in out[1]]
    if ref is None:
        ref = sig
    assert sig == ref, (sig, ref)
    res.setdefault('fuse=%d' % fuse, []).append(
        {'phases_ms': [round(x, 4) for x in ms], 'sum_ms': round(sum(ms), 4),
         'wall_ms_per_fanout': round(wall, 4), 'unique_last': nu[layers - 1]})
print(json.dumps(res, indent=1))
