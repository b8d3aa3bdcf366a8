"""A/B of the one-kernel step's search on the metric graph: weight-bucket index (key 45 = 1) vs
pivot levels (45 = 0), at the register budgets of key 35.  Kernel alone (HIP events around 20
launches, euler_gpu_time_sample_fanout) and the two-stream loop; outputs compared bit for bit.
  python tools/ab_wb.py [--configs 45=0,35=5 45=1,35=5 ...]"""
import argparse, ctypes as C, json, sys, time
sys.path.insert(0, '.')
import torch, euler_amd
from euler_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument('--configs', nargs='*', default=['45=0,35=5', '45=1,35=5', '45=1,35=6', '45=1,35=8'])
ap.add_argument('--nodes', type=int, default=100_000_000)
ap.add_argument('--edges', type=int, default=1_000_000_000)
ap.add_argument('--batch', type=int, default=131072)
a = ap.parse_args()
L = _lib.lib()
N, B = a.nodes, a.batch
G = euler_amd.Graph.synthetic(euler_amd.synth_params(20240521, N, a.edges, weighted=True))
G.set_seed(20240521)
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
roots = torch.randint(1, N + 1, (24, B), generator=gen, device='cuda')
FAN = [25, 10]
cnt = (C.c_int32 * 2)(*FAN); et = (C.c_int32 * 2)(0, 0)
o_n, o_w, o_t, m = [], [], [], B
for c in FAN:
    m *= c
    o_n.append(torch.empty(m, dtype=torch.int64, device='cuda'))
    o_w.append(torch.empty(m, dtype=torch.float32, device='cuda'))
    o_t.append(torch.empty(m, dtype=torch.int32, device='cuda'))
ws = torch.empty(max(int(L.euler_gpu_sample_fanout_workspace(B, cnt, 2)), 16), dtype=torch.uint8, device='cuda')
pn = (C.c_void_p * 2)(*[t.data_ptr() for t in o_n]); pw = (C.c_void_p * 2)(*[t.data_ptr() for t in o_w])
pt = (C.c_void_p * 2)(*[t.data_ptr() for t in o_t])
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
side = [torch.cuda.Stream(), torch.cuda.Stream()]
ref = None
out = {}
t0 = time.time()
for cfg in a.configs:
    for kv in cfg.split(','):
        k, v = kv.split('=')
        _lib.check(L.euler_gpu_set_tuning(int(k), int(v)))
    ms = C.c_float(0)
    best = []
    for rep in range(3):
        _lib.check(L.euler_gpu_time_sample_fanout(G._h, st, 20240521, C.c_void_p(roots[0].data_ptr()), B, et, 1, cnt, 2,
                                                  N + 1, pn, pw, pt, C.c_void_p(ws.data_ptr()), 20, C.byref(ms)))
        best.append(round(ms.value, 4))
    got = (o_n[0].clone(), o_n[1].clone(), o_w[1].clone())
    if ref is None:
        ref = got
    same = all(torch.equal(x, y) for x, y in zip(ref, got))
    two = []
    for rep in range(3):
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for i in range(24):
            with torch.cuda.stream(side[i % 2]):
                G.sample_fanout(roots[i], [[0], [0]], FAN, N + 1, call_id=2 * i)
        torch.cuda.synchronize(); two.append(round((time.perf_counter() - t1) / 24 * 1e3, 4))
    out[cfg] = {'alone_ms': best, 'two_stream_ms_per_step': two, 'same_as_first': same}
    print(cfg, out[cfg], flush=True)
print(json.dumps({'graph_bytes': G.device_bytes, 'results': out}))
