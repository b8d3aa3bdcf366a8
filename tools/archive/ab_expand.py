"""DedupExpandKernel settings on the metric's fanout, timed in place
(euler_gpu_time_sample_fanout_phases): steps in flight per lane (key 10),
type column rebuilt from the mask (key 11), workgroup cap (key 12).

  python tools/ab_expand.py"""
import sys, json, ctypes as C
sys.path.insert(0, '.')
import torch, euler_amd
from euler_amd import _lib
L = _lib.lib()
fan = [25, 10]
layers = 2
N = 100_000_000
p = euler_amd.synth_params(20240521, N, 1_000_000_000, weighted=True)
G = euler_amd.Graph.synthetic(p)
G.set_seed(20240521)
B = 131072
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


def run():
    ms = (C.c_float * (3 * layers))()
    nu = (C.c_int64 * layers)()
    _lib.check(L.euler_gpu_time_sample_fanout_phases(
        G._h, st, 20240521, C.c_void_p(roots.data_ptr()), B, et, 1, cnt, layers, N + 1,
        pn, pw, pt, C.c_void_p(ws.data_ptr()), 20, ms, nu))
    return [round(x, 4) for x in ms]


def sig():
    out = G.sample_fanout(roots, [[0]] * layers, fan, N + 1, call_id=0)
    return ([int(x.sum().item()) for x in out[0]] + [float(x.double().sum().item()) for x in out[1]]
            + [int(x.sum().item()) for x in out[2]])


run()
res = {}
ref = None
for steps in (1, 2, 4):
    for ct in (0, 1):
        for cap in (0, 8192, 4096, 2048):
            L.euler_gpu_set_tuning(10, steps)
            L.euler_gpu_set_tuning(11, ct)
            L.euler_gpu_set_tuning(12, cap)
            ms = run()
            s = sig()
            if ref is None:
                ref = s
            assert s == ref, (steps, ct, cap)
            res['steps=%d const_type=%d cap=%d' % (steps, ct, cap)] = {
                'expand_ms': ms[5], 'k1u_ms': ms[4], 'dedup_ms': ms[3], 'hop1_ms': ms[1],
                'sum_ms': round(sum(ms), 4)}
print(json.dumps(res, indent=1))
