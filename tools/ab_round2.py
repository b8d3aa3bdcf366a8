"""Round-2 A/B on the metric workload (100M / 1B, 131 072 roots, fanout [25, 10]):
the fanout timed in place, phase by phase, under

  key 19  hop-1 kernel: one lane per ROOT (k1_row.h) vs one lane per sample
  key 14  numbering of the distinct roots: 2 = one pass, 1 = count / scan / assign
  key 20  last hop: the expansion resolves its row through the owner table

  python tools/ab_round2.py [nodes] [edges]
"""
import ctypes as C
import json
import sys

sys.path.insert(0, '.')
import torch
import euler_amd
from euler_amd import _lib

L = _lib.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
E = int(sys.argv[2]) if len(sys.argv) > 2 else 10 * N
p = euler_amd.synth_params(20240521, N, E, weighted=True)
G = euler_amd.Graph.synthetic(p)
G.set_seed(20240521)
B = 131072
FAN = [25, 10]
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
roots = torch.randint(1, N + 1, (B,), generator=gen, device='cuda')
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
layers = len(FAN)
cnt_a = (C.c_int32 * layers)(*FAN)
et_a = (C.c_int32 * layers)(*([0] * layers))
o_n, o_w, o_t, m = [], [], [], B
for c in FAN:
    m *= c
    o_n.append(torch.empty(m, dtype=torch.int64, device='cuda'))
    o_w.append(torch.empty(m, dtype=torch.float32, device='cuda'))
    o_t.append(torch.empty(m, dtype=torch.int32, device='cuda'))
wsz = int(L.euler_gpu_sample_fanout_workspace(B, cnt_a, layers))
fws = torch.empty(max(wsz, 16), dtype=torch.uint8, device='cuda')
pn = (C.c_void_p * layers)(*[t.data_ptr() for t in o_n])
pw = (C.c_void_p * layers)(*[t.data_ptr() for t in o_w])
pt = (C.c_void_p * layers)(*[t.data_ptr() for t in o_t])


def phases(iters=10):
    ms = (C.c_float * (3 * layers))()
    nu = (C.c_int64 * layers)()
    _lib.check(L.euler_gpu_time_sample_fanout_phases(
        G._h, st, 20240521, C.c_void_p(roots.data_ptr()), B, et_a, 1, cnt_a, layers, N + 1,
        pn, pw, pt, C.c_void_p(fws.data_ptr()), iters, ms, nu))
    v = [round(x, 4) for x in ms]
    return {"hop1_k1": v[1], "hop2_dedup": v[3], "hop2_k1": v[4], "hop2_expand": v[5],
            "sum": round(sum(v), 4), "unique": nu[layers - 1]}


def whole(steps=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        G.sample_fanout(roots, [[0], [0]], FAN, N + 1, call_id=2 * i)
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / steps, 4)


ref = None
res = {}
# (19 row kernel, 14 numbering, 21 lean expand, 22 pair mode over the distinct roots)
for row, num, lean, pair2 in ((0, 2, 0, 0), (0, 2, 1, 0), (0, 2, 1, 1), (0, 2, 0, 1), (1, 2, 1, 0), (0, 1, 1, 0)):
    L.euler_gpu_set_tuning(19, row)
    L.euler_gpu_set_tuning(14, num)
    L.euler_gpu_set_tuning(20, 0)
    L.euler_gpu_set_tuning(21, lean)
    L.euler_gpu_set_tuning(22, pair2)
    out = G.sample_fanout(roots, [[0], [0]], FAN, N + 1, call_id=0)
    sig = [int(x.sum().item()) for x in out[0][1:]] + [float(x.double().sum().item()) for x in out[1]] \
        + [int(x.sum().item()) for x in out[2]]
    if ref is None:
        ref = sig
    assert sig == ref, ("results differ between tunings", sig, ref)
    phases(2)
    key = "row=%d numbering=%d lean_expand=%d pair_distinct=%d" % (row, num, lean, pair2)
    res[key] = {"phases_ms": phases(), "ms_per_step": whole()}
    print(key, json.dumps(res[key]), flush=True)
L.euler_gpu_set_tuning(21, 1); L.euler_gpu_set_tuning(22, 0)
L.euler_gpu_set_tuning(19, 1); L.euler_gpu_set_tuning(14, 2); L.euler_gpu_set_tuning(20, 0)
# B = 1024 latency (the batch of the reference's examples)
small = roots[:1024].contiguous()
for row in (2, 0):
    L.euler_gpu_set_tuning(19, row)
    for _ in range(20):
        G.sample_fanout(small, [[0], [0]], FAN, N + 1, call_id=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(200):
        G.sample_fanout(small, [[0], [0]], FAN, N + 1, call_id=i)
    e1.record(); torch.cuda.synchronize()
    res["B1024 row=%d us_per_step" % row] = round(e0.elapsed_time(e1) / 200 * 1e3, 2)
    print("B1024 row=%d" % row, res["B1024 row=%d us_per_step" % row], flush=True)
L.euler_gpu_set_tuning(19, 1)
print(json.dumps(res))
