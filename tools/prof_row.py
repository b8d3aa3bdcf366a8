"""Row-kernel decomposition on the metric's first hop: per-kernel times under
rocprofv3 (run: rocprofv3 --kernel-trace --stats -- python tools/prof_row.py) and
ablations (tuning key 2: 1 = no draws, 2 = no write phase, 4 = no Philox)."""
import sys, json, ctypes as C
sys.path.insert(0, '.')
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


def hop1(iters=20):
    for i in range(3):
        G.sample_neighbor(roots, [0], 25, N + 1, call_id=i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        G.sample_neighbor(roots, [0], 25, N + 1, call_id=i)
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / iters * 1e3, 1)


res = {}
for row in (1, 0):
    L.euler_gpu_set_tuning(19, row)
    for ab in ((0, 1, 2, 3, 4, 7) if row else (0,)):
        L.euler_gpu_set_tuning(2, ab)
        res["row=%d ablate=%d" % (row, ab)] = hop1()
L.euler_gpu_set_tuning(2, 0); L.euler_gpu_set_tuning(19, 1)
# count 10 on the same roots (even count: pair mode of the lane-per-sample kernel)
print(json.dumps(res))
