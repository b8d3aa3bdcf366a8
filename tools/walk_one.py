"""A few DeepWalk calls (1M walkers x 40 steps, metric graph) - the command a rocprofv3
kernel trace wraps.  python tools/walk_one.py [--tuning 38=0] [--calls 3]"""
import argparse, sys
sys.path.insert(0, '.')
import torch, euler_amd
from euler_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument('--tuning', default='')
ap.add_argument('--calls', type=int, default=3)
a = ap.parse_args()
L = _lib.lib()
for kv in filter(None, a.tuning.split(',')):
    k, v = kv.split('=')
    _lib.check(L.euler_gpu_set_tuning(int(k), int(v)))
N = 100_000_000
G = euler_amd.Graph.synthetic(euler_amd.synth_params(20240521, N, 1_000_000_000, weighted=True))
G.set_seed(20240521)
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
starts = torch.randint(1, N + 1, (a.calls, 1_000_000), generator=gen, device='cuda')
import time
G.random_walk(starts[0], [[0]] * 40, 1.0, 1.0, N + 1, call_id=0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.calls):
    G.random_walk(starts[i], [[0]] * 40, 1.0, 1.0, N + 1, call_id=40 * i)
torch.cuda.synchronize()
print('tuning [%s]: %.4f ms per walk of 1M x 40' % (a.tuning, (time.perf_counter() - t0) / a.calls * 1e3))
