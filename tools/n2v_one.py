import sys
sys.path.insert(0, '/root/repo')
import torch, euler_amd
from euler_amd import _lib
N=100_000_000
G = euler_amd.Graph.synthetic(euler_amd.synth_params(20240521, N, 10*N, weighted=True))
G.set_seed(20240521)
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
starts = torch.randint(1, N + 1, (100_000,), generator=gen, device='cuda', dtype=torch.int64)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
big = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
W = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
_lib.lib().euler_gpu_set_tuning(7, mode)
_lib.lib().euler_gpu_set_tuning(25, big)
import os
for kv in filter(None, os.environ.get('N2V_KEYS', '').split(',')):      # e.g. N2V_KEYS=73=0
    k, v = kv.split('='); _lib.check(_lib.lib().euler_gpu_set_tuning(int(k), int(v)))
s = starts[:W].contiguous()
import ctypes as C
st = (C.c_uint64 * 8)()
G.random_walk(s, [[0]]*10, 0.25, 4.0, N+1, call_id=3)
_lib.lib().euler_gpu_random_walk_stats(None, 2)
w = G.random_walk(s, [[0]]*10, 0.25, 4.0, N+1, call_id=3)
_lib.lib().euler_gpu_random_walk_stats(st, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(3):
    G.random_walk(s, [[0]]*10, 0.25, 4.0, N+1, call_id=3)
e1.record(); torch.cuda.synchronize()
print("mode", mode, "big", big, "walkers", W, "ms", round(e0.elapsed_time(e1) / 3, 3), "stats", list(st))
torch.cuda.synchronize()
