import sys
sys.path.insert(0, '/root/repo')
import torch, euler_amd
from euler_amd import _lib
N=100_000_000
G = euler_amd.Graph.synthetic(euler_amd.synth_params(20240521, N, 10*N, weighted=True))
G.set_seed(20240521)
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
starts = torch.randint(1, N + 1, (100_000,), generator=gen, device='cuda', dtype=torch.int64)
big = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
_lib.lib().euler_gpu_set_tuning(25, big)
for _ in range(2):
    G.random_walk(starts, [[0]]*10, 0.25, 4.0, N+1, call_id=3)
torch.cuda.synchronize()
