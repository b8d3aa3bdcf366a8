import sys
sys.path.insert(0, '/root/repo')
import torch, euler_amd
from euler_amd import _lib
N = 100_000_000
G = euler_amd.Graph.synthetic(euler_amd.synth_params(20240521, N, 10 * N, weighted=True))
G.set_seed(20240521)
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
roots = torch.randint(1, N + 1, (131072,), generator=gen, device='cuda', dtype=torch.int64)
_lib.lib().euler_gpu_set_tuning(19, int(sys.argv[1]) if len(sys.argv) > 1 else 2)
for i in range(4):
    G.sample_fanout(roots, [[0], [0]], [25, 10], N + 1, call_id=2 * i)
torch.cuda.synchronize()
