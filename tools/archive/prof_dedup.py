import sys, ctypes as C
sys.path.insert(0, '.')
import torch, euler_amd
from euler_amd import _lib
L = _lib.lib()
p = euler_amd.synth_params(20240521, 100_000_000, 1_000_000_000, weighted=True)
G = euler_amd.Graph.synthetic(p)
G.set_seed(20240521)
B = 131072
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
roots = torch.randint(1, 100_000_001, (B,), generator=gen, device='cuda')
for i in range(4):
    out = G.sample_fanout(roots, [[0], [0]], [25, 10], 100_000_001, call_id=2 * i)
torch.cuda.synchronize()
