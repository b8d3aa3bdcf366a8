import sys, json, time
sys.path.insert(0, '.')
import torch, euler_amd
from euler_amd import ops
p = euler_amd.synth_params(20240521, 100_000_000, 1_000_000_000, weighted=True)
G = euler_amd.Graph.synthetic(p)
G.set_seed(20240521)
B = 131072
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
roots = torch.randint(1, 100_000_001, (B,), generator=gen, device='cuda')
out = G.sample_fanout(roots, [[0],[0]], [25,10], 100_000_001, call_id=0)
hop2 = out[0][1].contiguous()
u = torch.unique(hop2)
print('hop2 roots', hop2.numel(), 'unique', u.numel(), 'frac', u.numel() / hop2.numel())
cnt = torch.unique(hop2, return_counts=True)[1].sort(descending=True)[0]
print('top counts', cnt[:10].tolist(), 'singletons', int((cnt == 1).sum()))
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = ops.id_unique(hop2)
    torch.cuda.synchronize(); print('id_unique ms', (time.perf_counter() - t0) * 1e3)
