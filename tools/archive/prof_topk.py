import sys
sys.path.insert(0, '.')
import torch, euler_amd
N = 100_000_000
p = euler_amd.synth_params(20240521, N, 1_000_000_000, weighted=True)
G = euler_amd.Graph.synthetic(p)
gen = torch.Generator(device='cuda'); gen.manual_seed(5)
roots = torch.randint(1, N + 1, (131072,), generator=gen, device='cuda')
for i in range(3):
    G.get_top_k_neighbor(roots, [0], 5, -1)
torch.cuda.synchronize()
