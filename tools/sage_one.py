"""SageDataFlow block construction (euler_gpu_sage_blocks) on the metric graph, a few
minibatches - the command `rocprofv3 --kernel-trace --stats` wraps for
profiles/r4_sage_blocks_kernel_stats.csv.   python tools/sage_one.py [--roots 16384] [--steps 10]"""
import argparse, sys, time
sys.path.insert(0, '.')
import torch, euler_amd
from euler_amd.dataflow import SageDataFlow
ap = argparse.ArgumentParser()
ap.add_argument('--roots', type=int, default=16384)
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--nodes', type=int, default=100_000_000)
ap.add_argument('--edges', type=int, default=1_000_000_000)
a = ap.parse_args()
N = a.nodes
G = euler_amd.Graph.synthetic(euler_amd.synth_params(20240521, N, a.edges, weighted=True))
G.set_seed(20240521)
gen = torch.Generator(device='cuda'); gen.manual_seed(77)
r = torch.randint(1, N + 1, (a.roots,), generator=gen, device='cuda', dtype=torch.int64)
flow = SageDataFlow(G, [25, 10], [[0], [0]], add_self_loops=True, max_id=N)
for i in range(3):
    df = flow(r)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps):
    df = flow(r)
torch.cuda.synchronize()
print('%.4f ms per minibatch; layers %s' % ((time.perf_counter() - t0) / a.steps * 1e3, [b.size for b in df]))
