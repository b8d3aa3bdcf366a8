"""A few steps of the metric fanout (131 072 roots x [25, 10], 100M / 1B graph) with a
given tuning - the command the rocprofv3 passes of tools/pmc_fl.sh wrap.
  python tools/fl_one.py [--tuning 28=4,29=32] [--steps 6]"""
import argparse, sys
sys.path.insert(0, '.')
import torch, euler_amd
from euler_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument('--tuning', default='')
ap.add_argument('--steps', type=int, default=6)
ap.add_argument('--nodes', type=int, default=100_000_000)
ap.add_argument('--edges', type=int, default=1_000_000_000)
ap.add_argument('--batch', type=int, default=131072)
a = ap.parse_args()
L = _lib.lib()
for kv in filter(None, a.tuning.split(',')):
    k, v = kv.split('=')
    _lib.check(L.euler_gpu_set_tuning(int(k), int(v)))
p = euler_amd.synth_params(20240521, a.nodes, a.edges, weighted=True)
G = euler_amd.Graph.synthetic(p)
G.set_seed(20240521)
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
roots = torch.randint(1, a.nodes + 1, (a.steps, a.batch), generator=gen, device='cuda')
for i in range(a.steps):
    G.sample_fanout(roots[i], [[0], [0]], [25, 10], a.nodes + 1, call_id=2 * i)
torch.cuda.synchronize()
