"""The headline's two-stream loop, alone, for `rocprofv3 --kernel-trace`: consecutive minibatches
of the metric step alternate between two HIP streams exactly as bench.py's timed loop does
(bench.py: loop(first, last, side)).  The trace shows whether two SampleFanoutLeanKernel
dispatches really overlap - tools/trace_overlap.py turns it into profiles/r4_two_stream_trace.csv.
  python tools/two_stream_trace.py [--steps 24] [--streams 2]"""
import argparse, sys, time
sys.path.insert(0, '.')
import torch, euler_amd
ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=24)
ap.add_argument('--streams', type=int, default=3)
ap.add_argument('--nodes', type=int, default=100_000_000)
ap.add_argument('--edges', type=int, default=1_000_000_000)
ap.add_argument('--batch', type=int, default=131072)
a = ap.parse_args()
p = euler_amd.synth_params(20240521, a.nodes, a.edges, weighted=True)
G = euler_amd.Graph.synthetic(p)
G.set_seed(20240521)
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
roots = torch.randint(1, a.nodes + 1, (a.steps, a.batch), generator=gen, device='cuda')
side = [torch.cuda.Stream() for _ in range(a.streams)]
torch.cuda.synchronize()
for rep in range(2):                       # the second pass is the one to read (allocator warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        with torch.cuda.stream(side[i % a.streams]):
            G.sample_fanout(roots[i], [[0], [0]], [25, 10], a.nodes + 1, call_id=2 * i)
    torch.cuda.synchronize()
    print('pass %d: %.4f ms / step' % (rep, (time.perf_counter() - t0) / a.steps * 1e3))
