"""Sweep of the one-kernel fanout (fanout_local.h) on the metric workload: the step
(131 072 roots x [25, 10] on the 100M / 1B graph) timed with HIP events on the stream
the kernels run on, for a list of geometries, against the hop-by-hop path (key 27 = 0).

  python tools/ab_fanout_local.py [--quick] [--configs "28=4,29=32;28=2,29=16"]

Every configuration's outputs are checksummed against the hop-by-hop path's."""
import argparse, json, sys, time, ctypes as C
sys.path.insert(0, '.')
import torch, euler_amd
from euler_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument('--quick', action='store_true')
ap.add_argument('--configs', default='')
ap.add_argument('--nodes', type=int, default=100_000_000)
ap.add_argument('--edges', type=int, default=1_000_000_000)
ap.add_argument('--batch', type=int, default=131072)
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--out', default='gpurun_out/ab_fanout_local.json')
args = ap.parse_args()

L = _lib.lib()
N = args.nodes
p = euler_amd.synth_params(20240521, N, args.edges, weighted=True)
G = euler_amd.Graph.synthetic(p)
G.set_seed(20240521)
B = args.batch
fan = [25, 10]
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
n_sets = 8
roots = torch.randint(1, N + 1, (n_sets, B), generator=gen, device='cuda')
DEFAULTS = {27: 1, 28: 4, 29: 0, 30: 64, 31: 1, 32: -1, 33: 32768, 34: 2, 35: 5, 36: 0}


def apply(cfg):
    for k, v in DEFAULTS.items():
        _lib.check(L.euler_gpu_set_tuning(k, v))
    for k, v in cfg.items():
        _lib.check(L.euler_gpu_set_tuning(k, v))


def sig(out):
    return ([int(x.sum().item()) for x in out[0]] + [float(x.double().sum().item()) for x in out[1]]
            + [int(x.sum().item()) for x in out[2]])


def timed(iters):
    for i in range(3):
        G.sample_fanout(roots[i % n_sets], [[0], [0]], fan, N + 1, call_id=2 * i)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        G.sample_fanout(roots[i % n_sets], [[0], [0]], fan, N + 1, call_id=2 * i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def two_streams(iters):
    ss = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for i in range(4):
        with torch.cuda.stream(ss[i % 2]):
            G.sample_fanout(roots[i % n_sets], [[0], [0]], fan, N + 1, call_id=2 * i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        with torch.cuda.stream(ss[i % 2]):
            G.sample_fanout(roots[i % n_sets], [[0], [0]], fan, N + 1, call_id=2 * i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


if args.configs:
    configs = [dict((int(kv.split('=')[0]), int(kv.split('=')[1])) for kv in c.split(',') if kv)
               for c in args.configs.split(';')]
else:
    configs = [{}, {34: 1}]
    for gr in (2, 4, 8, 9, 12, 16):
        for capm in (4, 8):
            configs.append({28: gr, 29: gr * capm})
    configs += [{28: 4, 29: 32, 30: 64}, {28: 4, 29: 32, 30: 128}, {28: 4, 29: 32, 31: 0},
                {28: 4, 29: 32, 32: 2048}, {28: 4, 29: 32, 32: 4096}, {28: 8, 29: 64, 30: 64},
                {28: 4, 29: 16, 30: 64}, {28: 4, 29: 24, 30: 64}, {28: 9, 29: 72, 30: 64}]
    if args.quick:
        configs = configs[:6]

apply({27: 0})
ref = sig(G.sample_fanout(roots[0], [[0], [0]], fan, N + 1, call_id=0))
rows = [{'config': 'hop by hop (27=0)', 'ms': round(timed(args.iters), 4),
         'ms_two_streams': round(two_streams(args.iters), 4)}]
print(json.dumps(rows[-1]), flush=True)
for cfg in configs:
    apply(cfg)
    s = sig(G.sample_fanout(roots[0], [[0], [0]], fan, N + 1, call_id=0))
    ok = s == ref or cfg.get(36, 0) != 0
    row = {'config': cfg, 'ms': round(timed(args.iters), 4),
           'ms_two_streams': round(two_streams(args.iters), 4), 'matches_hop_by_hop': ok}
    rows.append(row)
    print(json.dumps(row), flush=True)
apply({})
json.dump(rows, open(args.out, 'w'), indent=1)
