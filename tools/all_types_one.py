"""The metric step listing ALL edge types per hop (what the reference's evaluation does:
metapath = [all_edge_type] * layers, examples/graphsage/run_graphsage.py:57) on the hashed-id /
2-type graph of bench.py's metric_hashed_T2 leg: ms per step on alternating streams and alone."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, euler_amd
import bench
if '--lib' in sys.argv:                 # A/B against another build of the library
    from euler_amd import _lib
    _lib.LIB_PATH = sys.argv[sys.argv.index('--lib') + 1]
N = 100_000_000
WEIGHTED = '--unweighted' not in sys.argv
if '--tuning' in sys.argv:
    from euler_amd import _lib as _l
    for kv in sys.argv[sys.argv.index('--tuning') + 1].split(','):
        k_, v_ = kv.split('=')
        _l.check(_l.lib().euler_gpu_set_tuning(int(k_), int(v_)))
p = euler_amd.synth_params(bench.GRAPH_SEED, N, 10 * N, n_types=2, weighted=WEIGHTED, hashed_ids=True)
G = euler_amd.Graph.synthetic(p)
G.set_seed(bench.GRAPH_SEED)
gen = torch.Generator(device="cuda"); gen.manual_seed(2468)
B = 131072
roots = bench._mix64_t(torch.randint(1, N + 1, (16, B), generator=gen, device="cuda", dtype=torch.int64))
side = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()]
ONE = '--one' in sys.argv       # one stream, the all-types case only (for a kernel trace)
for et in ([[0, 1], [0, 1]],) if ONE else ([[0], [0]], [[0, 1], [0, 1]], [[1, 0], [1, 0]]):
    def loop(a, b):
        for i in range(a, b):
            with torch.cuda.stream(side[i % 3]):
                G.sample_fanout(roots[i % 16], et, [25, 10], -1, call_id=2 * i)
    ms = float('nan')
    if not ONE:
        loop(0, 6); torch.cuda.synchronize()
        t0 = time.perf_counter(); loop(6, 46); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 40 * 1e3
    t0 = time.perf_counter()
    for i in range(20):
        G.sample_fanout(roots[i % 16], et, [25, 10], -1, call_id=2 * i)
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / 20 * 1e3
    print(et, "three streams %.4f ms/step = %.1f G edges/s; one stream %.4f ms" % (ms, B * 275 / ms / 1e6, one))
