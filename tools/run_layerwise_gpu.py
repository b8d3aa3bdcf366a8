"""One-process GPU check of the layerwise / SparseGetAdj / typed-SampleNode ops:
(1) their parity tests + the golden subset of the existing suite, (2) timings
on the metric-sized graph -> gpurun_out/r1_layerwise_{pytest.txt,bench.json}.
One process so that `import torch` is paid once (GPU minutes are scarce)."""
import io, json, os, sys, contextlib
sys.path.insert(0, '.')
os.makedirs('gpurun_out', exist_ok=True)
import pytest

BENCH_ONLY = '--bench-only' in sys.argv
if BENCH_ONLY:
    sys.argv.remove('--bench-only')
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    rc = 0 if BENCH_ONLY else pytest.main(['-q', '-m', 'gpu', '-p', 'no:cacheprovider',
                      'tests/test_layerwise_gpu.py',
                      'tests/test_gpu_parity.py::test_fixture_goldens_gpu',
                      'tests/test_gpu_parity.py::test_random_graph_goldens_gpu',
                      'tests/test_gpu_parity.py::test_op_registry_and_dat_loader',
                      'tests/test_gpu_parity.py::test_sample_node_and_walks_vs_oracle'])
out = buf.getvalue()
if not BENCH_ONLY:
    open('gpurun_out/r1_layerwise_pytest.txt', 'w').write(out)
print(out[-3000:])
print('pytest rc', int(rc))

import torch, euler_amd

def timed(fn, iters=5, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, out

res = {}
N, E = 100_000_000, 1_000_000_000
if len(sys.argv) > 1:                      # smaller graph for a dry run
    N, E = int(sys.argv[1]), int(sys.argv[1]) * 10
G = euler_amd.Graph.synthetic(euler_amd.synth_params(20240521, N, E, weighted=True))
G.set_seed(1)
gen = torch.Generator(device='cuda'); gen.manual_seed(5)
ids = torch.randint(1, N + 1, (1 << 20,), generator=gen, device='cuda')
ms, w = timed(lambda: G.get_edge_sum_weight(ids, [0]))
res['get_edge_sum_weight_1M'] = {'ms': round(ms, 4)}
ms, _ = timed(lambda: G.sample_layer(ids, [0], -1, call_id=3))
res['sample_layer_1M'] = {'ms': round(ms, 4), 'samples_per_s': ids.numel() / ms * 1e3}
from euler_amd import _lib
for mode, host_rows in (('device', 0), ('host', 2), ('auto', 1)):      # tuning key 15 A/B
    _lib.lib().euler_gpu_set_tuning(15, host_rows)
    for batch, n, m in ((1024, 25, 10), (1024, 256, 256), (256, 256, 256), (64, 1000, 100),
                        (8, 10000, 1000), (1, 100000, 1000)):
        r = ids[:batch * n].reshape(batch, n)
        wr = w[:batch * n].reshape(batch, n)
        slow = mode == 'device' and n >= 10000
        ms, _ = timed(lambda: G.sample_root(r, wr, m, -1, call_id=4),
                      iters=1 if slow else 5, warm=0 if slow else 1)
        res['sample_root_%s_b%d_n%d_m%d' % (mode, batch, n, m)] = {'ms': round(ms, 4)}
_lib.lib().euler_gpu_set_tuning(15, 1)
for batch, n, m in ((1024, 25, 10), (128, 250, 100), (1, 10000, 1000)):
    r = ids[:batch * n].reshape(batch, n)
    ms, out = timed(lambda: G.sample_neighbor_layerwise(r, [0], m, -1, call_id=5), iters=3)
    res['sample_neighbor_layerwise_b%d_n%d_m%d' % (batch, n, m)] = {
        'ms': round(ms, 4), 'nnz': int(out[1][0].shape[0]),
        'note': 'sum weight + root draw + layer draw + adjacency (2 passes, 1 host sync)'}
    nb = out[0]
    ms, _ = timed(lambda: G.sparse_get_adj(r, nb, [0], n, m), iters=3)
    res['sparse_get_adj_b%d_n%d_m%d' % (batch, n, m)] = {
        'ms': round(ms, 4), 'pairs_per_s': batch * n * m / ms * 1e3}
sub = ids[:4096]
for mode, key in (('hash', 0), ('scan', 1)):
    _lib.lib().euler_gpu_set_tuning(16, key)
    ms, out = timed(lambda: G.sparse_get_adj(sub, sub, [0], -1, -1), iters=3)
    res['sparse_get_adj_whole_4096x4096_' + mode] = {
        'ms': round(ms, 4), 'nnz': int(out[0].shape[0]), 'pairs_per_s': 4096 * 4096 / ms * 1e3}
    r = ids[:1024 * 25].reshape(1024, 25)
    nb = ids[30000:30000 + 1024 * 10].reshape(1024, 10)
    ms, out = timed(lambda: G.sparse_get_adj(r, nb, [0], 25, 10), iters=3)
    res['sparse_get_adj_b1024_n25_m10_' + mode] = {'ms': round(ms, 4)}
_lib.lib().euler_gpu_set_tuning(16, 0)
hub = torch.arange(1, 4097, device='cuda')           # the heaviest rows of the graph
ms, out = timed(lambda: G.sparse_get_adj(hub, hub, [0], -1, -1), iters=2)
res['sparse_get_adj_hubs_4096x4096_hash'] = {'ms': round(ms, 4), 'nnz': int(out[0].shape[0])}
for mode, key in (('scalar', 1), ('dpp', 0)):
    _lib.lib().euler_gpu_set_tuning(18, key)
    ms, _ = timed(lambda: G.get_edge_sum_weight(hub, [0]), iters=2)
    res['get_edge_sum_weight_hubs_4096_' + mode] = {'ms': round(ms, 4)}
    ms, _ = timed(lambda: G.get_edge_sum_weight(ids, [0]))
    res['get_edge_sum_weight_1M_' + mode] = {'ms': round(ms, 4)}
_lib.lib().euler_gpu_set_tuning(18, 0)
ms, t = timed(lambda: G.get_node_type(ids))
res['get_node_type_1M'] = {'ms': round(ms, 4)}
ms, out = timed(lambda: G.sample_neighbor_layerwise(ids[:25600].reshape(1024, 25), [0], 10, -1,
                                                    'sqrt', call_id=6), iters=3)
res['sample_neighbor_layerwise_sqrt_b1024_n25_m10'] = {
    'ms': round(ms, 4), 'note': 'full neighbours -> host tables (unordered_map order) -> draws'}
print(json.dumps(res, indent=1))
json.dump(res, open('gpurun_out/r1_layerwise_bench.json', 'w'), indent=1)
