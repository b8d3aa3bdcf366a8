"""Secondary kernels of the hot path on the metric-sized graph: time (HIP
events via torch on the current stream), algorithmic bytes and
GB/s.  Writes one JSON object; the committed copy lives in profiles/."""
import json, sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, euler_amd
from euler_amd import ops

def timed(fn, iters=10, warm=2):
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
p = euler_amd.synth_params(20240521, N, E, weighted=True)
G = euler_amd.Graph.synthetic(p)
G.set_seed(1)
gen = torch.Generator(device='cuda'); gen.manual_seed(5)

# K4 random walk (config 4 shape on one GPU): 1M walkers, 40 steps, p = q = 1
W, L = 1_000_000, 40
starts = torch.randint(1, N + 1, (W,), generator=gen, device='cuda')
ms, walks = timed(lambda: G.random_walk(starts, [[0]] * L, 1.0, 1.0, N + 1, call_id=7), iters=3, warm=1)
res['random_walk_len40_1M'] = {'ms': round(ms, 3), 'steps_per_s': W * L / ms * 1e3,
                               'note': 'lane per walker, 40 dependent count=1 samples'}
# node2vec, 100K walkers, 10 steps
w2 = starts[:100_000]
ms, _ = timed(lambda: G.random_walk(w2, [[0]] * 10, 0.25, 4.0, N + 1, call_id=9), iters=3, warm=1)
res['node2vec_len10_100K'] = {'ms': round(ms, 3), 'steps_per_s': 100_000 * 10 / ms * 1e3}

# full neighbours of 131072 uniform roots
roots = torch.randint(1, N + 1, (131072,), generator=gen, device='cuda')
ms, full = timed(lambda: G.get_full_neighbor(roots, [0]), iters=5)
tot = full[1].numel()
res['get_full_neighbor_131072'] = {'ms': round(ms, 3), 'neighbors': tot,
                                   'GBps': round((tot * (12 + 16) + roots.numel() * 40) / ms / 1e6, 1),
                                   'note': 'two passes (count + fill) with a host sync for the total'}
ms, _ = timed(lambda: G.get_top_k_neighbor(roots, [0], 5, -1), iters=5)
res['get_top_k_neighbor_131072_k5'] = {'ms': round(ms, 3)}

# message passing on a sampled block: E = 3.28M edges into 131072 rows, D = 128
D = 128
Eb = 131072 * 25
feat = torch.randn(Eb, D, device='cuda')
dst = torch.arange(131072, device='cuda', dtype=torch.int32).repeat_interleave(25)
ms, _ = timed(lambda: ops.scatter_add(feat, dst, 131072))
res['scatter_add_E3.28M_D128'] = {'ms': round(ms, 3),
                                  'GBps': round((4 * Eb * D + 4 * Eb + 4 * 131072 * D) / ms / 1e6, 1)}
ms, _ = timed(lambda: ops.scatter_max(feat, dst, 131072))
res['scatter_max_E3.28M_D128'] = {'ms': round(ms, 3),
                                  'GBps': round((4 * Eb * D + 4 * Eb + 4 * 131072 * D) / ms / 1e6, 1)}
table = torch.randn(2_000_000, D, device='cuda')
idx = torch.randint(0, 2_000_000, (Eb,), generator=gen, device='cuda', dtype=torch.int32)
ms, _ = timed(lambda: ops.gather(table, idx))
res['gather_E3.28M_D128'] = {'ms': round(ms, 3), 'GBps': round((8 * Eb * D + 4 * Eb) / ms / 1e6, 1)}

# ID_UNIQUE op (first-occurrence order) and the sharded front end on the hop-2 roots
out = G.sample_fanout(roots, [[0], [0]], [25, 10], N + 1, call_id=0)
hop2 = out[0][1].contiguous()
ms, u = timed(lambda: ops.id_unique(hop2), iters=5)
res['id_unique_3.28M'] = {'ms': round(ms, 3), 'unique': int(u[0].numel())}
ms, d = timed(lambda: ops.dedup_split(hop2, 8, 8), iters=5)
res['dedup_split_3.28M_8shards'] = {'ms': round(ms, 3), 'distinct_sent': int(d[1].numel())}
ms, _ = timed(lambda: ops.id_split(hop2, 8, 8), iters=5)
res['id_split_3.28M_8shards'] = {'ms': round(ms, 3)}
# dense feature fetch: 2M-node table of 128 floats (1 GB), 3.28M random + repeated nodes
import numpy as np
nf, D = 2_000_000, 128
ids = np.arange(1, nf + 1).astype(np.uint64)
rp = np.arange(nf + 1, dtype=np.int64)
Gf = euler_amd.Graph.from_csr(ids, rp, np.ones(nf, np.int32), ids[::-1].copy(),
                              np.ones(nf, np.float32), np.ones(nf, np.float32), 1,
                              features=(1, np.arange(nf + 1, dtype=np.int64) * D,
                                        np.full(nf, D, np.int32),
                                        np.random.default_rng(0).standard_normal(nf * D).astype(np.float32)))
q = torch.randint(1, nf + 1, (Eb,), generator=gen, device='cuda')
from euler_amd import _lib
for vec4 in (1, 0, 1):
    _lib.lib().euler_gpu_set_tuning(8, vec4)
    ms, _ = timed(lambda: Gf.get_dense_feature(q, [0], [D]))
    res['get_dense_feature_3.28M_D128_vec4=%d' % vec4] = {
        'ms': round(ms, 3), 'GBps': round((8 * Eb * D + 8 * Eb + 16 * Eb) / ms / 1e6, 1)}
print(json.dumps(res, indent=1))
