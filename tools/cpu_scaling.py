"""Reference sampler (oracle/_ref) throughput vs host threads and graph size."""
import sys, os, time, json
sys.path.insert(0, '.')
import numpy as np
from oracle import oracle as O
res = {}
cores = os.cpu_count()
for n in (1_000_000, 4_000_000):
    po = O.synth_params(20240521, n, 10 * n, weighted=True)
    t0 = time.time(); csr = O.synth_csr(po)
    w = csr.prefix_w.copy(); w[1:] -= csr.prefix_w[:-1]
    starts = csr.row_ptr[:-1]; w[starts] = csr.prefix_w[starts]
    R = O.RefGraph.build_raw(csr.row_id, csr.row_ptr.copy(), csr.nbr, w, 1)
    build = time.time() - t0
    rng = np.random.default_rng(1)
    for threads in (8, 32, 64, 128, cores):
        batch, iters = 1024, max(64, threads * 8)
        roots = rng.integers(1, n + 1, batch * iters).astype(np.uint64)
        R.bench_fanout(20240521, roots[:batch * 4], batch, 4, [25, 10], threads)
        secs, edges = R.bench_fanout(20240521, roots, batch, iters, [25, 10], threads)
        res['n=%d threads=%d' % (n, threads)] = {'Medges_per_s': round(edges / secs / 1e6, 1), 'secs': round(secs, 2), 'build_s': round(build, 1)}
        print(n, threads, res['n=%d threads=%d' % (n, threads)], flush=True)
print(json.dumps(res))
