"""Round 6: does every call of a repeated workload take the same time?  (One hipMallocAsync in a few hundred took
seconds in the sharded node2vec walk - tools/sharded_n2v_ab.py; this looks at the single-GPU entry points that
allocate stream-ordered scratch per call.)  python tools/stall_check.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, euler_amd
N, SEED = 100_000_000, 20240521
G = euler_amd.Graph.synthetic(euler_amd.synth_params(SEED, N, 10 * N, weighted=True)); G.set_seed(SEED)
gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
def run(name, fn, reps):
    fn(); torch.cuda.synchronize()
    ts = []
    for i in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.array(ts)
    print("RESULT %s: median %.3f ms, max %.3f ms, calls over 3x the median: %d of %d"
          % (name, np.median(ts), ts.max(), int((ts > 3 * np.median(ts)).sum()), reps), flush=True)
starts = torch.randint(1, N + 1, (1_000_000,), generator=gen, device="cuda", dtype=torch.int64)
run("deepwalk 1M x 40", lambda: G.random_walk(starts, [[0]] * 40, 1.0, 1.0, N + 1, call_id=1), 60)
s2 = starts[:100_000].contiguous()
run("node2vec 100K x 10", lambda: G.random_walk(s2, [[0]] * 10, 0.25, 4.0, N + 1, call_id=3), 40)
r = starts[:131072].contiguous()
run("sage_blocks 131072", lambda: G.sage_blocks(r, [[0], [0]], [25, 10], default_node=N + 1, sync=False), 100)
run("fanout 131072", lambda: G.sample_fanout(r, [[0], [0]], [25, 10], N + 1, call_id=5), 200)
run("full neighbours 131072", lambda: G.get_full_neighbor(r, [0]), 100)
run("unique 3.28M", lambda: euler_amd.ops.id_unique(starts[:3_276_800] % 300_000), 100)
