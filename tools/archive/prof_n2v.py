"""node2vec (p = 0.25, q = 4) on the metric graph: time against the number of walkers
(a tail-bound kernel does not get faster with fewer walkers) and the distribution of
the list lengths the steps meet (export_rows gives the degree of every visited node).

  python tools/prof_n2v.py [nodes] [edges]
"""
import json
import sys

sys.path.insert(0, '.')
import numpy as np
import torch
import euler_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
E = int(sys.argv[2]) if len(sys.argv) > 2 else 10 * N
G = euler_amd.Graph.synthetic(euler_amd.synth_params(20240521, N, E, weighted=True))
G.set_seed(20240521)
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
starts = torch.randint(1, N + 1, (100_000,), generator=gen, device='cuda', dtype=torch.int64)
L = 10
et = [[0]] * L


def timed(fn, iters=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


from euler_amd import _lib
Lb = _lib.lib()
res = {}
ref = None
for mode, big in ((2, 0), (3, 4096), (3, 16384), (3, 65536)):
    Lb.euler_gpu_set_tuning(7, mode)
    Lb.euler_gpu_set_tuning(25, big)
    for W in (1000, 100_000):
        s = starts[:W].contiguous()
        ms = timed(lambda: G.random_walk(s, et, 0.25, 4.0, N + 1, call_id=3))
        key = "key7=%d big=%d walkers=%d" % (mode, big, W)
        res[key] = {"ms": round(ms, 3), "steps_per_s": round(W * L / ms * 1e3)}
        print(key, res[key], flush=True)
    out = G.random_walk(starts, et, 0.25, 4.0, N + 1, call_id=3)
    if ref is None:
        ref = out
    assert torch.equal(out, ref), "walks differ between modes"
Lb.euler_gpu_set_tuning(7, 3)
Lb.euler_gpu_set_tuning(25, 8192)
w = G.random_walk(starts, et, 0.25, 4.0, N + 1, call_id=3)
seg = G.get_edge_sum_weight  # noqa: F841 (kept for interactive use)
# degrees of the visited nodes: full-neighbour index of the walk's nodes
flat = w.reshape(-1)
idx = G.get_full_neighbor(flat, [0])[0]
deg = (idx[:, 1] - idx[:, 0]).reshape(w.shape).cpu().numpy().astype(np.int64)
cur = deg[:, :-1]                       # child list of step s = neighbours of position s
par = np.concatenate([np.zeros((len(deg), 1), np.int64), deg[:, :-2]], axis=1)
work = cur + 0 * par
per_walker = cur.sum(axis=1)
q = [50, 90, 99, 99.9, 100]
res["child_list_len percentiles " + str(q)] = [int(x) for x in np.percentile(cur.reshape(-1), q)]
res["per_walker_total_child_entries percentiles " + str(q)] = [int(x) for x in np.percentile(per_walker, q)]
res["mean_child_entries_per_step"] = float(cur.mean())
res["mean_parent_entries_per_step"] = float(par.mean())
print(json.dumps(res))
