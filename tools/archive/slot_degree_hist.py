"""Degrees of the hop-2 rows of the metric step (the distinct children of 131 072 roots x 25
samples): how many of the one-kernel fanout's level-window / leaf lines come from rows of
which size.  -> gpurun_out/slot_degree_hist.json"""
import json, sys
sys.path.insert(0, '.')
import numpy as np, torch, euler_amd

N = 100_000_000
p = euler_amd.synth_params(20240521, N, 1_000_000_000, weighted=True)
G = euler_amd.Graph.synthetic(p)
G.set_seed(20240521)
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
roots = torch.randint(1, N + 1, (131072,), generator=gen, device='cuda')
nb, w, t = G.sample_neighbor(roots, [0], 25, N + 1, call_id=0)
# slots: distinct children per tile of 4 roots
tile = torch.arange(131072, device='cuda').div(4, rounding_mode='floor').repeat_interleave(25)
key = tile * (N + 2) + nb.reshape(-1)
slots = torch.unique(key) % (N + 2)
idx = G.get_full_neighbor(slots, [0])[0]
deg = (idx[:, 1] - idx[:, 0]).cpu().numpy().astype(np.int64)
edges = [0, 1, 10, 20, 50, 100, 280, 320, 1000, 1600, 10000, 40000, 10**6, 10**9]
rows = []
for a, b in zip(edges[:-1], edges[1:]):
    m = (deg >= a) & (deg < b)
    d = deg[m]
    nb_blocks = np.maximum(1, (d + 9) // 10)
    leaves = np.minimum(10, nb_blocks) * 0 + nb_blocks * (1 - (1 - 1 / nb_blocks) ** 10)   # expected distinct leaf lines of 10 draws
    rows.append({'deg': [a, b], 'slots': int(m.sum()), 'share': round(float(m.mean()), 4),
                 'expected_leaf_lines': int(leaves.sum())})
out = {'slots': int(len(deg)), 'mean_deg': float(deg.mean()), 'median_deg': float(np.median(deg)), 'classes': rows}
print(json.dumps(out, indent=1))
json.dump(out, open('gpurun_out/slot_degree_hist.json', 'w'), indent=1)
