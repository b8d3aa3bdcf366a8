"""DeepWalk on the metric graph: how many DISTINCT current nodes do the walkers of a step stand
on?  Walkers on one node in one step draw the same next node (the draw is keyed by node id and
step - the reference's ID_UNIQUE semantics), so the distinct count is the step's real work.
  python tools/walk_coincidence.py [walkers] [steps]"""
import json, sys
sys.path.insert(0, '.')
import torch, euler_amd
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 40
N = 100_000_000
G = euler_amd.Graph.synthetic(euler_amd.synth_params(20240521, N, 1_000_000_000, weighted=True))
G.set_seed(20240521)
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
starts = torch.randint(1, N + 1, (W,), generator=gen, device='cuda')
walks = G.random_walk(starts, [[0]] * L, 1.0, 1.0, N + 1, call_id=0)
frac = []
for s in range(L + 1):
    col = walks[:, s]
    frac.append(round(torch.unique(col).numel() / W, 4))
deg_like = {}
print(json.dumps({'walkers': W, 'steps': L, 'distinct_fraction_per_step': frac,
                  'mean_over_steps_1..L-1': round(sum(frac[1:L]) / (L - 1), 4)}))
