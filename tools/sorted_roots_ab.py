"""Round 6, last sessions: does the ORDER of the roots matter to the metric kernel?  The same 131 072 roots as drawn
(random order), ascending, and bucketed by their top bits (a cheap partial order) - one stream, HIP events.
The per-request cost of the kernel's read side is an address-translation cost: ascending roots keep hop 1's
records and blocks inside the translation caches' reach.   python tools/sorted_roots_ab.py"""
import sys
sys.path.insert(0, '.')
import torch, euler_amd
N = 100_000_000
G = euler_amd.Graph.synthetic(euler_amd.synth_params(20240521, N, 10 * N, weighted=True))
G.set_seed(20240521)
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
B, steps = 131072, 12
roots = torch.randint(1, N + 1, (steps, B), generator=gen, device='cuda')
variants = {"as drawn": roots,
            "ascending": torch.sort(roots, dim=1).values,
            "by top 6 bits of 27": torch.gather(roots, 1, torch.argsort(roots >> 21, dim=1, stable=True))}
for name, r in variants.items():
    for i in range(3):
        G.sample_fanout(r[i], [[0], [0]], [25, 10], N + 1, call_id=2 * i)
    res = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for i in range(steps):
            G.sample_fanout(r[i], [[0], [0]], [25, 10], N + 1, call_id=2 * i)
        e1.record(); torch.cuda.synchronize()
        res.append(round(e0.elapsed_time(e1) / steps, 4))
    print("RESULT roots %-22s %s ms per step (one stream)" % (name, res), flush=True)
