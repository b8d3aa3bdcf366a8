"""random_walk (p = q = 1) on the metric graph: time against the number of walkers."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, euler_amd
N = 100_000_000
G = euler_amd.Graph.synthetic(euler_amd.synth_params(20240521, N, 10 * N, weighted=True))
G.set_seed(20240521)
gen = torch.Generator(device='cuda'); gen.manual_seed(1234)
starts = torch.randint(1, N + 1, (4_000_000,), generator=gen, device='cuda', dtype=torch.int64)
from euler_amd import _lib
res = {}
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 6
_lib.lib().euler_gpu_set_tuning(0, variant)
_lib.lib().euler_gpu_set_tuning(2, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ref = None
for W in (250_000, 1_000_000):
    for L in (40,):
        s = starts[:W].contiguous(); et = [[0]] * L
        G.random_walk(s, et, 1.0, 1.0, N + 1, call_id=0); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(5):
            G.random_walk(s, et, 1.0, 1.0, N + 1, call_id=L * i)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        res["walkers=%d len=%d" % (W, L)] = {"ms": round(ms, 3), "G_steps_per_s": round(W * L / ms / 1e6, 2)}
        out = G.random_walk(s, et, 1.0, 1.0, N + 1, call_id=0)
        print("variant=%d walkers=%d len=%d" % (variant, W, L), res["walkers=%d len=%d" % (W, L)],
              "checksum", int(out.sum().item()), flush=True)
