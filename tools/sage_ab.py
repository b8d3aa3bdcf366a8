"""Round 6: euler_gpu_sage_blocks by HIP events for a few tuning settings on one box:
  python tools/sage_ab.py [key=value,...]..."""
import sys
sys.path.insert(0, '.')
import torch, euler_amd
from euler_amd import _lib
L = _lib.lib()
N = 100_000_000
G = euler_amd.Graph.synthetic(euler_amd.synth_params(20240521, N, 10 * N, weighted=True)); G.set_seed(20240521)
gen = torch.Generator(device="cuda"); gen.manual_seed(77)
for cfg in (sys.argv[1:] or [""]):
    for kv in filter(None, cfg.split(",")):
        k, v = kv.split("="); _lib.check(L.euler_gpu_set_tuning(int(k), int(v)))
    for B in (1024, 16384, 131072):
        r = torch.randint(1, N + 1, (B,), generator=gen, device="cuda", dtype=torch.int64)
        for i in range(3):
            G.sage_blocks(r, [[0], [0]], [25, 10], default_node=N + 1, sync=False)
        res = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for i in range(10):
                G.sage_blocks(r, [[0], [0]], [25, 10], default_node=N + 1, sync=False)
            e1.record(); torch.cuda.synchronize()
            res.append(round(e0.elapsed_time(e1) / 10, 4))
        print("RESULT cfg '%s' B=%d: %s ms per flow" % (cfg, B, res), flush=True)
