"""A/B of two builds of the library on the metric step (plain 100M / 1B graph): the kernel alone
(euler_gpu_time_sample_fanout) and the three-stream loop.  python tools/ab_lib_plain.py [--lib PATH]"""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import torch, euler_amd, bench
from euler_amd import _lib
if '--lib' in sys.argv:
    _lib.LIB_PATH = sys.argv[sys.argv.index('--lib') + 1]
L = _lib.lib()
N, B = 100_000_000, 131072
G = euler_amd.Graph.synthetic(euler_amd.synth_params(bench.GRAPH_SEED, N, 10 * N, weighted=True))
G.set_seed(1)
gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
roots = torch.randint(1, N + 1, (16, B), generator=gen, device="cuda", dtype=torch.int64)
side = [torch.cuda.Stream() for _ in range(3)]
def loop(a, b):
    for i in range(a, b):
        with torch.cuda.stream(side[i % 3]):
            G.sample_fanout(roots[i % 16], [[0], [0]], [25, 10], N + 1, call_id=2 * i)
loop(0, 12); torch.cuda.synchronize()
res = []
for rep in range(3):
    t0 = time.perf_counter(); loop(0, 400); torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 400 * 1e3)
one = []
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(100):
        G.sample_fanout(roots[i % 16], [[0], [0]], [25, 10], N + 1, call_id=2 * i)
    torch.cuda.synchronize()
    one.append((time.perf_counter() - t0) / 100 * 1e3)
print("three streams %s ms/step; one stream %s" % ([round(x, 4) for x in res], [round(x, 4) for x in one]))
