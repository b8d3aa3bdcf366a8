"""SampleNodeKernel by HIP events: 32M draws over 100M-node alias tables, 4 node types, type -1 (a
type draw per sample) and one fixed type.  EULER_GPU_LIB_PATH picks the library (A/B on one box)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, euler_amd
N, SEED = 100_000_000, 20240521
G = euler_amd.Graph.synthetic(euler_amd.synth_params(SEED, N, 10 * N, weighted=True)); G.set_seed(SEED)
ids = np.arange(1, N + 1, dtype=np.uint64)
types = (((ids * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(61)) & np.uint64(3)).astype(np.int32)
weights = (0.5 + 4.0 * np.random.default_rng(11).random(N, dtype=np.float32)).astype(np.float32)
G.set_node_sampler(None, types, weights, 4)
M = 32 * 1024 * 1024
for t in (-1, 1, [0, 2]):
    for i in range(3):
        out = G.sample_node(M, t, call_id=i)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
    ev[0].record()
    for i in range(10):
        out = G.sample_node(M, t, call_id=i); ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(10))
    print("RESULT sample_node type %s: median %.4f ms  min %.4f  (%.1f G draws/s)  checksum %d"
          % (t, ms[5], ms[0], M / ms[5] / 1e6, int(out.sum().item() & 0xFFFFFFFF)))
