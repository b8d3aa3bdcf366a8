"""A/B of how SageDataFlow's one host read is done (Tensor.cpu() vs a pinned buffer + stream
synchronize): ms per minibatch of Graph.sage_blocks(sync=True), 16 384 roots, fanout [25, 10]."""
import sys, time
sys.path.insert(0, '.')
import torch, euler_amd
from euler_amd import graph as G_
N = 100_000_000
G = euler_amd.Graph.synthetic(euler_amd.synth_params(20240521, N, 10 * N, weighted=True))
G.set_seed(20240521)
gen = torch.Generator(device='cuda'); gen.manual_seed(77)
r = torch.randint(1, N + 1, (16384,), generator=gen, device='cuda', dtype=torch.int64)
pinned = G_._read_counts
_buf = torch.empty(16, dtype=torch.int32, pin_memory=True)
def pinned_kept(counts):
    host = _buf[:counts.numel()]
    host.copy_(counts, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return host.tolist()
_ev = torch.cuda.Event()
def pinned_event(counts):
    host = _buf[:counts.numel()]
    host.copy_(counts, non_blocking=True)
    _ev.record()
    _ev.synchronize()
    return host.tolist()
def pageable(counts):
    return [int(c) for c in counts.cpu().tolist()]
def run(steps):
    for i in range(20):
        G.sage_blocks(r, [[0], [0]], [25, 10], default_node=N + 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        G.sage_blocks(r, [[0], [0]], [25, 10], default_node=N + 1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for rep in range(3):
    for name, fn in (("pinned", pinned), ("pinned_kept", pinned_kept), ("pinned_event", pinned_event), ("pageable", pageable)):
        G_._read_counts = fn
        print(name, "%.4f ms" % run(300))
G_._read_counts = pinned
t0 = time.perf_counter()
for i in range(300):
    G.sage_blocks(r, [[0], [0]], [25, 10], default_node=N + 1, sync=False)
torch.cuda.synchronize()
print("no read %.4f ms" % ((time.perf_counter() - t0) / 300 * 1e3))
