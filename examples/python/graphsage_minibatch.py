#!/usr/bin/env python3
"""A GraphSAGE input pipeline on the tf_euler operator surface, end to end on one MI355X.

What examples/graphsage/run_graphsage.py + tf_euler/python/dataflow/sage_dataflow.py do per
training step in the reference - sample a batch of source nodes, build the 2-hop SageDataFlow
(sample_neighbor of the unique frontier per hop, tf.unique, edge_index), fetch the dense
features of the outermost layer and mean-aggregate them block by block - with the same
function names (euler_amd.euler_ops mirrors tf_euler.python.euler_ops; tensors are torch
tensors in HBM).  Nothing returns to the host between the ops except the layer sizes.

    python examples/python/graphsage_minibatch.py [--data DIR] [--batch 1024] [--steps 50]

--data: a directory written by euler/tools (euler.meta + Node/*.dat, float feature 0 = the
input features); without it a synthetic power-law graph with a random feature table is used.
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import euler_amd                                   # noqa: E402
from euler_amd import euler_ops, ops               # noqa: E402
from euler_amd.dataflow import SageDataFlow        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default="")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--fanouts", type=int, nargs=2, default=[10, 5])      # run_graphsage.py:41-42
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--nodes", type=int, default=2_000_000)
    a = ap.parse_args()
    if a.data:
        euler_ops.initialize_graph({"mode": "local", "data_path": a.data})       # euler_ops/base.py
        G = euler_ops.get_default_graph()
        max_id = int(G.id_range()[0])
        feat = None
    else:
        G = euler_amd.Graph.synthetic(euler_amd.synth_params(1, a.nodes, 10 * a.nodes, weighted=True))
        euler_ops.set_default_graph(G)
        max_id = a.nodes
        feat = torch.randn(a.nodes + 2, a.dim, device="cuda")                   # row = node id
    G.set_seed(42)
    flow = SageDataFlow(G, a.fanouts, [[0], [0]], add_self_loops=True, max_id=max_id)

    def step():
        src = euler_ops.sample_node(a.batch, 0) if a.data else \
            torch.randint(1, a.nodes + 1, (a.batch,), device="cuda")
        df = flow(src)                                   # blocks from the outermost hop inwards
        x = euler_ops.get_dense_feature(df[0].n_id, [0], [a.dim])[0] if feat is None else feat[df[0].n_id]
        for blk in df:                                   # mean aggregation, the SAGE layer's input
            x = ops.gather_scatter("mean", x, blk.edge_index[1], blk.edge_index[0], blk.size[0])
        return x

    out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print("batch %d, fanouts %s: %.3f ms per training-step input (%d x %d aggregated rows), %.0f steps/s"
          % (a.batch, a.fanouts, dt * 1e3, out.shape[0], out.shape[1], 1.0 / dt))


if __name__ == "__main__":
    main()
