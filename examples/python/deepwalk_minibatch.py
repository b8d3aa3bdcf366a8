#!/usr/bin/env python3
"""A DeepWalk / node2vec input pipeline on the tf_euler operator surface, end to end on one MI355X.

What examples/deepwalk/deepwalk.py:47-63 (`BaseNode2Vec.to_sample`) does per training step in the
reference - random_walk from the batch's source nodes, gen_pair over every path, sample_node for
batch x pairs x num_negs negatives - with the same function names (euler_amd.euler_ops mirrors
tf_euler.python.euler_ops; tensors are torch tensors in HBM).  Nothing returns to the host.

    python examples/python/deepwalk_minibatch.py [--data DIR] [--batch 1024] [--steps 50]

--data: a directory written by euler/tools (euler.meta + Node/*.dat); without it a synthetic
power-law graph with unit node weights is used (the global node sampler is then built with
Graph.set_node_sampler, the reference's Graph::BuildGlobalSampler as a call of its own).
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import euler_amd                                   # noqa: E402
from euler_amd import euler_ops                    # noqa: E402


def to_sample(inputs, node_type, edge_type, max_id, walk_len=3, walk_p=1.0, walk_q=1.0,
              left_win_size=1, right_win_size=1, num_negs=5):
    """BaseNode2Vec.to_sample (examples/deepwalk/deepwalk.py:47-63), line by line:
    inputs [batch] -> (src [batch * pairs, 1], pos [batch * pairs, 1], negs [batch * pairs, num_negs])."""
    inputs = inputs.reshape(-1)
    batch_size = inputs.numel()
    path = euler_ops.random_walk(inputs, [edge_type] * walk_len, p=walk_p, q=walk_q,
                                 default_node=max_id + 1)
    pair = euler_ops.gen_pair(path, left_win_size, right_win_size)
    num_pairs = pair.shape[1]
    src, pos = pair[..., 0], pair[..., 1]
    src = src.reshape(batch_size * num_pairs, 1)
    pos = pos.reshape(batch_size * num_pairs, 1)
    negs = euler_ops.sample_node(batch_size * num_pairs * num_negs, node_type)
    negs = negs.reshape(batch_size * num_pairs, num_negs)
    return src, pos, negs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default="")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--walk-len", type=int, default=3)          # run_deepwalk.py:34
    ap.add_argument("--num-negs", type=int, default=5)          # run_deepwalk.py:39
    ap.add_argument("--p", type=float, default=1.0)
    ap.add_argument("--q", type=float, default=1.0)
    ap.add_argument("--nodes", type=int, default=2_000_000)
    a = ap.parse_args()
    if a.data:
        euler_ops.initialize_graph({"mode": "local", "data_path": a.data})       # euler_ops/base.py
        G = euler_ops.get_default_graph()
        max_id = int(G.id_range()[0])
    else:
        G = euler_amd.Graph.synthetic(euler_amd.synth_params(1, a.nodes, 10 * a.nodes, weighted=True))
        G.set_node_sampler()                       # unit weights, one node type, row order
        euler_ops.set_default_graph(G)
        max_id = a.nodes
    G.set_seed(42)

    def step():
        inputs = euler_ops.sample_node(a.batch, 0)           # the source nodes of the step
        return to_sample(inputs, 0, [0], max_id, a.walk_len, a.p, a.q, 1, 1, a.num_negs)

    src, pos, negs = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        src, pos, negs = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print("batch %d, walk_len %d, %d negatives: %.3f ms per training-step input (%d pairs, %d negatives), "
          "%.0f steps/s" % (a.batch, a.walk_len, a.num_negs, dt * 1e3, src.shape[0], negs.numel(), 1.0 / dt))


if __name__ == "__main__":
    main()
