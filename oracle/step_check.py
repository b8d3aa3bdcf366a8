"""TEST INFRASTRUCTURE (the checker, never the product path; callers: tests/, bench.py's
post-timing parity check).  A whole 2-hop fanout step of the GPU against the CPU oracle at
FULL size without holding the whole graph on the host:

  hop 1   every root's row is exported from HBM, the oracle samples all of them
          (Node::SampleNeighbor, core/graph/node.cc:98-161, TF layout of
          tf_euler/kernels/sample_neighbor_op.cc:54-129) - every hop-1 id / weight / type is
          compared;
  hop 2   a row is a pure function of (seed, call id, node id) (DESIGN "RNG contract"; the
          reference's ID_UNIQUE -> sample -> GATHER rewrite, parser/compiler.cc:76-90, gives
          every position of a node the same row): on the GPU every position's row is compared
          with the row at the FIRST position of its node (all positions, torch), and the rows
          at those first positions - one per distinct hop-1 child - go to the oracle, chunk by
          chunk of exported rows.  Together: every hop-2 sample of the step is verified.

Returns the number of sampled edges verified; raises AssertionError on the first difference."""
import numpy as np


def check_fanout_step(G, OracleGraph, CSR, seed, call_id, roots, gn, gw, gt, fanout, default_node,
                      n_nodes, chunk_nodes=8192, edge_type=0):
    """G: euler_amd.Graph (plain graph, ids 1 .. n_nodes); roots: 1-D torch int64 (device);
    (gn, gw, gt) = G.sample_fanout(roots, [[t], [t]], fanout, default_node, call_id=call_id)."""
    import torch
    c1, c2 = int(fanout[0]), int(fanout[1])
    B = roots.numel()
    h1_id = gn[1].reshape(B, c1)
    h1_w, h1_t = gw[0].reshape(B, c1), gt[0].reshape(B, c1)
    h2_id = gn[2].reshape(B * c1, c2)
    h2_w, h2_t = gw[1].reshape(B * c1, c2), gt[1].reshape(B * c1, c2)

    def oracle_for(ids_np):
        need = np.unique(ids_np.astype(np.int64).view(np.uint64))
        need = need[(need >= 1) & (need <= n_nodes)]
        rp, te, nb, pw, tp = G.export_rows(need)
        return OracleGraph(CSR(need, rp, te, nb, pw, tp, 1))

    # ---- hop 1: every root
    r_np = roots.cpu().numpy()
    for lo in range(0, B, 262144):
        part = r_np[lo:lo + 262144]
        OG = oracle_for(part)
        on, ow, ot = OG.sample_neighbor(seed, call_id, part, [edge_type], c1, default_node)
        sl = slice(lo, lo + len(part))
        assert np.array_equal(on, h1_id[sl].cpu().numpy()), "hop-1 ids differ from the oracle"
        assert np.array_equal(ow, h1_w[sl].cpu().numpy()), "hop-1 weights differ from the oracle"
        assert np.array_equal(ot, h1_t[sl].cpu().numpy()), "hop-1 types differ from the oracle"
        del OG
    # ---- hop 2: all positions against the first position of their node (device) ...
    children = h1_id.reshape(-1)
    uniq, inv = torch.unique(children, return_inverse=True)
    pos = torch.arange(children.numel(), device=children.device, dtype=torch.int64)
    first = torch.full((uniq.numel(),), children.numel(), device=children.device, dtype=torch.int64)
    first.scatter_reduce_(0, inv, pos, reduce="amin")
    rep = first[inv]
    step = 4 * 1024 * 1024
    for lo in range(0, children.numel(), step):
        sl = slice(lo, lo + step)
        assert torch.equal(h2_id[sl], h2_id[rep[sl]]), "two positions of one node hold different hop-2 ids"
        assert torch.equal(h2_w[sl], h2_w[rep[sl]]) and torch.equal(h2_t[sl], h2_t[rep[sl]]), \
            "two positions of one node hold different hop-2 weights / types"
    # ... and the first positions' rows against the oracle, chunk by chunk of exported rows
    u_np = uniq.cpu().numpy()
    rid = h2_id[first].cpu().numpy()
    rw = h2_w[first].cpu().numpy()
    rt = h2_t[first].cpu().numpy()
    for lo in range(0, len(u_np), chunk_nodes):
        part = u_np[lo:lo + chunk_nodes]
        OG = oracle_for(part)
        on, ow, ot = OG.sample_neighbor(seed, call_id + 1, part, [edge_type], c2, default_node)
        sl = slice(lo, lo + len(part))
        assert np.array_equal(on, rid[sl]), "hop-2 ids differ from the oracle (children %d..)" % lo
        assert np.array_equal(ow, rw[sl]), "hop-2 weights differ from the oracle (children %d..)" % lo
        assert np.array_equal(ot, rt[sl]), "hop-2 types differ from the oracle (children %d..)" % lo
        del OG
    return int(B * c1 + B * c1 * c2), int(uniq.numel())
