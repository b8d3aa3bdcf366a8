// Per-item logic of the layerwise-sampling ops (sampleLNB without a weight
// function: API_GET_EDGE_SUM_WEIGHT -> API_SAMPLE_ROOT -> API_SAMPLE_L ->
// API_SPARSE_GET_ADJ, euler/parser/translator.cc:338-386,489-527).  Every
// function is EG_HD: layer_kernels.hip runs them one item per lane on the
// GPU, tests/csrc/host_check.hip runs the same source on the CPU against the
// oracle (a check of the logic, not a product path).
#pragma once

#include <math.h>

#include "device_fns.h"

namespace euler_gpu {

// API_GET_EDGE_SUM_WEIGHT (core/kernels/get_edge_sum_weight_op.cc:33-66): the
// f32 sum, added in Node::GetFullNeighbor order (node.cc:175-197: listed
// types in listed order, storage order inside a type), of the per-edge weights
// `nw[j] - nw[j-1]`.  Unknown node: 0.
EG_HD float EdgeSumWeight(const GraphView& g, uint64_t id, const int32_t* et,
                          int32_t k) {
  const int64_t row = FindRow(g, id);
  float sum = 0.f;
  if (row < 0) return sum;
  const RowMeta m = LoadRowMeta(g, row);
  const float* nw = g.prefix_w + m.row_ptr;
  for (int32_t x = 0; x < k; ++x) {
    const int32_t t = et[x];
    if (t < 0 || t >= g.T) continue;
    const int32_t b = t == 0 ? 0 : m.type_end[t - 1];
    const int32_t e = m.type_end[t];
    for (int32_t p = b; p < e; ++p) {
      const float pre = p == 0 ? 0.f : nw[p - 1];
      sum = EG_FADD(sum, EG_FSUB(nw[p], pre));
    }
  }
  return sum;
}

// One batch row of API_SAMPLE_ROOT (core/kernels/sample_root_op.cc:62-66):
// FastWeightedCollection::Init (common/fast_weighted_collection.h:55-75: f32
// sum, weights divided by it) then AliasMethod::Init (common/alias_method.cc:
// 23-63: LIFO small / large stacks, avg = 1/n in double, weights updated in
// float, prob in float).  Slot j of every scratch array sits at j * st, so
// that the rows of a batch interleave (lanes = rows touch adjacent words).
// `stack` holds both stacks: small grows up from slot 0, large grows down
// from slot n-1 (an index is on at most one of them, so they never meet).
// Returns the f32 weight sum; a zero sum builds nothing (the op then emits
// default_node, sample_root_op.cc:74-78).
EG_HD float AliasBuildRow(const float* w_in, int32_t n, int64_t st, float* wn,
                          float* prob, int32_t* alias, int32_t* stack) {
  float sum = 0.f;
  for (int32_t i = 0; i < n; ++i) sum = EG_FADD(sum, w_in[i]);
  if (!(sum != 0.f)) return sum;
  const double avg = 1 / (double)n;
  int32_t ns = 0, nl = 0;
  for (int32_t i = 0; i < n; ++i) {
    const float x = EG_FDIV(w_in[i], sum);
    wn[i * st] = x;
    prob[i * st] = 0.f;
    alias[i * st] = 0;
    if ((double)x > avg) { stack[(int64_t)(n - 1 - nl) * st] = i; ++nl; }
    else { stack[(int64_t)ns * st] = i; ++ns; }
  }
  while (nl > 0 && ns > 0) {
    --ns;
    const int32_t less = stack[(int64_t)ns * st];
    const int32_t more = stack[(int64_t)(n - nl) * st];
    --nl;
    prob[less * st] = EG_FMUL(wn[less * st], (float)n);
    alias[less * st] = more;
    // weights_[more] = weights_[more] + weights_[less] - avg : a float add,
    // then a double subtract rounded back to float
    const float wm =
        (float)EG_DADD((double)EG_FADD(wn[more * st], wn[less * st]), -avg);
    wn[more * st] = wm;
    if ((double)wm > avg) { stack[(int64_t)(n - 1 - nl) * st] = more; ++nl; }
    else { stack[(int64_t)ns * st] = more; ++ns; }
  }
  while (ns > 0) { --ns; prob[(int64_t)stack[(int64_t)ns * st] * st] = 1.f; }
  while (nl > 0) { prob[(int64_t)stack[(int64_t)(n - nl) * st] * st] = 1.f; --nl; }
  return sum;
}

// Sample j of batch row b (sample_root_op.cc:79-81 -> FastWeightedCollection::
// Sample -> AliasMethod::Next, alias_method.cc:66-78): draws 2j (column) and
// 2j+1 (coin) of stream b = one Philox block.  Returns the chosen slot.
EG_HD int32_t SampleRootSlot(uint64_t seed, uint32_t call_id, int64_t b,
                             int32_t j, int32_t n, int64_t st,
                             const float* prob, const int32_t* alias) {
  const Philox4 r = RngBlock(seed, call_id, kDomainRoot, (uint64_t)b, (uint32_t)j);
  const double u_col = UnitFromWords(r.w[0], r.w[1]);
  const double u_coin = UnitFromWords(r.w[2], r.w[3]);
  int64_t col = (int64_t)floor(EG_DMUL((double)n, u_col));
  if (col >= n) col = n - 1;      // unreachable (u < 1); keeps the read inside the row
  return u_coin < (double)prob[col * st] ? (int32_t)col : alias[col * st];
}

// One position of API_SAMPLE_L (core/kernels/sample_layer_op.cc:54-70):
// euler::SampleNeighbor({root}, edge_types, 1); an empty result gives
// (default_node, 0, 0).  The RNG stream is the POSITION i, not the node id:
// API_SAMPLE_ROOT draws roots with replacement and the reference samples
// every occurrence independently.
EG_HD void SampleLayerAt(const GraphView& g, uint64_t seed, uint32_t call_id,
                         int64_t i, uint64_t root, const int32_t* et, int32_t k,
                         int64_t default_node, uint64_t* out_id, float* out_w,
                         int32_t* out_t) {
  RowSampler rs;
  InitRowSampler(rs, g, FindRow(g, root), et, k);
  if (!rs.valid) {
    *out_id = (uint64_t)default_node;
    *out_w = 0.f;
    *out_t = 0;
    return;
  }
  SampleAt(rs, seed, call_id, (uint64_t)i, 0, out_id, out_w, out_t, kDomainLayer);
}

// EdgeExist(EdgeId(src, dst, t)) for any listed t (core/kernels/
// sparse_get_adj_op.cc:63-67, core/api/api.cc:46-48), answered from the
// adjacency row of src: the reference's converter writes one Edge record per
// (src, dst, type) entry of the node rows, so the Edge map and the rows hold
// the same triples (pinned on the reference's own fixture, tests/).
EG_HD bool EdgeExistAny(const GraphView& g, int64_t row, uint64_t dst,
                        const int32_t* et, int32_t k) {
  if (row < 0) return false;
  const RowMeta m = LoadRowMeta(g, row);
  const uint64_t* nbr = g.nbr + m.row_ptr;
  for (int32_t x = 0; x < k; ++x) {
    const int32_t t = et[x];
    if (t < 0 || t >= g.T) continue;
    const int32_t b = t == 0 ? 0 : m.type_end[t - 1];
    const int32_t e = m.type_end[t];
    for (int32_t p = b; p < e; ++p)
      if (nbr[p] == dst) return true;
  }
  return false;
}

// Draw j of batch row b of API_LOCAL_SAMPLE_L (local_sample_layer_op.cc:128-145):
// CompactWeightedCollection::Sample = RandomSelect over the row's running sums
// (cnt entries at sum_w), or -1 for the rows the op memsets (no candidates /
// zero weight).  RNG: domain 6, stream = b, draw j.
EG_HD int64_t LocalLayerPick(uint64_t seed, uint32_t call_id, int64_t b, int32_t j,
                             const float* sum_w, int64_t cnt) {
  if (cnt == 0 || sum_w[cnt - 1] == 0.f) return -1;
  const double u = RngDraw(seed, call_id, kDomainLocalLayer, (uint64_t)b, (uint64_t)j);
  return (int64_t)RandomSelect(sum_w, 0, (uint64_t)(cnt - 1), u);
}

// Graph::GetNodeByID(id)->GetType() (core/api/api.cc:50-61, GetNodeType):
// DEFAULT_INT32 = numeric_limits<int32_t>::lowest() for an unknown node
// (common/data_types.cc:23).  node_type == nullptr: every node has type 0.
EG_HD int32_t NodeTypeOf(const GraphView& g, const int32_t* node_type, uint64_t id) {
  const int64_t row = FindRow(g, id);
  if (row < 0) return (int32_t)0x80000000;
  return node_type ? node_type[row] : 0;
}

// Sample j of ONE Graph::SampleNode(type, count) call (core/graph/graph.cc:
// 221-245), the call API_SAMPLE_N_WITH_TYPES makes per listed type
// (core/kernels/sample_n_with_types_op.cc:44-52).  RNG: domain NODE, stream =
// the index of the call inside the op, draws in the call's program order (2 per
// sample for a fixed type, 4 for type -1).  *ok = false where the reference
// returns an empty vector (zero weight) or indexes node_samplers_ out of range.
EG_HD uint64_t SampleNodeOfType(const NodeSamplerView& s, uint64_t seed,
                                uint32_t call_id, uint64_t stream, int32_t type,
                                int32_t j, bool* ok) {
  *ok = false;
  int32_t t = type;
  uint32_t block = (uint32_t)j;
  if (type == -1) {
    if (s.tc_sum == 0.f) return 0;
    const Philox4 b = RngBlock(seed, call_id, kDomainNode, stream, 2u * (uint32_t)j);
    const int64_t col = (int64_t)floor(EG_DMUL((double)s.n_types,
                                               UnitFromWords(b.w[0], b.w[1])));
    t = UnitFromWords(b.w[2], b.w[3]) < (double)s.tc_prob[col] ? (int32_t)col
                                                                : s.tc_alias[col];
    block = 2u * (uint32_t)j + 1u;
  } else {
    if (type < 0 || type >= s.n_types) return 0;
    if (s.sampler_sum[type] == 0.f || s.type_off[type + 1] == s.type_off[type]) return 0;
  }
  const Philox4 b = RngBlock(seed, call_id, kDomainNode, stream, block);
  const int64_t base = s.type_off[t];
  const int64_t n = s.type_off[t + 1] - base;
  // AliasMethod::Next (alias_method.cc:66-78)
  const int64_t column = (int64_t)floor(EG_DMUL((double)n, UnitFromWords(b.w[0], b.w[1])));
  const AliasEntry e = s.entries[base + column];
  *ok = true;
  return UnitFromWords(b.w[2], b.w[3]) < (double)e.prob ? e.id_self : e.id_alias;
}

}  // namespace euler_gpu
