// ORACLE (test infrastructure, NOT product code).
//
// C-ABI harness around the REFERENCE sampler, compiled unmodified from
// /root/reference by oracle/Makefile into oracle/_ref/libeuler_ref.so.  The
// only reference file replaced is euler/common/random.cc (see
// ref_seam_random.cc).  Compiled with -fno-access-control so that the harness
// can read the reference's private storage (Node::neighbor_info_,
// Graph::node_map_, AliasMethod::prob_/alias_) without touching its sources.
//
// What is REFERENCE code here : Node::Init / Node::SampleNeighbor /
//   Node::GetFullNeighbor (core/graph/node.cc), Graph::SampleNode /
//   BuildGlobalSampler / Graph::Init + GraphBuilder (core/graph/graph.cc,
//   graph_builder.cc), CompactWeightedCollection, AliasMethod,
//   FastWeightedCollection (euler/common).
// What is RESTATED here (their sources need protobuf / TensorFlow and cannot
//   be compiled): the batch loop of api.cc:223-236 (so the RNG context can be
//   set per root), the empty-row fill of core/kernels/sample_neighbor_op.cc:
//   134-143, FillNeighbor (core/kernels/common.cc:275-334) and the TF
//   RandomWalk kernel (tf_euler/kernels/random_walk_op.cc:83-168,207-247);
//   the bodies of the layerwise ops (core/kernels/get_edge_sum_weight_op.cc,
//   sample_root_op.cc, sample_layer_op.cc, sparse_get_adj_op.cc) and the
//   sparse assembly of tf_euler/kernels/sparse_get_adj_op.cc:92-124, each
//   around the reference's own euler::GetFullNeighbor / SampleNeighbor /
//   EdgeExist, FastWeightedCollection and std::set.
#include <stdint.h>
#include <string.h>

#include <cmath>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "euler/common/compact_weighted_collection.h"
#include "euler/common/data_types.h"
#include "euler/common/fast_weighted_collection.h"
#include "euler/common/server_register.h"
#include "euler/core/api/api.h"
#include "euler/core/graph/edge.h"
#include "euler/core/graph/graph.h"
#include "euler/core/graph/node.h"

#include "eo_rng.h"

extern "C" void euler_ref_set_rng(uint64_t seed, uint32_t call_id,
                                  uint32_t domain, uint64_t stream);

namespace euler {
// graph.cc references the ZooKeeper register only in Deregister paths; the
// zk sources are not part of the hot path and are not linked.
std::shared_ptr<ServerRegister> GetServerRegister(const std::string&,
                                                  const std::string&) {
  return nullptr;
}
}  // namespace euler

namespace {

euler::Graph& G() { return euler::Graph::Instance(); }

void SetTypeMaps(int n_node_types, int n_edge_types) {
  auto& meta = G().meta_;
  meta.node_type_map_.clear();
  meta.edge_type_map_.clear();
  for (int i = 0; i < n_node_types; ++i)
    meta.node_type_map_[std::to_string(i)] = i;
  for (int i = 0; i < n_edge_types; ++i)
    meta.edge_type_map_[std::to_string(i)] = i;
  meta.partitions_num_ = 1;
}

}  // namespace

extern "C" {

int euler_ref_graph_clear() {
  auto& g = G();
  for (auto& kv : g.node_map_) delete kv.second;
  g.node_map_.clear();
  for (auto& kv : g.edge_map_) delete kv.second;
  g.edge_map_.clear();
  g.edge_id_map_.clear();
  g.node_samplers_.clear();
  g.node_weight_sums_.clear();
  g.node_type_collection_ = euler::common::FastWeightedCollection<int32_t>();
  g.global_sampler_ok_ = false;
  g.initialized_ = false;
  return 0;
}

// Typed adjacency in (node, type)-major CSR: segment (i, t) is
// [seg_ptr[i*T+t], seg_ptr[i*T+t+1]) of nbr/w (RAW weights).  Prefix sums are
// produced by the reference's own Node::Init (node.cc:37-96).
int euler_ref_graph_build(int64_t n, const uint64_t* ids,
                          const int32_t* node_type, const float* node_weight,
                          int32_t T, int32_t n_node_types,
                          const int64_t* seg_ptr, const uint64_t* nbr,
                          const float* w) {
  euler_ref_graph_clear();
  auto& g = G();
  g.reserveNodeMap(n);
  std::vector<std::vector<uint64_t>> nb(T);
  std::vector<std::vector<float>> nw(T);
  std::vector<std::vector<uint64_t>> f_u64;
  std::vector<std::vector<float>> f_f32;
  std::vector<std::string> f_bin;
  for (int64_t i = 0; i < n; ++i) {
    for (int t = 0; t < T; ++t) {
      int64_t b = seg_ptr[i * T + t], e = seg_ptr[i * T + t + 1];
      nb[t].assign(nbr + b, nbr + e);
      nw[t].assign(w + b, w + e);
    }
    auto* node = new euler::Node(ids[i], node_weight[i], node_type[i]);
    if (!node->Init(nb, nw, f_u64, f_f32, f_bin)) return -1;
    g.AddNode(node);
  }
  SetTypeMaps(n_node_types, T);
  g.BuildGlobalSampler();
  return 0;
}

// Load a directory of .dat partitions written by the reference's own
// euler/tools (Graph::Init, graph.cc:72-120, local mode = shard 0 of 1).
int euler_ref_graph_load(const char* dir) {
  euler_ref_graph_clear();
  auto s = G().Init(0, 1, "node", dir, "node");
  return s.ok() ? 0 : -1;
}

// Dense (float32) node features: written straight into the reference Node's
// own storage (float_features_ / float_features_idx_, node.h) so that the
// reference's Node::GetFloat32Feature (node.cc:330-394) serves them.
// feat_idx is [n, F] cumulative ends (row-relative), feat_ptr [n+1].
int euler_ref_set_float_features(const uint64_t* ids, int64_t n, int32_t F,
                                 const int64_t* feat_ptr, const int32_t* feat_idx,
                                 const float* feat_val) {
  for (int64_t i = 0; i < n; ++i) {
    auto it = G().node_map_.find(ids[i]);
    if (it == G().node_map_.end()) return -1;
    euler::Node* node = it->second;
    node->float_features_idx_.assign(feat_idx + i * F, feat_idx + (i + 1) * F);
    node->float_features_.assign(feat_val + feat_ptr[i], feat_val + feat_ptr[i + 1]);
  }
  return 0;
}

// What the reference holds for these nodes (loaded .dat files included):
// first call with feat_val == NULL sizes it.  Slots a node does not have
// repeat its last end (length 0), which is how GET_NODE_FEATURE treats them.
int64_t euler_ref_export_float_features(const uint64_t* ids, int64_t n, int32_t F,
                                        int64_t* feat_ptr, int32_t* feat_idx,
                                        float* feat_val) {
  int64_t off = 0;
  for (int64_t i = 0; i < n; ++i) {
    auto it = G().node_map_.find(ids[i]);
    if (feat_ptr) feat_ptr[i] = off;
    int32_t last = 0;
    for (int32_t f = 0; f < F; ++f) {
      if (it != G().node_map_.end() &&
          f < (int32_t)it->second->float_features_idx_.size())
        last = it->second->float_features_idx_[f];
      if (feat_idx) feat_idx[i * F + f] = last;
    }
    if (it != G().node_map_.end()) {
      const auto& v = it->second->float_features_;
      if (feat_val) std::copy(v.begin(), v.end(), feat_val + off);
      off += (int64_t)v.size();
    }
  }
  if (feat_ptr) feat_ptr[n] = off;
  return off;
}

int32_t euler_ref_num_float_features() {
  int32_t m = 0;
  for (auto& kv : G().node_map_)
    m = std::max(m, (int32_t)kv.second->float_features_idx_.size());
  return m;
}

// TF GetDenseFeature (tf_euler/kernels/get_dense_feature_op.cc:63-125) over
// the reference's Node::GetFloat32Feature: out is [n, dim], zero filled, row j
// receives the node's stored values of feature `fid`.
int euler_ref_get_dense_feature(const uint64_t* ids, int64_t n, int32_t fid,
                                int32_t dim, float* out) {
  std::fill(out, out + n * (int64_t)dim, 0.0f);
  std::vector<int32_t> fids(1, fid);
  for (int64_t j = 0; j < n; ++j) {
    auto it = G().node_map_.find(ids[j]);
    if (it == G().node_map_.end()) continue;
    std::vector<uint32_t> nums;
    std::vector<float> vals;
    it->second->GetFloat32Feature(fids, &nums, &vals);
    if ((int64_t)vals.size() > dim) return -2;   // the TF kernel would overrun
    std::copy(vals.begin(), vals.end(), out + j * (int64_t)dim);
  }
  return 0;
}

int64_t euler_ref_num_nodes() { return (int64_t)G().node_map_.size(); }

int32_t euler_ref_num_node_types() {
  return (int32_t)G().meta_.node_type_map_.size();
}

// node_map_ iteration order (graph.cc:349 builds the alias tables in it, Q7).
int64_t euler_ref_node_order(uint64_t* ids_out) {
  int64_t k = 0;
  for (auto& kv : G().node_map_) ids_out[k++] = kv.first;
  return k;
}

int euler_ref_node_info(const uint64_t* ids, int64_t n, int32_t* type,
                        float* weight) {
  for (int64_t i = 0; i < n; ++i) {
    auto* node = G().GetNodeByID(ids[i]);
    type[i] = node ? node->GetType() : -1;
    weight[i] = node ? node->GetWeight() : 0.f;
  }
  return 0;
}

// Number of edge-type groups of a node (-1 if unknown) and its out degree.
int euler_ref_row_shape(uint64_t id, int32_t* n_groups, int64_t* degree) {
  auto* node = G().GetNodeByID(id);
  if (!node) { *n_groups = -1; *degree = 0; return -1; }
  *n_groups = (int32_t)node->neighbor_info_.neighbor_groups_idx.size();
  *degree = (int64_t)node->neighbor_info_.neighbors.size();
  return 0;
}

// Export the reference's stored adjacency (node.h:49-57) verbatim for the
// listed ids: row_ptr[n+1], type_end[n*T] (cumulative, row-relative),
// nbr[E], prefix_w[E] (running float sums ACROSS types), type_prefix[n*T]
// (edge_group_collection running sums).  Unknown ids export an empty row.
int64_t euler_ref_export_rows(const uint64_t* ids, int64_t n, int32_t T,
                              int64_t* row_ptr, int32_t* type_end,
                              uint64_t* nbr, float* prefix_w,
                              float* type_prefix) {
  int64_t off = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (row_ptr) row_ptr[i] = off;
    auto* node = G().GetNodeByID(ids[i]);
    if (!node) {
      if (type_end) for (int t = 0; t < T; ++t) {
        type_end[i * T + t] = 0; type_prefix[i * T + t] = 0.f;
      }
      continue;
    }
    const auto& ni = node->neighbor_info_;
    if ((int32_t)ni.neighbor_groups_idx.size() != T) return -2;
    if (type_end) {
      for (int t = 0; t < T; ++t) {
        type_end[i * T + t] = ni.neighbor_groups_idx[t];
        type_prefix[i * T + t] = ni.edge_group_collection.sum_weights_[t];
        if (ni.edge_group_collection.ids_[t] != t) return -3;
      }
      memcpy(nbr + off, ni.neighbors.data(), ni.neighbors.size() * 8);
      memcpy(prefix_w + off, ni.neighbors_weight.data(),
             ni.neighbors_weight.size() * 4);
    }
    off += (int64_t)ni.neighbors.size();
  }
  if (row_ptr) row_ptr[n] = off;
  return off;
}

// API_SAMPLE_NB core result (GQL layout): idx[n,2], ids/w/t[n*count].
// Loop = api.cc:223-236 with the RNG stream set per root; empty rows filled
// with count x (0, 0.0, 0) (sample_neighbor_op.cc:134-143); flattened as
// FillNeighbor does (common.cc:275-334).  Returns total element count.
int64_t euler_ref_sample_neighbor(uint64_t seed, uint32_t call_id,
                                  const uint64_t* ids, int64_t n,
                                  const int32_t* edge_types, int32_t k,
                                  int32_t count, int32_t* idx,
                                  uint64_t* out_id, float* out_w,
                                  int32_t* out_t) {
  std::vector<int32_t> et(edge_types, edge_types + k);
  int64_t off = 0;
  for (int64_t i = 0; i < n; ++i) {
    std::vector<euler::common::IDWeightPair> res;
    auto* node = G().GetNodeByID(ids[i]);
    if (node != nullptr) {
      euler_ref_set_rng(seed, call_id, EO_DOMAIN_NEIGHBOR, ids[i]);
      res = node->SampleNeighbor(et, count);
    }
    if (res.empty()) {
      for (int32_t j = 0; j < count; ++j)
        res.push_back(euler::common::IDWeightPair(
            euler::common::DEFAULT_UINT64, 0, 0));
    }
    idx[2 * i] = (int32_t)off;
    idx[2 * i + 1] = (int32_t)(off + res.size());
    for (auto& iw : res) {
      out_id[off] = std::get<0>(iw);
      out_w[off] = std::get<1>(iw);
      out_t[off] = std::get<2>(iw);
      ++off;
    }
  }
  return off;
}

// API_SAMPLE_NODE core path without conditions: euler::SampleNode
// (api.cc:32-37 -> graph.cc:221-275).  Returns the number of ids produced
// (0 when the type weight sum is 0; the op then logs and emits nothing,
// sample_node_op.cc:118-122).
int64_t euler_ref_sample_node(uint64_t seed, uint32_t call_id,
                              const int32_t* node_types, int32_t k,
                              int32_t count, uint64_t* out) {
  std::vector<int> types(node_types, node_types + k);
  euler_ref_set_rng(seed, call_id, EO_DOMAIN_NODE, 0);
  auto vec = euler::SampleNode(types, count);
  for (size_t i = 0; i < vec.size(); ++i) out[i] = vec[i];
  return (int64_t)vec.size();
}

int64_t euler_ref_alias_size(int32_t type) {
  auto& g = G();
  if (type < 0) return (int64_t)g.node_type_collection_.ids_.size();
  if (type >= (int32_t)g.node_samplers_.size()) return -1;
  return (int64_t)g.node_samplers_[type].ids_.size();
}

// Export an alias table built by the reference (type < 0: the node-type
// collection; ids are then the type ids).
int euler_ref_alias_table(int32_t type, uint64_t* ids, float* weights,
                          float* prob, int64_t* alias, float* sum_weight) {
  auto& g = G();
  if (type < 0) {
    auto& c = g.node_type_collection_;
    for (size_t i = 0; i < c.ids_.size(); ++i) {
      ids[i] = (uint64_t)c.ids_[i];
      weights[i] = c.weights_[i];
      prob[i] = c.alias_.prob_[i];
      alias[i] = c.alias_.alias_[i];
    }
    *sum_weight = c.sum_weight_;
    return 0;
  }
  auto& c = g.node_samplers_[type];
  for (size_t i = 0; i < c.ids_.size(); ++i) {
    ids[i] = c.ids_[i];
    weights[i] = c.weights_[i];
    prob[i] = c.alias_.prob_[i];
    alias[i] = c.alias_.alias_[i];
  }
  *sum_weight = c.sum_weight_;
  return 0;
}

// euler::GetFullNeighbor (api.cc:208-221) flattened like FillNeighbor.
// Pass out_id == NULL to size the result.
int64_t euler_ref_get_full_neighbor(const uint64_t* ids, int64_t n,
                                    const int32_t* edge_types, int32_t k,
                                    int32_t* idx, uint64_t* out_id,
                                    float* out_w, int32_t* out_t) {
  std::vector<uint64_t> v(ids, ids + n);
  std::vector<int> et(edge_types, edge_types + k);
  auto res = euler::GetFullNeighbor(v, et);
  int64_t off = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (idx) { idx[2 * i] = (int32_t)off; }
    for (auto& iw : res[i]) {
      if (out_id) {
        out_id[off] = std::get<0>(iw);
        out_w[off] = std::get<1>(iw);
        out_t[off] = std::get<2>(iw);
      }
      ++off;
    }
    if (idx) { idx[2 * i + 1] = (int32_t)off; }
  }
  return off;
}

// API_GET_NB_NODE with its post-process (core/kernels/get_neighbor_op.cc:
// 117-168; the op itself needs protobuf, so the loop is restated around the
// reference's GetFullNeighbor with the reference's comparators and std::sort).
int64_t euler_ref_get_neighbor(const uint64_t* ids, int64_t n,
                               const int32_t* edge_types, int32_t k,
                               int32_t order_by, int32_t desc, int64_t limit,
                               int32_t* idx, uint64_t* out_id, float* out_w,
                               int32_t* out_t) {
  typedef euler::common::IDWeightPair IdWeightPair;
  std::vector<uint64_t> v(ids, ids + n);
  std::vector<int> et(edge_types, edge_types + k);
  auto res = euler::GetFullNeighbor(v, et);
  const int sign = desc ? -1 : 1;
  if (order_by == 1) {
    auto cmp = [sign] (IdWeightPair a, IdWeightPair b) {
      return (std::get<0>(a) <= std::get<0>(b) ? 1 : -1) * sign > 0;
    };
    for (auto& item : res) std::sort(item.begin(), item.end(), cmp);
  } else if (order_by == 2) {
    auto cmp = [sign] (IdWeightPair a, IdWeightPair b) {
      return (std::get<1>(a) <= std::get<1>(b) ? 1 : -1) * sign > 0;
    };
    for (auto& item : res) std::sort(item.begin(), item.end(), cmp);
  }
  if (limit >= 0)
    for (auto& item : res) if ((int64_t)item.size() > limit) item.resize(limit);
  int64_t off = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (idx) idx[2 * i] = (int32_t)off;
    for (auto& iw : res[i]) {
      if (out_id) {
        out_id[off] = std::get<0>(iw);
        out_w[off] = std::get<1>(iw);
        out_t[off] = std::get<2>(iw);
      }
      ++off;
    }
    if (idx) idx[2 * i + 1] = (int32_t)off;
  }
  return off;
}

// TF RandomWalk (tf_euler/kernels/random_walk_op.cc).  edge_types is
// [walk_len, k].  |p-1|,|q-1| <= 1e-6 -> TraditionalRandomWalk (:207-247): a
// chain of count=1 sampleNB hops on the CORE id tensor (sentinel 0 rows walk
// on from node id 0), output 0 -> default_node.  Otherwise node2vec
// (RWCallback :83-168) with the reference CompactWeightedCollection.
int euler_ref_random_walk(uint64_t seed, uint32_t call_id,
                          const int64_t* nodes, int64_t n,
                          const int32_t* edge_types, int32_t k,
                          int32_t walk_len, float p, float q,
                          int64_t default_node, int64_t* out) {
  const int64_t L = walk_len + 1;
  for (int64_t i = 0; i < n; ++i) out[i * L] = nodes[i];
  const float kEps = 1.0e-6;
  if (fabs(p - 1.0) <= kEps && fabs(q - 1.0) <= kEps) {
    for (int64_t i = 0; i < n; ++i) {
      uint64_t cur = (uint64_t)nodes[i];
      for (int32_t s = 0; s < walk_len; ++s) {
        std::vector<int32_t> et(edge_types + s * k, edge_types + (s + 1) * k);
        std::vector<euler::common::IDWeightPair> res;
        auto* node = G().GetNodeByID(cur);
        if (node != nullptr) {
          euler_ref_set_rng(seed, call_id + s, EO_DOMAIN_NEIGHBOR, cur);
          res = node->SampleNeighbor(et, 1);
        }
        uint64_t nb = res.empty() ? euler::common::DEFAULT_UINT64
                                  : std::get<0>(res[0]);
        out[i * L + s + 1] = nb == euler::common::DEFAULT_UINT64
                                 ? (int64_t)(uint64_t)default_node
                                 : (int64_t)nb;
        cur = nb;
      }
    }
    return 0;
  }
  std::vector<std::vector<int64_t>> parent_neighbors(n);
  std::vector<int64_t> parent_ids(nodes, nodes + n);
  std::vector<int64_t> cur(nodes, nodes + n);
  for (int32_t s = 0; s < walk_len; ++s) {
    std::vector<int> et(edge_types + s * k, edge_types + (s + 1) * k);
    std::vector<uint64_t> cur_u(cur.begin(), cur.end());
    auto full = euler::GetFullNeighbor(cur_u, et);
    std::vector<std::vector<int64_t>> neighbors(n);
    std::vector<int64_t> next(n);
    for (int64_t i = 0; i < n; ++i) {
      std::vector<float> w;
      for (auto& iw : full[i]) {
        neighbors[i].push_back((int64_t)std::get<0>(iw));
        w.push_back(std::get<1>(iw));
      }
      auto& cn = neighbors[i];
      int64_t sample_id = default_node;
      if (!cn.empty()) {
        auto parent_id = parent_ids[i];
        auto& pn = parent_neighbors[i];
        // BuildWeights (:140-168)
        size_t j = 0, kk = 0;
        while (j < cn.size() && kk < pn.size()) {
          if (cn[j] < pn[kk]) {
            if (cn[j] != parent_id) w[j] /= q; else w[j] /= p;
            ++j;
          } else if (cn[j] == pn[kk]) {
            ++kk; ++j;
          } else {
            ++kk;
          }
        }
        while (j < cn.size()) {
          if (cn[j] != parent_id) w[j] /= q; else w[j] /= p;
          ++j;
        }
        euler::common::CompactWeightedCollection<int64_t> sampler;
        sampler.Init(cn, w);
        euler_ref_set_rng(seed, call_id + s, EO_DOMAIN_WALK, (uint64_t)i);
        sample_id = sampler.Sample().first;
      }
      out[i * L + s + 1] = sample_id;
      next[i] = sample_id;
    }
    parent_neighbors = neighbors;
    parent_ids = cur;
    cur = next;
  }
  return 0;
}

// CPU baseline timing: `iters` minibatches of 2-hop (or L-hop) fanout through
// the reference Node::SampleNeighbor, batch loop as api.cc:223-236, split over
// `threads` query threads (the reference client pool has 8,
// client/query_proxy.cc:205-210).  Roots of hop h+1 are the ids sampled at hop
// h.  Returns seconds; *edges receives the number of sampled edges.
double euler_ref_bench_fanout(uint64_t seed, const uint64_t* roots,
                              int64_t batch, int32_t iters,
                              const int32_t* counts, int32_t hops,
                              int32_t threads, int64_t* edges) {
  std::vector<int64_t> per_thread(threads, 0);
  auto t0 = std::chrono::steady_clock::now();
  auto work = [&](int tid) {
    std::vector<int32_t> et(1, 0);
    int64_t produced = 0;
    for (int32_t it = tid; it < iters; it += threads) {
      std::vector<uint64_t> frontier(roots + (int64_t)it * batch,
                                     roots + (int64_t)(it + 1) * batch);
      for (int32_t h = 0; h < hops; ++h) {
        std::vector<uint64_t> next;
        next.reserve(frontier.size() * counts[h]);
        for (size_t i = 0; i < frontier.size(); ++i) {
          auto* node = G().GetNodeByID(frontier[i]);
          std::vector<euler::common::IDWeightPair> res;
          if (node != nullptr) {
            euler_ref_set_rng(seed, (uint32_t)(it * hops + h),
                              EO_DOMAIN_NEIGHBOR, frontier[i]);
            res = node->SampleNeighbor(et, counts[h]);
          }
          if (res.empty()) {
            next.insert(next.end(), counts[h], 0);
          } else {
            for (auto& iw : res) next.push_back(std::get<0>(iw));
          }
        }
        produced += (int64_t)next.size();
        frontier.swap(next);
      }
    }
    per_thread[tid] = produced;
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) pool.emplace_back(work, t);
  for (auto& th : pool) th.join();
  auto t1 = std::chrono::steady_clock::now();
  int64_t total = 0;
  for (auto v : per_thread) total += v;
  *edges = total;
  return std::chrono::duration<double>(t1 - t0).count();
}

// CPU baseline on SURVEY 8(d)'s protocol.  One QUERY = the DAG the reference's
// optimizer builds for `v(roots).sampleNB(...).sampleNB(...)`: per hop
// ID_UNIQUE -> API_SAMPLE_NB over the distinct ids -> IDX_GATHER / DATA_GATHER
// (parser/compiler.cc:76-90).  The pieces follow
//   ID_UNIQUE     core/kernels/id_unique_op.cc:35-64 (unordered_map, first
//                 occurrence order, gather_idx per position),
//   API_SAMPLE_NB core/api/api.cc:223-236 batch loop over Node::SampleNeighbor,
//                 empty rows -> count x (0, 0.0f, 0) (sample_neighbor_op.cc:134-143),
//                 FillNeighbor's flat id / weight / type arrays (common.cc:275-334),
//   DATA_GATHER   core/kernels/data_gather_op.cc:33-80 (row copies through
//                 gather_idx; rows are `count` wide, so IDX_GATHER is arithmetic).
// mode 0 "as shipped" (USE_OPENMP off, CMakeLists.txt:15): `threads` query
//   threads - the client pool has 8, client/query_proxy.cc:205-210 - each runs
//   one whole query per round; a round ends when all have finished.
// mode 1 "best case" (-DOPENMP): one query at a time, the api.cc batch loop is
//   an `omp parallel for` over `threads` threads; ID_UNIQUE / gathers stay
//   serial, as they are in the reference.
// rounds = warmup + timed; secs_out[timed] receives each timed round's wall
// time.  Query q of a round reads roots[(q % n_batches) * batch ...].  Returns
// the sampled edges (sum of rows x count over hops) ONE round produces.
namespace {
struct HopResult {
  std::vector<uint64_t> ids;
  std::vector<float> w;
  std::vector<int32_t> t;
};

void SampleRows(uint64_t seed, uint32_t call_id, const std::vector<uint64_t>& roots,
                const std::vector<int32_t>& et, int32_t count, bool omp,
                int32_t threads, HopResult* out) {
  const int64_t n = (int64_t)roots.size();
  out->ids.resize(n * count);
  out->w.resize(n * count);
  out->t.resize(n * count);
  auto body = [&](int64_t i) {
    auto* node = G().GetNodeByID(roots[i]);
    std::vector<euler::common::IDWeightPair> res;
    if (node != nullptr) {
      euler_ref_set_rng(seed, call_id, EO_DOMAIN_NEIGHBOR, roots[i]);
      res = node->SampleNeighbor(et, count);
    }
    uint64_t* oi = out->ids.data() + i * count;
    float* ow = out->w.data() + i * count;
    int32_t* ot = out->t.data() + i * count;
    if (res.empty()) {
      for (int32_t j = 0; j < count; ++j) { oi[j] = 0; ow[j] = 0.f; ot[j] = 0; }
    } else {
      for (int32_t j = 0; j < count; ++j) {
        oi[j] = std::get<0>(res[j]); ow[j] = std::get<1>(res[j]); ot[j] = std::get<2>(res[j]);
      }
    }
  };
  if (omp) {
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t i = 0; i < n; ++i) body(i);
  } else {
    for (int64_t i = 0; i < n; ++i) body(i);
  }
}

int64_t RunDagQuery(uint64_t seed, uint32_t call_base, const uint64_t* roots, int64_t batch,
                    const int32_t* counts, int32_t hops, bool dedup, bool omp,
                    int32_t threads, uint64_t* checksum) {
  std::vector<int32_t> et(1, 0);
  std::vector<uint64_t> frontier(roots, roots + batch);
  int64_t produced = 0;
  uint64_t sum = 0;
  for (int32_t h = 0; h < hops; ++h) {
    const int32_t count = counts[h];
    HopResult full;
    if (dedup) {
      // ID_UNIQUE
      std::unordered_map<uint64_t, int32_t> ids_map;
      ids_map.reserve(frontier.size());
      std::vector<uint64_t> unique_vec;
      unique_vec.reserve(frontier.size());
      std::vector<int32_t> gather_idx(frontier.size());
      int32_t cnt = 0;
      for (size_t i = 0; i < frontier.size(); ++i) {
        auto it = ids_map.find(frontier[i]);
        if (it == ids_map.end()) {
          ids_map[frontier[i]] = cnt++;
          unique_vec.push_back(frontier[i]);
        }
      }
      for (size_t i = 0; i < frontier.size(); ++i) gather_idx[i] = ids_map.at(frontier[i]);
      HopResult uq;
      SampleRows(seed, call_base + h, unique_vec, et, count, omp, threads, &uq);
      // DATA_GATHER x3 (ids, weights, types)
      const size_t n = frontier.size();
      full.ids.resize(n * count); full.w.resize(n * count); full.t.resize(n * count);
      for (size_t i = 0; i < n; ++i) {
        const size_t b = (size_t)gather_idx[i] * count;
        std::copy(uq.ids.begin() + b, uq.ids.begin() + b + count, full.ids.begin() + i * count);
      }
      for (size_t i = 0; i < n; ++i) {
        const size_t b = (size_t)gather_idx[i] * count;
        std::copy(uq.w.begin() + b, uq.w.begin() + b + count, full.w.begin() + i * count);
      }
      for (size_t i = 0; i < n; ++i) {
        const size_t b = (size_t)gather_idx[i] * count;
        std::copy(uq.t.begin() + b, uq.t.begin() + b + count, full.t.begin() + i * count);
      }
    } else {
      SampleRows(seed, call_base + h, frontier, et, count, omp, threads, &full);
    }
    produced += (int64_t)full.ids.size();
    for (size_t i = 0; i < full.ids.size(); i += 997) sum += full.ids[i];
    frontier.swap(full.ids);
  }
  if (checksum) *checksum += sum;
  return produced;
}
}  // namespace

int64_t euler_ref_bench_fanout_dag(uint64_t seed, const uint64_t* roots, int64_t batch,
                                   int64_t n_batches, const int32_t* counts, int32_t hops,
                                   int32_t threads, int32_t mode, int32_t dedup,
                                   int32_t warmup, int32_t timed, double* secs_out) {
  if (threads < 1) threads = 1;
  const int32_t rounds = warmup + timed;
  const int32_t queries_per_round = mode == 0 ? threads : 1;
  int64_t per_round = 0;
  uint64_t sink = 0;
  if (mode == 1) {
    for (int32_t r = 0; r < rounds; ++r) {
      auto t0 = std::chrono::steady_clock::now();
      int64_t e = RunDagQuery(seed, (uint32_t)(r * hops), roots + (r % n_batches) * batch, batch,
                              counts, hops, dedup != 0, true, threads, &sink);
      auto t1 = std::chrono::steady_clock::now();
      per_round = e;
      if (r >= warmup) secs_out[r - warmup] = std::chrono::duration<double>(t1 - t0).count();
    }
    return per_round + (sink == 0x7fffffffffffffffULL ? 1 : 0);
  }
  // mode 0: persistent query threads, two barriers per round
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0, generation = 0;
  auto barrier = [&]() {
    std::unique_lock<std::mutex> lk(mu);
    int gen = generation;
    if (++arrived == threads + 1) { arrived = 0; ++generation; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != generation; });
  };
  std::vector<int64_t> produced(threads, 0);
  std::vector<uint64_t> sums(threads, 0);
  auto work = [&](int tid) {
    for (int32_t r = 0; r < rounds; ++r) {
      barrier();
      int64_t q = (int64_t)r * queries_per_round + tid;
      produced[tid] = RunDagQuery(seed, (uint32_t)(q * hops), roots + (q % n_batches) * batch,
                                  batch, counts, hops, dedup != 0, false, 1, &sums[tid]);
      barrier();
    }
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) pool.emplace_back(work, t);
  for (int32_t r = 0; r < rounds; ++r) {
    barrier();
    auto t0 = std::chrono::steady_clock::now();
    barrier();
    auto t1 = std::chrono::steady_clock::now();
    if (r >= warmup) secs_out[r - warmup] = std::chrono::duration<double>(t1 - t0).count();
  }
  for (auto& th : pool) th.join();
  for (int t = 0; t < threads; ++t) { per_round += produced[t]; sink += sums[t]; }
  return per_round + (sink == 0x7fffffffffffffffULL ? 1 : 0);
}

// The same graph build as euler_ref_graph_build with the Node objects
// constructed by `threads` threads (Node::Init is per node); AddNode and
// BuildGlobalSampler stay serial.  For the 20M-node baseline graph.
int euler_ref_graph_build_mt(int64_t n, const uint64_t* ids, const int32_t* node_type,
                             const float* node_weight, int32_t T, int32_t n_node_types,
                             const int64_t* seg_ptr, const uint64_t* nbr, const float* w,
                             int32_t threads, int32_t build_sampler) {
  euler_ref_graph_clear();
  auto& g = G();
  g.reserveNodeMap(n);
  std::vector<euler::Node*> nodes(n, nullptr);
  std::atomic<int> bad(0);
  auto work = [&](int tid) {
    std::vector<std::vector<uint64_t>> nb(T);
    std::vector<std::vector<float>> nw(T);
    std::vector<std::vector<uint64_t>> f_u64;
    std::vector<std::vector<float>> f_f32;
    std::vector<std::string> f_bin;
    int64_t b0 = n * tid / threads, e0 = n * (tid + 1) / threads;
    for (int64_t i = b0; i < e0; ++i) {
      for (int t = 0; t < T; ++t) {
        int64_t b = seg_ptr[i * T + t], e = seg_ptr[i * T + t + 1];
        nb[t].assign(nbr + b, nbr + e);
        nw[t].assign(w + b, w + e);
      }
      auto* node = new euler::Node(ids[i], node_weight[i], node_type[i]);
      if (!node->Init(nb, nw, f_u64, f_f32, f_bin)) bad = 1;
      nodes[i] = node;
    }
  };
  if (threads < 1) threads = 1;
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) pool.emplace_back(work, t);
  for (auto& th : pool) th.join();
  if (bad) return -1;
  for (int64_t i = 0; i < n; ++i) g.AddNode(nodes[i]);
  SetTypeMaps(n_node_types, T);
  if (build_sampler) g.BuildGlobalSampler();
  return 0;
}

// ---------------------------------------------------------------- layerwise
// Load Node/ AND Edge/ partitions with the reference's own loader
// (data type "all"): EdgeExist then consults the Edge records.
int euler_ref_graph_load_all(const char* dir) {
  euler_ref_graph_clear();
  auto s = G().Init(0, 1, "node", dir, "all");
  return s.ok() ? 0 : -1;
}

// For graphs built from raw adjacency: one Edge record per (src, dst, type)
// entry of the node rows - what euler/tools writes for consistent input.
int64_t euler_ref_add_edges_from_adjacency() {
  auto& g = G();
  int64_t added = 0;
  for (auto& kv : g.node_map_) {
    euler::Node* node = kv.second;
    auto& ni = node->neighbor_info_;
    const int32_t T = (int32_t)ni.neighbor_groups_idx.size();
    for (int32_t t = 0; t < T; ++t) {
      int32_t b = t == 0 ? 0 : ni.neighbor_groups_idx[t - 1];
      for (int32_t j = b; j < ni.neighbor_groups_idx[t]; ++j) {
        float pre = j == 0 ? 0 : ni.neighbors_weight[j - 1];
        euler::common::EdgeID eid(kv.first, ni.neighbors[j], t);
        if (g.edge_map_.find(eid) != g.edge_map_.end()) continue;
        g.AddEdge(new euler::Edge(kv.first, ni.neighbors[j], t,
                                  ni.neighbors_weight[j] - pre));
        ++added;
      }
    }
  }
  return added;
}

int64_t euler_ref_num_edges() { return (int64_t)G().edge_map_.size(); }

int euler_ref_edge_exist(uint64_t src, uint64_t dst, int32_t type) {
  return euler::EdgeExist(euler::EdgeId(src, dst, type)) ? 1 : 0;
}

// API_GET_EDGE_SUM_WEIGHT body (get_edge_sum_weight_op.cc:51-62).
void euler_ref_get_edge_sum_weight(const uint64_t* ids, int64_t n,
                                   const int32_t* edge_types, int32_t k,
                                   float* out_w) {
  std::vector<int32_t> et(edge_types, edge_types + k);
  for (int64_t i = 0; i < n; ++i) {
    std::vector<uint64_t> roots = {ids[i]};
    euler::IdWeightPairVec nb = euler::GetFullNeighbor(roots, et);
    float sum_weight = 0;
    for (auto& iw : nb[0]) sum_weight += std::get<1>(iw);
    out_w[i] = sum_weight;
  }
}

// API_SAMPLE_ROOT body (sample_root_op.cc:48-86) over the reference's
// FastWeightedCollection; RNG stream = batch row.
void euler_ref_sample_root(uint64_t seed, uint32_t call_id, const uint64_t* roots_in,
                           const float* weights_in, int64_t batch, int32_t n,
                           int32_t m, int64_t default_node_in, uint64_t* out) {
  uint64_t default_node = (uint64_t)default_node_in;
  for (int64_t i = 0; i < batch; ++i) {
    std::vector<uint64_t> roots(roots_in + i * n, roots_in + (i + 1) * n);
    std::vector<float> weights(weights_in + i * n, weights_in + (i + 1) * n);
    euler::common::FastWeightedCollection<uint64_t> fwc;
    fwc.Init(roots, weights);
    std::vector<uint64_t> result(m);
    if (fwc.GetSumWeight() == 0) {
      for (int32_t j = 0; j < m; ++j) result[j] = default_node;
    } else {
      euler_ref_set_rng(seed, call_id, EO_DOMAIN_ROOT, (uint64_t)i);
      for (int32_t j = 0; j < m; ++j) result[j] = fwc.Sample().first;
    }
    std::copy(result.begin(), result.end(), out + i * m);
  }
}

// API_SAMPLE_L body (sample_layer_op.cc:54-70); RNG stream = position.
void euler_ref_sample_layer(uint64_t seed, uint32_t call_id, const uint64_t* l_root,
                            int64_t n, const int32_t* edge_types, int32_t k,
                            int64_t default_node, uint64_t* out_id, float* out_w,
                            int32_t* out_t) {
  std::vector<int32_t> et(edge_types, edge_types + k);
  for (int64_t i = 0; i < n; ++i) {
    std::vector<uint64_t> roots = {l_root[i]};
    euler_ref_set_rng(seed, call_id, EO_DOMAIN_LAYER, (uint64_t)i);
    euler::IdWeightPairVec nb = euler::SampleNeighbor(roots, et, 1);
    if (nb[0].empty()) {
      out_id[i] = (uint64_t)default_node; out_w[i] = 0; out_t[i] = 0;
    } else {
      out_id[i] = std::get<0>(nb[0][0]);
      out_w[i] = std::get<1>(nb[0][0]);
      out_t[i] = std::get<2>(nb[0][0]);
    }
  }
}

// API_SPARSE_GEN_ADJ + API_SPARSE_GET_ADJ (sparse_gen_adj_op.cc:52-61,
// sparse_get_adj_op.cc:55-91) over the reference's own EdgeExist: source r belongs
// to batch row r / n and keeps, in candidate order, the candidates of that row it
// has an edge of a listed type to.  out_id == NULL sizes the result.
int64_t euler_ref_sparse_get_adj(const uint64_t* roots, const uint64_t* l_nb,
                                 int64_t batch, int32_t n, int32_t m,
                                 const int32_t* edge_types, int32_t k,
                                 int32_t* idx, uint64_t* out_id) {
  int64_t written = 0;
  for (int64_t r = 0; r < batch * n; ++r) {
    const uint64_t* cand = l_nb + (r / n) * m;
    if (idx) idx[2 * r] = (int32_t)written;
    for (int32_t c = 0; c < m; ++c) {
      bool linked = false;
      for (int32_t x = 0; x < k; ++x)
        linked = linked || euler::EdgeExist(euler::EdgeId(roots[r], cand[c], edge_types[x]));
      if (!linked) continue;
      if (out_id) out_id[written] = cand[c];
      ++written;
    }
    if (idx) idx[2 * r + 1] = (int32_t)written;
  }
  return written;
}

// The sparse assembly of the TF kernels (tf_euler/kernels/sparse_get_adj_op.cc:
// 92-124 = sample_neighbor_layerwise_with_adj_op.cc:112-140): per batch row a
// std::set of (source id, neighbour id) pairs taken from the core result, a 1 for
// every (j, c) whose pair is in the set, a 0 at the last (j, c) otherwise; the
// shape follows SparseTensorBuilder (tf_euler/utils/sparse_tensor_builder.h:30-40:
// largest index + 1 per dimension).  indices == NULL sizes the result.
int64_t euler_ref_adj_to_sparse(const uint64_t* nodes, const uint64_t* nb_nodes,
                                int64_t batch_size, int32_t N, int32_t M,
                                const int32_t* idx_data, const uint64_t* val_data,
                                int64_t* indices, int64_t* values, int64_t* shape) {
  int64_t nnz = 0, extent[3] = {0, 0, 0};
  for (int64_t b = 0; b < batch_size; ++b) {
    std::set<std::pair<int64_t, int64_t>> pairs;
    for (int64_t j = 0; j < N; ++j) {
      const int64_t r = b * N + j;
      for (int32_t p = idx_data[2 * r]; p < idx_data[2 * r + 1]; ++p)
        pairs.insert({(int64_t)nodes[r], (int64_t)val_data[p]});
    }
    for (int64_t j = 0; j < N; ++j) {
      for (int64_t c = 0; c < M; ++c) {
        const bool one = pairs.count({(int64_t)nodes[b * N + j], (int64_t)nb_nodes[b * M + c]}) > 0;
        if (!one && !(j == N - 1 && c == M - 1)) continue;
        const int64_t at[3] = {b, j, c};
        for (int d = 0; d < 3; ++d) {
          if (indices) indices[3 * nnz + d] = at[d];
          if (at[d] + 1 > extent[d]) extent[d] = at[d] + 1;
        }
        if (values) values[nnz] = one ? 1 : 0;
        ++nnz;
      }
    }
  }
  if (shape) { shape[0] = extent[0]; shape[1] = extent[1]; shape[2] = extent[2]; }
  return nnz;
}

// API_SAMPLE_N_WITH_TYPES body (sample_n_with_types_op.cc:44-52) with the TF
// kernel's inputs (one count for all types); RNG stream = index of the call.
// Returns 0, or -1 where the TF kernel aborts on a size mismatch.
int euler_ref_sample_n_with_types(uint64_t seed, uint32_t call_id, const int32_t* types,
                                  int64_t n, int32_t count, uint64_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    // Graph::SampleNode indexes node_samplers_[type] unchecked (graph.cc:238)
    if (types[i] != -1 &&
        (types[i] < 0 || types[i] >= (int32_t)G().node_samplers_.size()))
      return -1;
    euler_ref_set_rng(seed, call_id, EO_DOMAIN_NODE, (uint64_t)i);
    auto vec = euler::SampleNode({types[i]}, count);
    if ((int32_t)vec.size() != count) return -1;
    std::copy(vec.begin(), vec.end(), out + i * count);
  }
  return 0;
}

// euler::GetNodeType (api.cc:50-61).
void euler_ref_get_node_type(const uint64_t* ids, int64_t n, int32_t* out) {
  std::vector<uint64_t> v(ids, ids + n);
  auto t = euler::GetNodeType(v);
  std::copy(t.begin(), t.end(), out);
}

// uint64 ("sparse") node features, the counterparts of the float helpers above.
int euler_ref_set_u64_features(const uint64_t* ids, int64_t n, int32_t U,
                               const int64_t* feat_ptr, const int32_t* feat_idx,
                               const uint64_t* feat_val) {
  for (int64_t i = 0; i < n; ++i) {
    auto it = G().node_map_.find(ids[i]);
    if (it == G().node_map_.end()) return -1;
    euler::Node* node = it->second;
    node->uint64_features_idx_.assign(feat_idx + i * U, feat_idx + (i + 1) * U);
    node->uint64_features_.assign(feat_val + feat_ptr[i], feat_val + feat_ptr[i + 1]);
  }
  return 0;
}

int64_t euler_ref_export_u64_features(const uint64_t* ids, int64_t n, int32_t U,
                                      int64_t* feat_ptr, int32_t* feat_idx,
                                      uint64_t* feat_val) {
  int64_t off = 0;
  for (int64_t i = 0; i < n; ++i) {
    auto it = G().node_map_.find(ids[i]);
    if (feat_ptr) feat_ptr[i] = off;
    int32_t last = 0;
    for (int32_t f = 0; f < U; ++f) {
      if (it != G().node_map_.end() &&
          f < (int32_t)it->second->uint64_features_idx_.size())
        last = it->second->uint64_features_idx_[f];
      if (feat_idx) feat_idx[i * U + f] = last;
    }
    if (it != G().node_map_.end()) {
      const auto& v = it->second->uint64_features_;
      if (feat_val) std::copy(v.begin(), v.end(), feat_val + off);
      off += (int64_t)v.size();
    }
  }
  if (feat_ptr) feat_ptr[n] = off;
  return off;
}

int32_t euler_ref_num_u64_features() {
  int32_t m = 0;
  for (auto& kv : G().node_map_)
    m = std::max(m, (int32_t)kv.second->uint64_features_idx_.size());
  return m;
}

// TF GetSparseFeature for one feature (get_sparse_feature_op.cc:84-113) over the
// reference's Node::GetUint64Feature + the SparseTensorBuilder rules.
// indices == NULL sizes the result.
int64_t euler_ref_get_sparse_feature(const uint64_t* ids, int64_t n, int32_t fid,
                                     int64_t default_value, int64_t* indices,
                                     int64_t* values, int64_t* shape) {
  std::vector<int32_t> fids(1, fid);
  int64_t nnz = 0;
  int64_t dense_shape[2] = {0, 0};
  auto emplace = [&](int64_t a, int64_t b, int64_t v) {
    if (indices) { indices[2 * nnz] = a; indices[2 * nnz + 1] = b; values[nnz] = v; }
    if (a + 1 > dense_shape[0]) dense_shape[0] = a + 1;
    if (b + 1 > dense_shape[1]) dense_shape[1] = b + 1;
    ++nnz;
  };
  for (int64_t j = 0; j < n; ++j) {
    std::vector<uint32_t> nums;
    std::vector<uint64_t> vals;
    auto it = G().node_map_.find(ids[j]);
    if (it != G().node_map_.end()) it->second->GetUint64Feature(fids, &nums, &vals);
    if (vals.empty()) {
      emplace(j, 0, default_value);
    } else {
      for (size_t k = 0; k < vals.size(); ++k) emplace(j, (int64_t)k, (int64_t)vals[k]);
    }
  }
  if (shape) { shape[0] = dense_shape[0]; shape[1] = dense_shape[1]; }
  return nnz;
}

// API_LOCAL_SAMPLE_L (local_sample_layer_op.cc:43-146) expressed over the SAME
// library types the op uses - std::unordered_map<std::string, ...> keyed by
// to_string(id) + to_string(type) for the distinct candidates of a batch row, the
// reference's CompactWeightedCollection for the draws, memset for empty rows - one
// batch row at a time; RNG stream = batch row.
void euler_ref_local_sample_layer(uint64_t seed, uint32_t call_id, const int32_t* idx_data,
                                  int64_t idx_elems, const uint64_t* nb_id,
                                  const float* nb_w, const int32_t* nb_type, int32_t n,
                                  int32_t m, const char* weight_func_c,
                                  int64_t default_node, uint64_t* o_nb, float* o_w,
                                  int32_t* o_t) {
  struct Cand { uint64_t id; float w; int32_t type; };
  const bool root = std::string(weight_func_c) == "sqrt";
  const int32_t rows = (int32_t)(idx_elems / (n * 2));
  for (int32_t b = 0; b < rows; ++b) {
    // the row's slice of the neighbour arrays: from its first node's begin to the
    // next row's (the last row ends at the very last offset)
    const int32_t lo = idx_data[(int64_t)b * n * 2];
    const int32_t hi = b + 1 < rows ? idx_data[(int64_t)(b + 1) * n * 2] : idx_data[idx_elems - 1];
    std::unordered_map<std::string, Cand> seen;
    for (int32_t j = lo; j < hi; ++j) {
      const std::string key = std::to_string(nb_id[j]) + std::to_string(nb_type[j]);
      auto hit = seen.find(key);
      if (hit == seen.end()) seen[key] = Cand{nb_id[j], nb_w[j], nb_type[j]};
      else seen[key].w += nb_w[j];
    }
    std::vector<Cand> cands;
    std::vector<float> weights;
    for (auto& kv : seen) {                       // the container's iteration order
      if (root) kv.second.w = sqrt(kv.second.w);
      cands.push_back(kv.second);
      weights.push_back(kv.second.w);
    }
    euler::common::CompactWeightedCollection<Cand> sampler;
    if (!cands.empty()) sampler.Init(cands, weights);
    uint64_t* ids_out = o_nb + (int64_t)b * m;
    float* w_out = o_w + (int64_t)b * m;
    int32_t* t_out = o_t + (int64_t)b * m;
    if (sampler.GetSize() == 0 || sampler.GetSumWeight() == 0) {
      memset(ids_out, default_node, sizeof(uint64_t) * m);    // byte fill, as the op does
      memset(w_out, 0, sizeof(float) * m);
      memset(t_out, 0, sizeof(int32_t) * m);
      continue;
    }
    euler_ref_set_rng(seed, call_id, EO_DOMAIN_LOCAL_LAYER, (uint64_t)b);
    for (int32_t j = 0; j < m; ++j) {
      const Cand c = sampler.Sample().first;
      ids_out[j] = c.id; w_out[j] = c.w; t_out[j] = c.type;
    }
  }
}

// Iteration order of a REAL std::unordered_map<std::string, int> after inserting
// the given keys (concatenated bytes, lengths) in order; duplicate keys are
// skipped like operator[] on an existing key.  order[k] = index (among the
// distinct keys, in first-occurrence order) of the k-th element.  Returns the
// number of distinct keys.  Also returns std::hash<std::string> of every key.
int64_t euler_ref_umap_order(const char* bytes, const int32_t* lens, int64_t n,
                             int64_t* order, uint64_t* hashes) {
  std::unordered_map<std::string, int64_t> m;
  int64_t off = 0, distinct = 0;
  for (int64_t i = 0; i < n; ++i) {
    std::string key(bytes + off, (size_t)lens[i]);
    off += lens[i];
    if (hashes) hashes[i] = std::hash<std::string>()(key);
    if (m.find(key) == m.end()) m[key] = distinct++;
  }
  int64_t k = 0;
  for (auto it = m.begin(); it != m.end(); ++it) order[k++] = it->second;
  return distinct;
}

}  // extern "C"
