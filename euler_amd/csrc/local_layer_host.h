// Host half of API_LOCAL_SAMPLE_L (core/kernels/local_sample_layer_op.cc:57-121):
// the candidate tables of every batch row.  The reference collects the distinct
// (neighbour id, edge type) pairs of a row in a
// std::unordered_map<std::string, ...> keyed by to_string(id) + to_string(type)
// and hands them to its sampler in that container's ITERATION ORDER - a property
// of libstdc++'s hash, bucket policy and list insertion, not of the data.  This
// library links the same libstdc++, so the tables are built here with the same
// container, on the host; the draws over them run on the device
// (LocalSampleLayerKernel).  Pure C++ (no HIP): tests/csrc/host_check.hip runs it
// on the CPU against the reference's golden vectors.
#pragma once

#include <math.h>
#include <stdint.h>

#include <string>
#include <unordered_map>
#include <vector>

namespace euler_gpu {

struct LocalLayerTables {
  std::vector<int64_t> seg;      // [batch + 1] offsets of the rows' entries
  std::vector<uint64_t> u_id;
  std::vector<float> u_w;        // accumulated (and transformed) weight
  std::vector<int32_t> u_t;
  std::vector<float> sum_w;      // CompactWeightedCollection running f32 sums per row
};

// idx [batch * n, 2] / ids / w / t [total]: the API_GET_NB_NODE result.  false
// when idx does not index the value arrays.
inline bool BuildLocalLayerTables(const int32_t* idx, const uint64_t* ids, const float* w,
                                  const int32_t* t, int64_t total, int64_t batch, int32_t n,
                                  bool take_sqrt, LocalLayerTables* out) {
  struct Entry { uint64_t dst_id; float weight; int32_t type; };
  const int64_t R = batch * n;
  out->seg.assign((size_t)batch + 1, 0);
  out->u_id.clear(); out->u_w.clear(); out->u_t.clear(); out->sum_w.clear();
  for (int64_t i = 0; i < batch; ++i) {
    const int32_t begin = idx[(size_t)i * n * 2];                       // :60-71
    const int32_t end = i + 1 < batch ? idx[(size_t)(i + 1) * n * 2] : idx[(size_t)R * 2 - 1];
    if (begin < 0 || end > total || begin > end) return false;
    std::unordered_map<std::string, Entry> uniq;
    for (int32_t j = begin; j < end; ++j) {                             // :72-83
      const std::string key = std::to_string(ids[j]) + std::to_string(t[j]);
      auto it = uniq.find(key);
      if (it == uniq.end()) uniq[key] = Entry{ids[j], w[j], t[j]};
      else it->second.weight += w[j];
    }
    float acc = 0.f;
    for (auto it = uniq.begin(); it != uniq.end(); ++it) {              // :86-114
      if (take_sqrt) it->second.weight = sqrtf(it->second.weight);
      out->u_id.push_back(it->second.dst_id);
      out->u_w.push_back(it->second.weight);
      out->u_t.push_back(it->second.type);
      acc += it->second.weight;           // CompactWeightedCollection::Init
      out->sum_w.push_back(acc);
    }
    out->seg[(size_t)i + 1] = (int64_t)out->u_id.size();
  }
  return true;
}

}  // namespace euler_gpu
