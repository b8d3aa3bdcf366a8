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
#include <thread>
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

// One batch row: its entries appended to the row's own vectors.
struct LocalLayerRow {
  std::vector<uint64_t> u_id;
  std::vector<float> u_w, sum_w;
  std::vector<int32_t> u_t;
};

inline void BuildLocalLayerRow(const uint64_t* ids, const float* w, const int32_t* t,
                               int32_t begin, int32_t end, bool take_sqrt, LocalLayerRow* row) {
  struct Entry { uint64_t dst_id; float weight; int32_t type; };
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
    row->u_id.push_back(it->second.dst_id);
    row->u_w.push_back(it->second.weight);
    row->u_t.push_back(it->second.type);
    acc += it->second.weight;           // CompactWeightedCollection::Init
    row->sum_w.push_back(acc);
  }
}

// idx [batch * n, 2] / ids / w / t [total]: the API_GET_NB_NODE result.  false
// when idx does not index the value arrays.  The batch rows are independent
// (one container each, as in the reference) and are built by up to 8 host threads.
inline bool BuildLocalLayerTables(const int32_t* idx, const uint64_t* ids, const float* w,
                                  const int32_t* t, int64_t total, int64_t batch, int32_t n,
                                  bool take_sqrt, LocalLayerTables* out) {
  const int64_t R = batch * n;
  std::vector<int32_t> begin((size_t)batch), end((size_t)batch);
  for (int64_t i = 0; i < batch; ++i) {
    begin[i] = idx[(size_t)i * n * 2];                                // :60-71
    end[i] = i + 1 < batch ? idx[(size_t)(i + 1) * n * 2] : idx[(size_t)R * 2 - 1];
    if (begin[i] < 0 || end[i] > total || begin[i] > end[i]) return false;
  }
  std::vector<LocalLayerRow> rows((size_t)batch);
  auto work = [&](int64_t b0, int64_t b1) {
    for (int64_t i = b0; i < b1; ++i)
      BuildLocalLayerRow(ids, w, t, begin[i], end[i], take_sqrt, &rows[i]);
  };
  const int64_t hw = (int64_t)std::thread::hardware_concurrency();
  int64_t n_thr = hw < 8 ? hw : 8;
  if (n_thr > batch) n_thr = batch;
  if (total < (1 << 14) || n_thr <= 1) {
    work(0, batch);
  } else {
    std::vector<std::thread> pool;
    for (int64_t k = 0; k < n_thr; ++k)
      pool.emplace_back(work, batch * k / n_thr, batch * (k + 1) / n_thr);
    for (auto& th : pool) th.join();
  }
  out->seg.assign((size_t)batch + 1, 0);
  out->u_id.clear(); out->u_w.clear(); out->u_t.clear(); out->sum_w.clear();
  for (int64_t i = 0; i < batch; ++i) {
    const LocalLayerRow& r = rows[i];
    out->u_id.insert(out->u_id.end(), r.u_id.begin(), r.u_id.end());
    out->u_w.insert(out->u_w.end(), r.u_w.begin(), r.u_w.end());
    out->u_t.insert(out->u_t.end(), r.u_t.begin(), r.u_t.end());
    out->sum_w.insert(out->sum_w.end(), r.sum_w.begin(), r.sum_w.end());
    out->seg[(size_t)i + 1] = (int64_t)out->u_id.size();
  }
  return true;
}

}  // namespace euler_gpu
