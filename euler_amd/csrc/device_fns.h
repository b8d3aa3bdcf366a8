// Device-side building blocks of the sampling kernels (gfx950).
#pragma once

#include "common.h"

// Correctly rounded, never-contracted f32 / f64 operations.  On the device
// these are the HIP intrinsics; the host spelling exists only so that
// tests/csrc/host_check.hip can run the SAME per-item functions on the CPU
// (the library is built -ffp-contract=off, so the plain operators round the
// same way).  No product path runs them on the host.
#if defined(__HIP_DEVICE_COMPILE__)
#define EG_FSUB(a, b) __fsub_rn((a), (b))
#define EG_FADD(a, b) __fadd_rn((a), (b))
#define EG_FMUL(a, b) __fmul_rn((a), (b))
#define EG_FDIV(a, b) __fdiv_rn((a), (b))
#define EG_DADD(a, b) __dadd_rn((a), (b))
#define EG_DMUL(a, b) __dmul_rn((a), (b))
#else
#define EG_FSUB(a, b) ((float)(a) - (float)(b))
#define EG_FADD(a, b) ((float)(a) + (float)(b))
#define EG_FMUL(a, b) ((float)(a) * (float)(b))
#define EG_FDIV(a, b) ((float)(a) / (float)(b))
#define EG_DADD(a, b) ((double)(a) + (double)(b))
#define EG_DMUL(a, b) ((double)(a) * (double)(b))
#endif

namespace euler_gpu {

// Graph::GetNodeByID (core/graph/graph.h:87-92): id -> row, -1 on a miss.
EG_HD int64_t FindRow(const GraphView& g, uint64_t id) {
  if (g.map_mode == 0) {
    const uint64_t d = id - g.id_base;
    if (id < g.id_base) return -1;
    if (g.id_stride == 1) return d < (uint64_t)g.n_rows ? (int64_t)d : -1;
    const uint64_t r = d / g.id_stride;
    return (r * g.id_stride == d && r < (uint64_t)g.n_rows) ? (int64_t)r : -1;
  }
  uint64_t h = Mix64(id) & g.hash_mask;
  // the table is at most half full, so a probe sequence always meets an empty
  // slot; the bound only keeps a corrupted table from hanging the GPU
  for (uint64_t probes = 0; probes <= g.hash_mask; ++probes) {
    // one 16-byte slot = {key, row}
    const ulonglong2 s =
        *reinterpret_cast<const ulonglong2*>(g.hash_slots + 2 * h);
    if ((int64_t)s.y < 0) return -1;
    if (s.x == id) return (int64_t)s.y;
    h = (h + 1) & g.hash_mask;
  }
  return -1;
}

struct RowMeta {
  int64_t row_ptr;
  const int32_t* type_end;     // [T]
  const float* type_prefix;    // [T]
};

EG_HD RowMeta LoadRowMeta(const GraphView& g, int64_t row) {
  const uint8_t* rec = g.row_meta + row * (int64_t)g.meta_stride;
  RowMeta m;
  m.row_ptr = *reinterpret_cast<const int64_t*>(rec);
  m.type_end = reinterpret_cast<const int32_t*>(rec + 8);
  m.type_prefix = reinterpret_cast<const float*>(rec + 8 + 4 * g.T);
  return m;
}

// r = u * (limit_end - limit_begin) + limit_begin exactly as the reference
// evaluates it on baseline x86-64 (no FMA): an f32 subtraction, then an fp64
// multiply and an fp64 add, each rounded (compact_weighted_collection.h:32-36).
EG_HD double ScaleDraw(double u, float limit_begin,
                                            float limit_end) {
  const float span = EG_FSUB(limit_end, limit_begin);
  return EG_DADD(EG_DMUL(u, (double)span), (double)limit_begin);
}

// RandomSelect<T> (common/compact_weighted_collection.h:30-52) over the
// running sums sw(i): same probe sequence, same unsigned index arithmetic,
// same fall-through (returns the last probed mid when no interval holds r).
template <typename SumAt>
EG_HD uint64_t RandomSelectT(const SumAt& sw,
                                                  uint64_t begin_pos,
                                                  uint64_t end_pos, double u) {
  const float limit_begin = begin_pos == 0 ? 0.f : sw(begin_pos - 1);
  const float limit_end = sw(end_pos);
  const double r = ScaleDraw(u, limit_begin, limit_end);
  uint64_t low = begin_pos, high = end_pos, mid = 0;
  bool finish = false;
  // a bisection over a 64-bit range ends within 64 probes; the counter only
  // bounds the loop on NaN sums (where the reference would spin forever)
  for (int probes = 0; low <= high && !finish && probes < 65; ++probes) {
    mid = (low + high) >> 1;
    const float interval_begin = mid == 0 ? 0.f : sw(mid - 1);
    const float interval_end = sw(mid);
    if ((double)interval_begin <= r && r < (double)interval_end) {
      finish = true;
    } else if ((double)interval_begin > r) {
      // r >= limit_begin = sw(begin_pos - 1) in every valid call, so this branch
      // is never taken at mid == begin_pos (Q4); the guard keeps a row with
      // decreasing sums (negative weights: out-of-range reads in the reference)
      // from walking outside [begin_pos, end_pos].
      if (mid == begin_pos) break;
      high = mid - 1;
    } else if ((double)interval_end <= r) {
      low = mid + 1;
    }
  }
  return mid;
}

struct ArraySum {
  const float* __restrict__ p;
  EG_HD float operator()(uint64_t i) const { return p[i]; }
};

EG_HD uint64_t RandomSelect(const float* sw,
                                                 uint64_t begin_pos,
                                                 uint64_t end_pos, double u) {
  return RandomSelectT(ArraySum{sw}, begin_pos, end_pos, u);
}

// Running sums of the sub collection rebuilt for the listed types
// (node.cc:106-121 + CompactWeightedCollection::Init): recomputed on demand
// with the same sequential f32 adds instead of being stored per lane.
struct SubTypeSum {
  const float* type_prefix;
  const int32_t* edge_types;
  EG_HD float operator()(uint64_t i) const {
    float s = 0.f;
    for (uint64_t x = 0; x <= i; ++x) {
      const int32_t t = edge_types[x];
      s = EG_FADD(s, EG_FSUB(type_prefix[t], t > 0 ? type_prefix[t - 1] : 0.f));
    }
    return s;
  }
};

// How Node::__SampleNeighbor (core/graph/node.cc:98-161) picks the edge type.
enum TypeMode : int32_t {
  kTypeSingle = 0,  // edge_types.size() == 1 : no type draw
  kTypeSub = 1,     // 1 < size < T : CDF over the listed types, listed order
  kTypeAll = 2      // size == 0 or >= T : CDF over all groups (list ignored, Q5)
};

EG_HD int32_t TypeModeOf(int32_t k, int32_t T) {
  return k == 1 ? kTypeSingle : (k > 1 && k < T) ? kTypeSub : kTypeAll;
}

// Per-row state every sample of the row needs.  `valid == false` reproduces
// the reference's err_vec / empty_vec returns: the row yields no samples.
struct RowSampler {
  bool valid;
  int32_t mode;
  int32_t k;
  int32_t single_type;
  int32_t T;
  const int32_t* edge_types;   // listed types (kernel-argument memory)
  const int32_t* type_end;
  const float* type_prefix;
  const float* nw;        // prefix_w + row_ptr
  const uint64_t* nbr;    // nbr + row_ptr
};

EG_HD void InitRowSampler(RowSampler& rs, const GraphView& g,
                                               int64_t row,
                                               const int32_t* edge_types,
                                               int32_t k) {
  rs.valid = false;
  rs.k = k;
  rs.T = g.T;
  rs.edge_types = edge_types;
  rs.mode = TypeModeOf(k, g.T);
  rs.single_type = 0;
  if (row < 0) return;
  const RowMeta m = LoadRowMeta(g, row);
  rs.type_end = m.type_end;
  rs.type_prefix = m.type_prefix;
  rs.nw = g.prefix_w + m.row_ptr;
  rs.nbr = g.nbr + m.row_ptr;
  if (rs.mode == kTypeSingle) {
    const int32_t t = edge_types[0];
    if (t < 0 || t >= g.T) return;                       // node.cc:127-130
    const int32_t pre_idx = t == 0 ? 0 : m.type_end[t - 1];
    const int32_t cur_idx = m.type_end[t] - 1;
    if (cur_idx < pre_idx) return;                       // node.cc:133-135
    rs.single_type = t;
    rs.valid = true;
  } else if (rs.mode == kTypeSub) {
    for (int32_t i = 0; i < k; ++i) {                    // node.cc:109-118
      const int32_t t = edge_types[i];
      if (t < 0 || t >= g.T) return;
    }
    const SubTypeSum sub{m.type_prefix, edge_types};
    if (sub((uint64_t)(k - 1)) == 0.f) return;           // node.cc:139-141
    rs.valid = true;
  } else {
    if (m.type_prefix[g.T - 1] == 0.f) return;           // node.cc:144-146
    rs.valid = true;
  }
}

// Sample number j of the row (node.cc:123-159).  Draw order per row is
// i-major, within i: [type draw,] neighbour draw; draw_idx therefore is j for
// the single-type mode and (2j, 2j+1) otherwise - one Philox block per sample.
EG_HD void SampleAt(const RowSampler& rs, uint64_t seed,
                                         uint32_t call_id, uint64_t node_id,
                                         int32_t j, uint64_t* out_id,
                                         float* out_w, int32_t* out_t,
                                         uint32_t domain = kDomainNeighbor) {
  int32_t t;
  double u_nb;
  if (rs.mode == kTypeSingle) {
    t = rs.single_type;
    const Philox4 b = RngBlock(seed, call_id, domain, node_id,
                               ((uint32_t)j) >> 1);
    const int h = j & 1;
    u_nb = h ? UnitFromWords(b.w[2], b.w[3]) : UnitFromWords(b.w[0], b.w[1]);
  } else {
    const Philox4 b = RngBlock(seed, call_id, domain, node_id,
                               (uint32_t)j);
    const double u_type = UnitFromWords(b.w[0], b.w[1]);
    u_nb = UnitFromWords(b.w[2], b.w[3]);
    if (rs.mode == kTypeSub) {
      const SubTypeSum sub{rs.type_prefix, rs.edge_types};
      t = rs.edge_types[RandomSelectT(sub, 0, (uint64_t)(rs.k - 1), u_type)];
    } else {
      t = (int32_t)RandomSelect(rs.type_prefix, 0, (uint64_t)(rs.T - 1), u_type);
    }
  }
  const int32_t b_idx = t == 0 ? 0 : rs.type_end[t - 1];
  const int32_t e_idx = rs.type_end[t] - 1;
  if (e_idx < b_idx) {
    // A zero-weight (empty) group can only be reached through an out-of-range
    // read in the reference (undefined behaviour); unreachable for consistent
    // rows.  Emit the sentinel instead of touching memory.
    *out_id = 0; *out_w = 0.f; *out_t = 0;
    return;
  }
  const uint64_t mid = RandomSelect(rs.nw, (uint64_t)b_idx, (uint64_t)e_idx, u_nb);
  const float pre = mid == 0 ? 0.f : rs.nw[mid - 1];
  *out_id = rs.nbr[mid];
  *out_w = EG_FSUB(rs.nw[mid], pre);
  *out_t = t;
}

}  // namespace euler_gpu
