// sample_neighbor kernels other than the default pivot kernels:
//  * SampleNeighborKernel - the reference's exact loop (type draws, zero-weight
//    rules, the id-0 sentinel rule, non-monotone rows): the PRODUCTION path of
//    every call that is not single-type on a monotone graph;
//  * the earlier single-type search variants (single-load bisection, ILP,
//    blocked index, wave-staged), kept selectable through euler_gpu_set_tuning
//    for A/B measurements and parity tests.
#include <hip/hip_runtime.h>

#include "k1_args.h"
#include "k1_search.h"

namespace euler_gpu {

__global__ __launch_bounds__(256, kWavesPerSimd) void SampleNeighborKernel(const SampleNbArgs a) {
  int64_t n_roots;
  if (!DedupGate(a, &n_roots)) return;
  const int64_t total = n_roots * (int64_t)a.count;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total;
       s += stride) {
    const int64_t r = s / a.count;
    const int32_t j = (int32_t)(s - r * a.count);
    uint64_t node = a.roots[r];
    if (a.root_mask != nullptr && a.root_mask[r / a.root_group]) node = 0;
    RowSampler rs;
    InitRowSampler(rs, a.g, FindRow(a.g, node), a.et, a.k);
    uint64_t id = 0;
    float w = 0.f;
    int32_t t = 0;
    bool masked = !rs.valid;
    if (rs.valid) {
      SampleAt(rs, a.seed, a.call_id, node, j, &id, &w, &t);
      if (a.layout == EULER_GPU_LAYOUT_TF) {
        // tf_euler/kernels/sample_neighbor_op.cc:114-122: the row is kept only
        // if its FIRST id is not the sentinel.  Only graphs that contain the
        // id 0 as a neighbour can have a live row that starts with 0.
        if (j == 0) {
          masked = id == 0;
        } else if (a.g.has_zero_nbr) {
          uint64_t id0; float w0; int32_t t0;
          SampleAt(rs, a.seed, a.call_id, node, 0, &id0, &w0, &t0);
          masked = id0 == 0;
        }
      }
    }
    if (a.layout == EULER_GPU_LAYOUT_TF) {
      if (masked) { id = (uint64_t)a.default_node; w = 0.f; t = -1; }
    } else if (masked) {
      id = 0; w = 0.f; t = 0;   // core/kernels/sample_neighbor_op.cc:134-143
    }
    a.out_id[s] = id;
    a.out_w[s] = w;
    a.out_t[s] = t;
    if (j == 0 && a.out_row_mask != nullptr) a.out_row_mask[r] = masked ? 1 : 0;
  }
}

// ------------------------------------------------------------------------
// Type draws on the block-pivot search.  A call that lists several edge types (or none:
// all of them) makes Node::__SampleNeighbor (core/graph/node.cc:98-161) draw the type
// first - a CDF over the row's per-type sums, in the LISTED order for 1 < k < T, over all
// groups otherwise - and then the neighbour inside that type's segment.  The reference
// loop above does the second draw with a bisection over the flat running sums
// (~log2(deg) dependent loads); here the type draw is the same code (a handful of
// entries out of the row record) and the neighbour draw is BlockPivotSample on the
// type's segment [lo, hi] of the row - the search of the single-type kernels, which
// takes any sub-range of a row.  Monotone graphs without the id-0 sentinel rule; same
// draws: one Philox block per sample, words 0-1 the type, words 2-3 the neighbour.
// ------------------------------------------------------------------------
template <bool TF_LAYOUT>
__global__ __launch_bounds__(256, kWavesPerSimd) void SampleNeighborTypedPivotKernel(
    const SampleNbArgs a) {
  int64_t n_roots;
  if (!DedupGate(a, &n_roots)) return;
  const int64_t total = n_roots * (int64_t)a.count;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int32_t mode = TypeModeOf(a.k, a.g.T);
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += stride) {
    const int64_t r = s / a.count;
    const int32_t j = (int32_t)(s - r * a.count);
    uint64_t node = a.roots[r];
    if (a.root_mask != nullptr && a.root_mask[r / a.root_group]) node = 0;
    const int64_t row = FindRow(a.g, node);
    uint64_t id = TF_LAYOUT ? (uint64_t)a.default_node : 0;
    float w = 0.f;
    int32_t t = TF_LAYOUT ? -1 : 0;
    bool valid = false;
    if (row >= 0) {
      const RowMeta m = LoadRowMeta(a.g, row);
      // node.cc:106-121,137-148: which rows have nothing to draw from
      if (mode == kTypeSub) {
        valid = true;
        for (int32_t i = 0; i < a.k; ++i) valid = valid && a.et[i] >= 0 && a.et[i] < a.g.T;
        if (valid) {
          const SubTypeSum sub{m.type_prefix, a.et};
          valid = sub((uint64_t)(a.k - 1)) != 0.f;
        }
      } else {
        valid = m.type_prefix[a.g.T - 1] != 0.f;
      }
      if (valid) {
        const Philox4 b = RngBlock(a.seed, a.call_id, kDomainNeighbor, node, (uint32_t)j);
        const double u_type = UnitFromWords(b.w[0], b.w[1]);
        const double u_nb = UnitFromWords(b.w[2], b.w[3]);
        if (mode == kTypeSub) {
          const SubTypeSum sub{m.type_prefix, a.et};
          t = a.et[RandomSelectT(sub, 0, (uint64_t)(a.k - 1), u_type)];
        } else {
          t = (int32_t)RandomSelect(m.type_prefix, 0, (uint64_t)(a.g.T - 1), u_type);
        }
        const int32_t b_idx = t == 0 ? 0 : m.type_end[t - 1];
        const int32_t e_idx = m.type_end[t] - 1;
        if (e_idx < b_idx) {
          // an empty group is only reachable through an out-of-range read in the
          // reference: the sentinel, as SampleAt (device_fns.h)
          id = 0; w = 0.f; t = 0;
        } else {
          Segment sg;
          sg.row_ptr = m.row_ptr; sg.b = b_idx; sg.e = e_idx;
          sg.lo = m.row_ptr + b_idx; sg.hi = m.row_ptr + e_idx;
          sg.limit_end = BlockedPw(a.g, sg.hi);
          sg.limit_begin = b_idx == 0 ? 0.f : BlockedPw(a.g, sg.lo - 1);
          sg.inl = nullptr;
          BlockPivotSample(a.g, sg, u_nb, &id, &w);
        }
      }
    }
    a.out_id[s] = id;
    a.out_w[s] = w;
    a.out_t[s] = t;
    if (j == 0 && a.out_row_mask != nullptr) a.out_row_mask[r] = valid ? 0 : 1;
  }
}

// ------------------------------------------------------------------------
// K1 fast path: single listed edge type (the GraphSAGE / DeepWalk case) on a
// graph whose prefix sums are monotone (GraphView::monotone).
//
// Same lane-per-sample mapping, but the per-sample instruction stream is cut
// to what the hardware needs:
//   * (root, slot) advance incrementally through the grid-stride loop - one
//     64-bit division per lane per launch instead of one per sample;
//   * 32-bit row-relative indices;
//   * the search is a single-load upper bound (first m with sw[m] > r).  With
//     non-decreasing sums the interval that holds r is unique, so this is the
//     index the reference's bisection returns (compact_weighted_collection.h:
//     37-50); when NO interval holds r (r rounded up to the segment's end, Q3)
//     the lane replays the reference's exact probe sequence instead.
// ------------------------------------------------------------------------
template <bool TF_LAYOUT, bool ZERO_CHECK>
__device__ __forceinline__ void FastSampleOne(const GraphView& g,
                                              const float* __restrict__ nw,
                                              const uint64_t* __restrict__ nbr,
                                              int32_t b, int32_t e, double u,
                                              uint64_t* out_id, float* out_w) {
  const float limit_begin = b == 0 ? 0.f : nw[b - 1];
  const float limit_end = nw[e];
  const double r = ScaleDraw(u, limit_begin, limit_end);
  int32_t lo = b, hi = e + 1;
  while (lo < hi) {
    const int32_t mid = (int32_t)(((uint32_t)lo + (uint32_t)hi) >> 1);
    if ((double)nw[mid] > r) hi = mid; else lo = mid + 1;
  }
  int32_t m = lo;
  float pre;
  if (m <= e) {
    pre = m == 0 ? 0.f : nw[m - 1];
  } else {
    // fall-through of RandomSelect: replay the reference probe sequence
    m = (int32_t)RandomSelect(nw, (uint64_t)b, (uint64_t)e, u);
    pre = m == 0 ? 0.f : nw[m - 1];
  }
  *out_id = nbr[m];
  *out_w = __fsub_rn(nw[m], pre);
}

template <bool TF_LAYOUT, bool ZERO_CHECK>
__global__ __launch_bounds__(256, kWavesPerSimd) void SampleNeighborFastKernel(
    const SampleNbArgs a, const int64_t stride_rows, const int32_t stride_slots) {
  int64_t n_roots;
  if (!DedupGate(a, &n_roots)) return;
  const int64_t total = n_roots * (int64_t)a.count;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= total) return;
  int64_t r = s / a.count;
  int32_t j = (int32_t)(s - r * a.count);
  const int32_t t = a.et[0];
  const int32_t T = a.g.T;
  for (; s < total; s += stride) {
    uint64_t node = a.roots[r];
    if (a.root_mask != nullptr && a.root_mask[r / a.root_group]) node = 0;
    const int64_t row = FindRow(a.g, node);
    uint64_t id = 0;
    float w = 0.f;
    bool valid = false;
    if (row >= 0 && t >= 0 && t < T) {
      const uint8_t* rec = a.g.row_meta + row * (int64_t)a.g.meta_stride;
      int64_t row_ptr;
      int32_t b, e;
      if (T == 1) {
        // {row_ptr, type_end[0], type_prefix[0]} in one 16-byte load
        const uint4 q = *reinterpret_cast<const uint4*>(rec);
        row_ptr = (int64_t)(((uint64_t)q.y << 32) | q.x);
        b = 0;
        e = (int32_t)q.z - 1;
      } else {
        row_ptr = *reinterpret_cast<const int64_t*>(rec);
        const int32_t* te = reinterpret_cast<const int32_t*>(rec + 8);
        b = t == 0 ? 0 : te[t - 1];
        e = te[t] - 1;
      }
      if (e >= b) {                                       // node.cc:133-135
        valid = true;
        const Philox4 blk = RngBlock(a.seed, a.call_id, kDomainNeighbor, node,
                                     ((uint32_t)j) >> 1);
        const double u = (j & 1) ? UnitFromWords(blk.w[2], blk.w[3])
                                 : UnitFromWords(blk.w[0], blk.w[1]);
        const float* nw = a.g.prefix_w + row_ptr;
        const uint64_t* nbr = a.g.nbr + row_ptr;
        FastSampleOne<TF_LAYOUT, ZERO_CHECK>(a.g, nw, nbr, b, e, u, &id, &w);
        if (TF_LAYOUT && ZERO_CHECK) {
          // Q1: the row is dropped when its FIRST sample is the sentinel id 0
          uint64_t id0 = id;
          if (j != 0) {
            const Philox4 b0 = RngBlock(a.seed, a.call_id, kDomainNeighbor, node, 0);
            float w0;
            FastSampleOne<TF_LAYOUT, ZERO_CHECK>(a.g, nw, nbr, b, e,
                                                 UnitFromWords(b0.w[0], b0.w[1]),
                                                 &id0, &w0);
          }
          valid = id0 != 0;
        }
      }
    }
    int32_t ot = t;
    if (!valid) {
      if (TF_LAYOUT) { id = (uint64_t)a.default_node; w = 0.f; ot = -1; }
      else { id = 0; w = 0.f; ot = 0; }
    }
    a.out_id[s] = id;
    a.out_w[s] = w;
    a.out_t[s] = ot;
    if (j == 0 && a.out_row_mask != nullptr) a.out_row_mask[r] = valid ? 0 : 1;
    // advance (root, slot) by the grid stride without dividing
    r += stride_rows;
    j += stride_slots;
    if (j >= a.count) { j -= a.count; ++r; }
  }
}

// ------------------------------------------------------------------------
// K1 ILP path: the same search as the fast path, U independent samples per
// lane advanced in lock step.  The kernel is bound by the latency of its chain
// of dependent loads (root -> row record -> limit -> ~log2(deg) probes -> id;
// SQ_WAIT_ANY = 87 % of wave time at full occupancy, 29 VGPRs), not by any
// one memory unit, so the lever is memory-level parallelism: U chains per
// lane keep U times as many requests in flight at the same occupancy.
// Values seen by the probes are carried along (nw[m], nw[m-1] are always among
// them), which removes the three trailing re-loads of the fast path.
// ------------------------------------------------------------------------
template <int U, bool TF_LAYOUT>
__global__ __launch_bounds__(256) void SampleNeighborIlpKernel(const SampleNbArgs a) {
  const int64_t total = a.n * (int64_t)a.count;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int32_t t = a.et[0];
  const int32_t T = a.g.T;
  const bool type_ok = t >= 0 && t < T;
  const bool small = total < (int64_t)0x7fffffff;
  for (int64_t s0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s0 < total;
       s0 += stride * U) {
    int64_t s[U], r[U];
    int32_t j[U], lo[U], hi[U], b[U], e[U];
    uint64_t node[U];
    const float* nw[U];
    const uint64_t* nbr[U];
    float vlo[U], vhi[U];
    double rr[U], u01[U];
    bool in[U], valid[U], replay[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s[u] = s0 + (int64_t)u * stride;
      in[u] = s[u] < total;
      const int64_t sc = in[u] ? s[u] : 0;
      if (small) {
        const uint32_t q = (uint32_t)sc / (uint32_t)a.count;
        r[u] = q;
        j[u] = (int32_t)((uint32_t)sc - q * (uint32_t)a.count);
      } else {
        r[u] = sc / a.count;
        j[u] = (int32_t)(sc - r[u] * a.count);
      }
      node[u] = a.roots[r[u]];
    }
    if (a.root_mask != nullptr) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (a.root_mask[r[u] / a.root_group]) node[u] = 0;
    }
    int64_t row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) row[u] = in[u] ? FindRow(a.g, node[u]) : -1;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      valid[u] = false;
      b[u] = 0; e[u] = -1;
      nw[u] = a.g.prefix_w; nbr[u] = a.g.nbr;
      if (row[u] >= 0 && type_ok) {
        const uint8_t* rec = a.g.row_meta + row[u] * (int64_t)a.g.meta_stride;
        int64_t row_ptr;
        if (T == 1) {
          const uint4 q = *reinterpret_cast<const uint4*>(rec);
          row_ptr = (int64_t)(((uint64_t)q.y << 32) | q.x);
          e[u] = (int32_t)q.z - 1;
        } else {
          row_ptr = *reinterpret_cast<const int64_t*>(rec);
          const int32_t* te = reinterpret_cast<const int32_t*>(rec + 8);
          b[u] = t == 0 ? 0 : te[t - 1];
          e[u] = te[t] - 1;
        }
        nw[u] = a.g.prefix_w + row_ptr;
        nbr[u] = a.g.nbr + row_ptr;
        valid[u] = e[u] >= b[u];                           // node.cc:133-135
      }
    }
    // limits of the searched segment (compact_weighted_collection.h:32-36)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      vlo[u] = 0.f; vhi[u] = 0.f;
      if (valid[u]) {
        vhi[u] = nw[u][e[u]];
        if (b[u] != 0) vlo[u] = nw[u][b[u] - 1];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const Philox4 blk = RngBlock(a.seed, a.call_id, kDomainNeighbor, node[u],
                                   ((uint32_t)j[u]) >> 1);
      u01[u] = (j[u] & 1) ? UnitFromWords(blk.w[2], blk.w[3])
                          : UnitFromWords(blk.w[0], blk.w[1]);
      rr[u] = ScaleDraw(u01[u], vlo[u], vhi[u]);
      // first m in [b, e] with nw[m] > r; nw[e] > r unless r was rounded up to
      // the segment's end (Q3) - those lanes replay the reference probes below
      replay[u] = valid[u] && !((double)vhi[u] > rr[u]);
      lo[u] = b[u];
      hi[u] = (valid[u] && !replay[u]) ? e[u] : b[u];
    }
    bool any = true;
    while (any) {
      any = false;
      float v[U];
      int32_t mid[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        mid[u] = (int32_t)(((uint32_t)lo[u] + (uint32_t)hi[u]) >> 1);
        if (lo[u] < hi[u]) v[u] = nw[u][mid[u]];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (lo[u] < hi[u]) {
          if ((double)v[u] > rr[u]) { hi[u] = mid[u]; vhi[u] = v[u]; }
          else { lo[u] = mid[u] + 1; vlo[u] = v[u]; }
          any |= lo[u] < hi[u];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint64_t id = 0;
      float w = 0.f;
      int32_t ot = t;
      if (valid[u]) {
        int32_t m = lo[u];
        if (replay[u]) {
          m = (int32_t)RandomSelect(nw[u], (uint64_t)b[u], (uint64_t)e[u], u01[u]);
          vhi[u] = nw[u][m];
          vlo[u] = m == 0 ? 0.f : nw[u][m - 1];
        }
        id = nbr[u][m];
        w = __fsub_rn(vhi[u], vlo[u]);
      } else if (TF_LAYOUT) {
        id = (uint64_t)a.default_node; ot = -1;
      } else {
        ot = 0;
      }
      if (in[u]) {
        a.out_id[s[u]] = id;
        a.out_w[s[u]] = w;
        a.out_t[s[u]] = ot;
        if (j[u] == 0 && a.out_row_mask != nullptr)
          a.out_row_mask[r[u]] = valid[u] ? 0 : 1;
      }
    }
  }
}
// ------------------------------------------------------------------------
// K1 blocked path (default): the search runs on the sampling index of
// common.h (EdgeBlock + skip levels) instead of the flat arrays.
//
// Measured on the metric workload (profiles/r1_*): the flat-array kernels are
// bound by the vector-memory pipeline, time ~= L1 accesses x 0.5 clk + L2 line
// fills x 2.3 clk + lines from beyond the L2 x 11.5 clk per CU, and the last
// term is the largest: every sample ends in two cold 128-byte lines, one of
// prefix_w (the last ~5 probes) and one of nbr (8 useful bytes).  Here both
// live in the same EdgeBlock line, and the upper probes walk skip arrays that
// are 10x / 320x / 10240x smaller than prefix_w (the last two stay in the L2),
// with every level confined to one line.  More samples per lane do not help
// (tools/ab_k1.py: U = 2/4/8 chains per lane are slower) - the kernel is
// bound by line throughput, not by latency.
//
// Search contract (same as the fast path): m = first index of [b, e] with
// nw[m] > r; rows are non-decreasing (GraphView::monotone), so that is the
// index RandomSelect returns; r >= nw[e] (Q3) replays the reference loop.
// ------------------------------------------------------------------------
// UpperBound32 / BlockedSearch: k1_search.h (the walk kernel shares them)

template <bool TF_LAYOUT>
__global__ __launch_bounds__(256, kWavesPerSimd) void SampleNeighborBlockedKernel(
    const SampleNbArgs a, const int64_t stride_rows, const int32_t stride_slots,
    const int32_t ablate) {
  const int64_t total = a.n * (int64_t)a.count;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= total) return;
  int64_t r = s / a.count;
  int32_t j = (int32_t)(s - r * a.count);
  const int32_t t = a.et[0];
  const int32_t T = a.g.T;
  for (; s < total; s += stride) {
    uint64_t node = a.roots[r];
    if (a.root_mask != nullptr && a.root_mask[r / a.root_group]) node = 0;
    const int64_t row = FindRow(a.g, node);
    uint64_t id = 0;
    float w = 0.f;
    bool valid = false;
    if (ablate >= 5 && ablate < 10) {
      id = node;
    } else if (row >= 0 && t >= 0 && t < T) {
      const uint8_t* rec = a.g.row_meta + row * (int64_t)a.g.meta_stride;
      int64_t row_ptr;
      int32_t b, e;
      if (ablate >= 4 && ablate < 10) {
        row_ptr = row * 10; b = 0; e = 9;
      } else if (T == 1) {
        const uint4 q = *reinterpret_cast<const uint4*>(rec);
        row_ptr = (int64_t)(((uint64_t)q.y << 32) | q.x);
        b = 0;
        e = (int32_t)q.z - 1;
      } else {
        row_ptr = *reinterpret_cast<const int64_t*>(rec);
        const int32_t* te = reinterpret_cast<const int32_t*>(rec + 8);
        b = t == 0 ? 0 : te[t - 1];
        e = te[t] - 1;
      }
      if (e >= b) {                                       // node.cc:133-135
        valid = true;
        double u;
        if (ablate >= 3 && ablate < 10) {
          u = (double)j * 0.03 + (double)(node & 1023) * 1e-4;
        } else {
          const Philox4 blk = RngBlock(a.seed, a.call_id, kDomainNeighbor, node,
                                       ((uint32_t)j) >> 1);
          u = (j & 1) ? UnitFromWords(blk.w[2], blk.w[3])
                      : UnitFromWords(blk.w[0], blk.w[1]);
        }
        float limit_begin = 0.f, limit_end = 1.f;
        if (ablate < 2 || ablate >= 10) {
          limit_begin = b == 0 ? 0.f : BlockedPw(a.g, row_ptr + b - 1);
          limit_end = BlockedPw(a.g, row_ptr + e);
        }
        const double rr = ScaleDraw(u, limit_begin, limit_end);
        if (ablate >= 1 && ablate < 10) {
          id = (uint64_t)row_ptr + (uint64_t)(int64_t)(rr * 1000.0);
          w = (float)rr;
        } else if ((double)limit_end > rr) {
          float pw_m, pw_prev;
          BlockedSearch(a.g, row_ptr, row_ptr + b, row_ptr + e, rr, &pw_m, &pw_prev,
                        &id, ablate);
          w = __fsub_rn(pw_m, pw_prev);
        } else {
          // Q3: r rounded up to the end of the segment - replay the reference
          const float* nw = a.g.prefix_w + row_ptr;
          const int32_t m = (int32_t)RandomSelect(nw, (uint64_t)b, (uint64_t)e, u);
          id = a.g.nbr[row_ptr + m];
          w = __fsub_rn(nw[m], m == 0 ? 0.f : nw[m - 1]);
        }
      }
    }
    int32_t ot = t;
    if (!valid) {
      if (TF_LAYOUT) { id = (uint64_t)a.default_node; w = 0.f; ot = -1; }
      else { id = 0; w = 0.f; ot = 0; }
    }
    a.out_id[s] = id;
    a.out_w[s] = w;
    a.out_t[s] = ot;
    if (j == 0 && a.out_row_mask != nullptr) a.out_row_mask[r] = valid ? 0 : 1;
    r += stride_rows;
    j += stride_slots;
    if (j >= a.count) { j -= a.count; ++r; }
  }
}

// ------------------------------------------------------------------------
// K1 wave-staged path (default for count >= 8).
//
// Ablation of the blocked kernel on the metric workload (tools/ab_k1.py, hop 2,
// 32.8 M samples): stores + root ids 0.14 ms, row record + limits + Philox
// 0.15 ms, skip-level probes 0.31 ms, leaf probes + id 0.29 ms.  Halving the L2
// and HBM traffic (flat -> blocked index) did not move the time; what it tracks
// is the number of lane-divergent vector-memory instructions (TCP busy ~100 %,
// 55-65 % of its cycles stalled on hits to pending lines): ~18 per sample.  A
// wave instruction whose 64 lanes touch 64 different lines costs the L1 about
// as much as eight instructions that fetch eight whole lines each.  So this
// kernel never lets a lane chase pointers on its own:
//
//   * one wave = 64 consecutive samples = the `count` samples of <= 10 roots;
//   * OWNER lanes (one per root) read the root id, the row record and the
//     segment limits once, pick the coarsest index level whose candidate range
//     fits 64 entries, and publish a descriptor in LDS;
//   * the wave copies each root's <= 64 candidate entries into LDS with one
//     coalesced load (the per-wavefront frontier buffer) and every sample
//     bisects them there;
//   * each remaining level is one 128-byte line per sample: 8 lanes fetch a
//     sample's skip node (3 lanes its leaf EdgeBlock sums) into LDS, again
//     searched in LDS; the only lane-divergent global load left is the 8-byte
//     neighbour id.
// Search results are those of BlockedSearch (same candidate ranges, same
// first-greater rule); Q3 lanes and rows beyond 64*32*32 blocks take the
// per-lane paths.
// ------------------------------------------------------------------------
constexpr int kStage = 64;           // staged candidate entries per root
constexpr int kMaxWaveRoots = 10;    // 64 / count + 2 for count >= 8
constexpr int kNodeStride = 36;      // floats per staged skip node (32 + pad)

struct RootDesc {
  int64_t row_ptr;      // first edge of the row (global index)
  int64_t lo, hi;       // searched segment [lo, hi], global edge indices
  uint64_t node;
  float limit_begin, limit_end;
  int32_t b_lo, b_hi;   // blocks of lo / hi
  int32_t level;        // staged level 1..3, 0 = per-lane search, -1 = invalid row
  int32_t x_lo;         // first staged entry
  int32_t cnt;          // staged entries: candidates [x_lo, x_lo + cnt), else x_lo + cnt
  int32_t pad;
};

struct alignas(16) WaveLds {
  float buf[32 * kNodeStride];               // 4.5 KB: skip nodes / leaf sums
  float top[kMaxWaveRoots][kStage];          // 2.5 KB: staged candidate entries
  RootDesc desc[kMaxWaveRoots];
};

// candidate range of a row at skip level `level` (see BlockedSearch)
__device__ __forceinline__ void LevelRange(int32_t level, int32_t b_lo, int32_t b_hi,
                                           int32_t* lo, int32_t* hi) {
  int32_t l = b_lo, h = b_hi;
  for (int32_t x = 1; x < level; ++x) {
    l = l / kSkipFanout;
    h = (h - 1) / kSkipFanout;
  }
  *lo = l; *hi = h;
}

template <bool TF_LAYOUT>
__global__ __launch_bounds__(256) void SampleNeighborWaveKernel(const SampleNbArgs a) {
  __shared__ WaveLds lds_all[4];
  WaveLds& L = lds_all[threadIdx.x >> 6];
  const int32_t lane = threadIdx.x & 63;
  const int64_t total = a.n * (int64_t)a.count;
  const int64_t n_chunks = (total + 63) >> 6;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int32_t t = a.et[0];
  const int32_t T = a.g.T;
  const bool small = total < (int64_t)0x7fffffff;
  for (int64_t chunk = wave0; chunk < n_chunks; chunk += n_waves) {
    const int64_t s_base = chunk << 6;
    const int64_t s = s_base + lane;
    const bool in = s < total;
    const int64_t s_last = s_base + 63 < total ? s_base + 63 : total - 1;
    int64_t r_first, r_last;
    if (small) {
      r_first = (uint32_t)s_base / (uint32_t)a.count;
      r_last = (uint32_t)s_last / (uint32_t)a.count;
    } else {
      r_first = s_base / a.count;
      r_last = s_last / a.count;
    }
    const int32_t nr = (int32_t)(r_last - r_first) + 1;
    // ---- owner phase: lane q < nr owns root r_first + q -------------------
    if (lane < nr) {
      RootDesc d;
      const int64_t r = r_first + lane;
      uint64_t node = a.roots[r];
      if (a.root_mask != nullptr && a.root_mask[r / a.root_group]) node = 0;
      d.node = node;
      d.level = -1;
      d.row_ptr = 0; d.lo = 0; d.hi = -1; d.limit_begin = 0.f; d.limit_end = 0.f;
      d.b_lo = 0; d.b_hi = 0; d.x_lo = 0; d.cnt = 0; d.pad = 0;
      const int64_t row = FindRow(a.g, node);
      if (row >= 0 && t >= 0 && t < T) {
        const uint8_t* rec = a.g.row_meta + row * (int64_t)a.g.meta_stride;
        int64_t row_ptr;
        int32_t b, e;
        if (T == 1) {
          const uint4 q = *reinterpret_cast<const uint4*>(rec);
          row_ptr = (int64_t)(((uint64_t)q.y << 32) | q.x);
          b = 0;
          e = (int32_t)q.z - 1;
        } else {
          row_ptr = *reinterpret_cast<const int64_t*>(rec);
          const int32_t* te = reinterpret_cast<const int32_t*>(rec + 8);
          b = t == 0 ? 0 : te[t - 1];
          e = te[t] - 1;
        }
        if (e >= b) {                                     // node.cc:133-135
          d.row_ptr = row_ptr;
          d.lo = row_ptr + b;
          d.hi = row_ptr + e;
          d.limit_begin = b == 0 ? 0.f : BlockedPw(a.g, d.lo - 1);
          d.limit_end = BlockedPw(a.g, d.hi);
          d.b_lo = (int32_t)(d.lo / kEdgesPerBlock);
          d.b_hi = (int32_t)(d.hi / kEdgesPerBlock);
          int32_t level = 1, xl = d.b_lo, xh = d.b_hi;
          while (level <= 3 && xh - xl > kStage) {
            xl = xl / kSkipFanout;
            xh = (xh - 1) / kSkipFanout;
            ++level;
          }
          d.level = level <= 3 ? level : 0;
          d.x_lo = xl;
          d.cnt = xh - xl;
        }
      }
      L.desc[lane] = d;
    }
    WaveSync();
    // ---- stage every root's candidate entries (one coalesced load each) -----
    for (int32_t q = 0; q < nr; ++q) {
      const int32_t level = L.desc[q].level;
      const int32_t cnt = L.desc[q].cnt;
      if (level >= 1 && lane < cnt) {
        const float* arr = level == 1 ? a.g.skip1 : level == 2 ? a.g.skip2 : a.g.skip3;
        L.top[q][lane] = arr[L.desc[q].x_lo + lane];
      }
    }
    WaveSync();
    // ---- sample phase ---------------------------------------------------------
    int64_t r = r_first;
    int32_t j = 0, q = 0;
    if (in) {
      if (small) {
        const uint32_t rq = (uint32_t)s / (uint32_t)a.count;
        r = rq;
        j = (int32_t)((uint32_t)s - rq * (uint32_t)a.count);
      } else {
        r = s / a.count;
        j = (int32_t)(s - r * a.count);
      }
      q = (int32_t)(r - r_first);
    }
    int32_t level = in ? L.desc[q].level : -1;
    const bool valid = level >= 0;
    const int32_t b_lo = L.desc[q].b_lo, b_hi = L.desc[q].b_hi;
    double u = 0.0, rr = 0.0;
    bool replay = false;
    int32_t x = 0;
    if (valid) {
      const uint64_t node = L.desc[q].node;
      const Philox4 blk = RngBlock(a.seed, a.call_id, kDomainNeighbor, node,
                                   ((uint32_t)j) >> 1);
      u = (j & 1) ? UnitFromWords(blk.w[2], blk.w[3])
                  : UnitFromWords(blk.w[0], blk.w[1]);
      const float limit_end = L.desc[q].limit_end;
      rr = ScaleDraw(u, L.desc[q].limit_begin, limit_end);
      replay = !((double)limit_end > rr);                 // Q3
      if (!replay && level >= 1)
        x = L.desc[q].x_lo + UpperBound32(L.top[q], 0, L.desc[q].cnt, rr);
    }
    // ---- remaining skip levels: one 128-byte node per sample, via LDS ---------
    for (int32_t lv = 3; lv >= 2; --lv) {
      const bool act = valid && !replay && level == lv;
      if (__ballot(act) == 0) continue;
      const float* arr = lv == 3 ? a.g.skip2 : a.g.skip1;   // level lv - 1
      const int32_t xs = act ? x : -1;
      int32_t c_lo = 0, c_hi = 0;
      if (act) {
        LevelRange(lv - 1, b_lo, b_hi, &c_lo, &c_hi);
        c_lo = max(c_lo, x * kSkipFanout);
        c_hi = min(c_hi, x * kSkipFanout + kSkipFanout);
      }
      for (int32_t half = 0; half < 2; ++half) {
        WaveSync();
#pragma unroll
        for (int32_t k = 0; k < 4; ++k) {
          const int32_t idx = 8 * k + (lane >> 3);          // sample of this half
          const int32_t src = __shfl(xs, 32 * half + idx);
          if (src >= 0) {
            const uint4 v = *reinterpret_cast<const uint4*>(
                arr + (int64_t)src * kSkipFanout + 4 * (lane & 7));
            *reinterpret_cast<uint4*>(&L.buf[idx * kNodeStride + 4 * (lane & 7)]) = v;
          }
        }
        WaveSync();
        if (act && (lane >> 5) == half) {
          const float* nodep = &L.buf[(lane & 31) * kNodeStride];
          const int32_t base = x * kSkipFanout;
          x = base + UpperBound32(nodep, c_lo - base, c_hi - base, rr);
          level = lv - 1;
        }
      }
    }
    // ---- leaf: the 10 running sums of block x, 3 lanes x 16 B per sample ------
    const bool leaf = valid && !replay && level == 1;
    uint64_t id = 0;
    float w = 0.f;
    {
      const int32_t xs = leaf ? x : -1;
      WaveSync();
#pragma unroll
      for (int32_t k = 0; k < 4; ++k) {
        const int32_t idx = 21 * k + lane / 3;
        const int32_t part = lane - (lane / 3) * 3;
        const int32_t src = __shfl(xs, idx < 64 ? idx : 63);
        if (lane < 63 && idx < 64 && src >= 0) {
          const uint4 v = *reinterpret_cast<const uint4*>(
              reinterpret_cast<const char*>(a.g.blk + src) + 16 * part);
          *reinterpret_cast<uint4*>(&L.buf[idx * 12 + 4 * part]) = v;
        }
      }
      WaveSync();
      if (leaf) {
        const float* pw = &L.buf[lane * 12];
        const int64_t base = (int64_t)x * kEdgesPerBlock;
        const int64_t lo = L.desc[q].lo, hi = L.desc[q].hi;
        const int32_t i_lo = lo > base ? (int32_t)(lo - base) : 0;
        const int32_t i_hi = hi - base < kEdgesPerBlock - 1 ? (int32_t)(hi - base)
                                                            : kEdgesPerBlock - 1;
        const int32_t i = UpperBound32(pw, i_lo, i_hi, rr);
        float prev;
        if (base + i == L.desc[q].row_ptr) prev = 0.f;
        else if (i > 0) prev = pw[i - 1];
        else prev = a.g.skip1[x - 1];
        w = __fsub_rn(pw[i], prev);
        id = a.g.blk[x].nbr[i];
      }
    }
    if (valid && !replay && level == 0) {
      // more than 64*32*32 blocks: per-lane search on the index
      float pw_m, pw_prev;
      BlockedSearch(a.g, L.desc[q].row_ptr, L.desc[q].lo, L.desc[q].hi, rr, &pw_m,
                    &pw_prev, &id);
      w = __fsub_rn(pw_m, pw_prev);
    }
    if (valid && replay) {
      // Q3: r rounded up to the end of the segment - replay the reference
      const int64_t row_ptr = L.desc[q].row_ptr;
      const float* nw = a.g.prefix_w + row_ptr;
      const int32_t m = (int32_t)RandomSelect(
          nw, (uint64_t)(L.desc[q].lo - row_ptr), (uint64_t)(L.desc[q].hi - row_ptr), u);
      id = a.g.nbr[row_ptr + m];
      w = __fsub_rn(nw[m], m == 0 ? 0.f : nw[m - 1]);
    }
    int32_t ot = t;
    if (!valid) {
      if (TF_LAYOUT) { id = (uint64_t)a.default_node; w = 0.f; ot = -1; }
      else { id = 0; w = 0.f; ot = 0; }
    }
    if (in) {
      a.out_id[s] = id;
      a.out_w[s] = w;
      a.out_t[s] = ot;
      if (j == 0 && a.out_row_mask != nullptr) a.out_row_mask[r] = valid ? 0 : 1;
    }
    WaveSync();   // descriptors / buffers are rewritten by the next chunk
  }
}

template <int U>
static void LaunchIlp(bool tf, int grid, int block, hipStream_t stream,
                      const SampleNbArgs& a) {
  if (tf) {
    hipLaunchKernelGGL((SampleNeighborIlpKernel<U, true>), dim3(grid), dim3(block),
                       0, stream, a);
  } else {
    hipLaunchKernelGGL((SampleNeighborIlpKernel<U, false>), dim3(grid), dim3(block),
                       0, stream, a);
  }
}

int LaunchK1Variant(const euler_gpu_graph* g, hipStream_t stream, const SampleNbArgs& a,
                    int grid) {
  const int64_t n = a.n;
  const int32_t count = a.count, layout = a.layout, k = a.k;
  const int block = 256;
  const bool single = k == 1 && g->view.monotone;
  const bool tf_zero = layout == EULER_GPU_LAYOUT_TF && g->view.has_zero_nbr != 0;
  if (g_k1_variant == 4 && single && !tf_zero && count >= 8) {
    const int64_t chunks = (n * (int64_t)count + 63) / 64;
    int64_t blocks = (chunks + 3) / 4;
    if (blocks > 256 * 5) blocks = 256 * 5;     // 5 blocks of 4 waves per CU (LDS)
    if (blocks < 1) blocks = 1;
    if (layout == EULER_GPU_LAYOUT_TF) {
      hipLaunchKernelGGL(SampleNeighborWaveKernel<true>, dim3((int)blocks),
                         dim3(block), 0, stream, a);
    } else {
      hipLaunchKernelGGL(SampleNeighborWaveKernel<false>, dim3((int)blocks),
                         dim3(block), 0, stream, a);
    }
  } else if (g_k1_variant >= 3 && single && !tf_zero) {
    const int64_t stride = (int64_t)grid * block;
    const int64_t stride_rows = stride / count;
    const int32_t stride_slots = (int32_t)(stride - stride_rows * count);
    if (layout == EULER_GPU_LAYOUT_TF) {
      hipLaunchKernelGGL(SampleNeighborBlockedKernel<true>, dim3(grid), dim3(block),
                         0, stream, a, stride_rows, stride_slots, g_k1_ablate);
    } else {
      hipLaunchKernelGGL(SampleNeighborBlockedKernel<false>, dim3(grid), dim3(block),
                         0, stream, a, stride_rows, stride_slots, g_k1_ablate);
    }
  } else if (g_k1_variant == 2 && single && !tf_zero) {
    const bool tf = layout == EULER_GPU_LAYOUT_TF;
    const int U = g_k1_ilp;
    const int gridu = GridFor((n * (int64_t)count + U - 1) / U, block);
    if (U == 1) LaunchIlp<1>(tf, gridu, block, stream, a);
    else if (U == 2) LaunchIlp<2>(tf, gridu, block, stream, a);
    else if (U == 8) LaunchIlp<8>(tf, gridu, block, stream, a);
    else LaunchIlp<4>(tf, gridu, block, stream, a);
  } else if (g_k1_variant >= 1 && single) {
    const int64_t stride = (int64_t)grid * block;
    const int64_t stride_rows = stride / count;
    const int32_t stride_slots = (int32_t)(stride - stride_rows * count);
    const bool tf = layout == EULER_GPU_LAYOUT_TF;
    const bool zc = g->view.has_zero_nbr != 0;
    auto kern = tf ? (zc ? SampleNeighborFastKernel<true, true>
                         : SampleNeighborFastKernel<true, false>)
                   : SampleNeighborFastKernel<false, false>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, stream, a, stride_rows,
                       stride_slots);
  } else if (g_k1_variant == 6 && g_k1_typed_pivot != 0 && k != 1 && g->view.monotone &&
             !tf_zero && g->view.blk != nullptr) {
    if (layout == EULER_GPU_LAYOUT_TF) {
      hipLaunchKernelGGL(SampleNeighborTypedPivotKernel<true>, dim3(grid), dim3(block), 0, stream, a);
    } else {
      hipLaunchKernelGGL(SampleNeighborTypedPivotKernel<false>, dim3(grid), dim3(block), 0, stream, a);
    }
  } else {
    hipLaunchKernelGGL(SampleNeighborKernel, dim3(grid), dim3(block), 0, stream, a);
  }
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

}  // namespace euler_gpu
