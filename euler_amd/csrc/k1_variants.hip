// sample_neighbor kernels other than the single-type pivot kernels:
//  * SampleNeighborKernel - the reference's exact loop (type draws, zero-weight
//    rules, the id-0 sentinel rule, non-monotone rows): every call the searches
//    below and in sample_kernels.hip do not serve;
//  * SampleNeighborTypedPivotKernel - type draws + the block-pivot search.
// (Rounds 1-2 kept four earlier single-type searches selectable here - single-load
// bisection, several samples per lane, the 32-ary blocked index, wave-staged lines;
// they are in the history, DESIGN.md 4 says what each showed.)
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
// Round 3 measured two further cuts and dropped both (hetero workload, 1.3 M samples per
// launch, 0.060-0.065 ms either way): the row record read into registers in one trip for
// T <= 8 (select trees over 16 registers cost more VALU than the 3-4 L1 probes they
// replace: 0.083 ms) and limits taken from the EdgeBlock's own sums when the segment lies
// inside one block (no change: the leaf read dominates).  Two samples of a root per lane
// (the single-type kernels' pair mode; one record lookup per pair, half the waves) was
// slower too: 0.041 / 0.044 vs 0.036 / 0.032 ms for 3 of 8 / all types (two Philox blocks,
// two type draws and two searches in one lane: 64 registers no longer hold it).
template <bool TF_LAYOUT>
__global__ __launch_bounds__(256, kWavesPerSimd) void SampleNeighborTypedPivotKernel(
    const SampleNbArgs a) {
  int64_t n_roots;
  if (!DedupGate(a, &n_roots)) return;
  const int64_t total = n_roots * (int64_t)a.count;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int32_t mode = TypeModeOf(a.k, a.g.T);
  const int32_t T = a.g.T;
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
      const RowMeta m = LoadRowMetaWb(a.g, row);
      // node.cc:106-121,137-148: which rows have nothing to draw from
      if (mode == kTypeSub) {
        valid = true;
        for (int32_t i = 0; i < a.k; ++i) valid = valid && a.et[i] >= 0 && a.et[i] < T;
        if (valid) valid = SubTypeSum{m.type_prefix, a.et}((uint64_t)(a.k - 1)) != 0.f;
      } else {
        valid = m.type_prefix[T - 1] != 0.f;
      }
      if (valid) {
        const Philox4 b = RngBlock(a.seed, a.call_id, kDomainNeighbor, node, (uint32_t)j);
        const double u_type = UnitFromWords(b.w[0], b.w[1]);
        const double u_nb = UnitFromWords(b.w[2], b.w[3]);
        if (mode == kTypeSub) {
          t = a.et[RandomSelectT(SubTypeSum{m.type_prefix, a.et}, 0, (uint64_t)(a.k - 1), u_type)];
        } else {
          t = (int32_t)RandomSelect(m.type_prefix, 0, (uint64_t)(T - 1), u_type);
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
          if (a.g.wbg != nullptr) {
            LoadWbSegment(a.g, row, t, m.type_end[T - 1], &sg);
          } else {
            sg.limit_end = BlockedPw(a.g, sg.hi);
            sg.limit_begin = b_idx == 0 ? 0.f : BlockedPw(a.g, sg.lo - 1);
          }
          BlockPivotSample(a.g, sg, u_nb, &id, &w);
        }
      }
    }
    if (a.packed != nullptr) {
      // wire row of root r (k1_args.h: PackedWords): ids (2 words each) | weights | types | mask, pad
      int32_t* prow = a.packed + r * (int64_t)PackedWords(a.count, a.packed_tcol);
      *reinterpret_cast<uint64_t*>(prow + 2 * j) = id;
      prow[2 * a.count + j] = __float_as_int(w);
      if (a.packed_tcol) prow[3 * a.count + j] = t;
      if (j == 0) {
        prow[(3 + a.packed_tcol) * a.count] = valid ? 0 : 1;
        prow[(3 + a.packed_tcol) * a.count + 1] = 0;
      }
    } else {
      a.out_id[s] = id;
      a.out_w[s] = w;
      a.out_t[s] = t;
    }
    if (j == 0 && a.out_row_mask != nullptr) a.out_row_mask[r] = valid ? 0 : 1;
  }
}

// ------------------------------------------------------------------------
// Several edge-type SETS of one batch of roots in ONE launch (euler_gpu_sample_neighbor_sets).
// A heterogeneous model samples the same roots once per relation set - RGCN-style:
// sample_neighbor(nodes, [3]), sample_neighbor(nodes, [1, 4, 6]), sample_neighbor(nodes, all) -
// three launches of ~2.5 rounds of waves each, every one a chain of dependent cold loads
// (root -> record -> [type draw] -> block).  The calls are independent, so one launch over
// sets x roots x count samples keeps three times as many of those chains in flight.  Set s
// draws with call_id + s: the results are those of S euler_gpu_sample_neighbor calls, bit for
// bit (one listed type: draw j = Philox block j >> 1, half j & 1; otherwise block j, words
// 0-1 the type, 2-3 the neighbour - node.cc:123-159).  TF layout, monotone graphs without
// the id-0 sentinel rule.
// ------------------------------------------------------------------------
constexpr int kMaxTypeSets = 8;
struct SampleSetsArgs {
  GraphView g;
  uint64_t seed;
  const uint64_t* roots;
  uint64_t* out_id; float* out_w; int32_t* out_t;     // [sets][n][count]
  int64_t n;
  int64_t default_node;
  uint32_t call_id;
  int32_t count;
  int32_t n_sets;
  int32_t set_k[kMaxTypeSets];
  int32_t set_off[kMaxTypeSets];
  int32_t et[kMaxListedTypes];
  // not null: wire rows (k1_args.h: PackedWords) instead of out_id / out_w / out_t - the owners' pass
  // of a typed sharded hop; set s's rows begin packed_off[s] words into `packed` (without the type
  // column when the set lists one type)
  int32_t* packed;
  int64_t packed_off[kMaxTypeSets];
};

// sample j of root r of set `set`: into the three arrays, or into the root's wire row (t < 0 = the row
// has nothing to draw from: the row mask)
__device__ __forceinline__ void SetsStore(const SampleSetsArgs& a, int32_t set, int64_t r, int32_t j,
                                          uint64_t id, float w, int32_t t) {
  if (a.packed != nullptr) {
    const int32_t tcol = a.set_k[set] == 1 ? 0 : 1;
    int32_t* prow = a.packed + a.packed_off[set] + r * (int64_t)PackedWords(a.count, tcol);
    *reinterpret_cast<uint64_t*>(prow + 2 * j) = id;
    prow[2 * a.count + j] = __float_as_int(w);
    if (tcol) prow[3 * a.count + j] = t;
    if (j == 0) {
      prow[(3 + tcol) * a.count] = t < 0 ? 1 : 0;
      prow[(3 + tcol) * a.count + 1] = 0;
    }
    return;
  }
  const int64_t s = ((int64_t)set * a.n + r) * a.count + j;
  a.out_id[s] = id;
  a.out_w[s] = w;
  a.out_t[s] = t;
}

__global__ __launch_bounds__(256, 6) void SampleNeighborSetsKernel(const SampleSetsArgs a) {   // 80 VGPRs: nothing spilled
  const int64_t per_set = a.n * (int64_t)a.count;
  const int64_t total = per_set * a.n_sets;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int32_t T = a.g.T;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += stride) {
    const int32_t set = (int32_t)(s / per_set);
    const int64_t in_set = s - (int64_t)set * per_set;
    const int64_t r = in_set / a.count;
    const int32_t j = (int32_t)(in_set - r * a.count);
    const int32_t k = a.set_k[set];
    const int32_t* et = a.et + a.set_off[set];
    const int32_t mode = TypeModeOf(k, T);
    const uint32_t call = a.call_id + (uint32_t)set;
    const uint64_t node = a.roots[r];
    const int64_t row = FindRow(a.g, node);
    uint64_t id = (uint64_t)a.default_node;
    float w = 0.f;
    int32_t t = -1;
    if (mode == kTypeSingle) {
      Segment sg;
      if (LoadSegment<true>(a.g, row, et[0], &sg)) {
        const Philox4 b = RngBlock(a.seed, call, kDomainNeighbor, node, ((uint32_t)j) >> 1);
        const double u = (j & 1) ? UnitFromWords(b.w[2], b.w[3]) : UnitFromWords(b.w[0], b.w[1]);
        BlockPivotSample(a.g, sg, u, &id, &w);
        t = et[0];
      }
    } else if (row >= 0) {
      const RowMeta m = LoadRowMetaWb(a.g, row);
      bool valid;
      if (mode == kTypeSub) {
        valid = true;
        for (int32_t i = 0; i < k; ++i) valid = valid && et[i] >= 0 && et[i] < T;
        if (valid) valid = SubTypeSum{m.type_prefix, et}((uint64_t)(k - 1)) != 0.f;
      } else {
        valid = m.type_prefix[T - 1] != 0.f;
      }
      if (valid) {
        const Philox4 b = RngBlock(a.seed, call, kDomainNeighbor, node, (uint32_t)j);
        const double u_type = UnitFromWords(b.w[0], b.w[1]);
        const double u_nb = UnitFromWords(b.w[2], b.w[3]);
        if (mode == kTypeSub) {
          t = et[RandomSelectT(SubTypeSum{m.type_prefix, et}, 0, (uint64_t)(k - 1), u_type)];
        } else {
          t = (int32_t)RandomSelect(m.type_prefix, 0, (uint64_t)(T - 1), u_type);
        }
        const int32_t b_idx = t == 0 ? 0 : m.type_end[t - 1];
        const int32_t e_idx = m.type_end[t] - 1;
        if (e_idx < b_idx) {
          id = 0; w = 0.f; t = 0;         // as SampleAt (device_fns.h): unreachable for consistent rows
        } else {
          Segment sg;
          sg.row_ptr = m.row_ptr; sg.b = b_idx; sg.e = e_idx;
          sg.lo = m.row_ptr + b_idx; sg.hi = m.row_ptr + e_idx;
          if (a.g.wbg != nullptr) {
            LoadWbSegment(a.g, row, t, m.type_end[T - 1], &sg);
          } else {
            sg.limit_end = BlockedPw(a.g, sg.hi);
            sg.limit_begin = b_idx == 0 ? 0.f : BlockedPw(a.g, sg.lo - 1);
          }
          BlockPivotSample(a.g, sg, u_nb, &id, &w);
        }
      }
    }
    SetsStore(a, set, r, j, id, w, t);
  }
}

// The same launch with every root's records staged ONCE per workgroup.  In the kernel above each
// of a root's `count` lanes reads the root's row record and weight-bucket record for itself -
// a dozen small loads per lane, all hits, and yet the launch is bound by exactly that: the CU's
// memory pipe works off load instructions x lanes, not bytes.  Here a workgroup takes
// 256 / count ROOTS: their rows are found by one lane each, their records (8 + 12 T bytes of
// {wb_lo, row_lo, type_end[T], lim[T], type_sum[T]}) are copied into LDS by all lanes word by word - once
// for all the sets - and a sample lane then needs global memory only for its block's keys and
// its id, set after set.  Graphs with the weight-bucket index only (the kernel above serves
// the rest).
template <int WPS>
__global__ __launch_bounds__(256, WPS) void SampleNeighborSetsLdsKernel(const SampleSetsArgs a,
                                                                       const int32_t rpb) {
  extern __shared__ __align__(16) uint32_t sl_smem[];
  const int32_t T = a.g.T;
  // per root: the record of the weight-bucket index (first block, first edge, group ends,
  // their running sums, the type sums: 2 + 3 T words; at T = 1 the total stands for the sum)
  const int32_t mw = 2 + 2 * T, W = T == 1 ? 4 : 2 + 3 * T;
  int64_t* s_row = reinterpret_cast<int64_t*>(sl_smem);                  // [rpb]
  uint32_t* s_rec = sl_smem + 2 * rpb;                                    // [rpb][W]
  const int32_t q = (int32_t)threadIdx.x / a.count, j = (int32_t)threadIdx.x - q * a.count;
  for (int64_t r0 = (int64_t)blockIdx.x * rpb; r0 < a.n; r0 += (int64_t)gridDim.x * rpb) {
    const int32_t nt = (int32_t)(a.n - r0 < (int64_t)rpb ? a.n - r0 : (int64_t)rpb);
    // (1) rows of the roots
    if ((int32_t)threadIdx.x < nt) s_row[threadIdx.x] = FindRow(a.g, a.roots[r0 + threadIdx.x]);
    __syncthreads();
    // (2) their records, word by word (consecutive lanes, consecutive words of one root)
    for (int32_t x = (int32_t)threadIdx.x; x < nt * W; x += 256) {
      const int32_t r = x / W, k = x - r * W;
      const int64_t row = s_row[r];
      uint32_t v = 0;
      if (row >= 0) v = reinterpret_cast<const uint32_t*>(a.g.wbg + row * (int64_t)a.g.wbg_stride)[k];
      s_rec[r * W + k] = v;
    }
    __syncthreads();
    // (3) the samples, one set after the other
    if (q < nt) {
      const int64_t r = r0 + q;
      const int64_t row = s_row[q];
      const uint32_t* rec = s_rec + q * W;
      const int32_t* type_end = reinterpret_cast<const int32_t*>(rec + 2);
      const float* lim = reinterpret_cast<const float*>(rec + 2 + T);
      const float* type_prefix = reinterpret_cast<const float*>(T == 1 ? rec + 3 : rec + mw);
      const uint64_t my_node = row >= 0 ? a.roots[r] : 0;
      for (int32_t set = 0; set < a.n_sets; ++set) {
        const int32_t k = a.set_k[set];
        const int32_t* et = a.et + a.set_off[set];
        const int32_t mode = TypeModeOf(k, T);
        const uint32_t call = a.call_id + (uint32_t)set;
        uint64_t id = (uint64_t)a.default_node;
        float w = 0.f;
        int32_t t = -1;
        bool draw = false;
        double u_nb = 0.0;
        if (row >= 0) {
          if (mode == kTypeSingle) {
            const int32_t tt = et[0];
            if (tt >= 0 && tt < T && type_end[tt] > (tt == 0 ? 0 : type_end[tt - 1])) {
              const Philox4 b = RngBlock(a.seed, call, kDomainNeighbor, my_node, ((uint32_t)j) >> 1);
              u_nb = (j & 1) ? UnitFromWords(b.w[2], b.w[3]) : UnitFromWords(b.w[0], b.w[1]);
              t = tt; draw = true;
            }
          } else {
            bool valid;
            if (mode == kTypeSub) {
              valid = true;
              for (int32_t i = 0; i < k; ++i) valid = valid && et[i] >= 0 && et[i] < T;
              if (valid) valid = SubTypeSum{type_prefix, et}((uint64_t)(k - 1)) != 0.f;
            } else {
              valid = type_prefix[T - 1] != 0.f;
            }
            if (valid) {
              const Philox4 b = RngBlock(a.seed, call, kDomainNeighbor, my_node, (uint32_t)j);
              const double u_type = UnitFromWords(b.w[0], b.w[1]);
              u_nb = UnitFromWords(b.w[2], b.w[3]);
              if (mode == kTypeSub) {
                t = et[RandomSelectT(SubTypeSum{type_prefix, et}, 0, (uint64_t)(k - 1), u_type)];
              } else {
                t = (int32_t)RandomSelect(type_prefix, 0, (uint64_t)(T - 1), u_type);
              }
              if (type_end[t] - 1 < (t == 0 ? 0 : type_end[t - 1])) { id = 0; w = 0.f; t = 0; }   // as SampleAt
              else draw = true;
            }
          }
        }
        if (draw) {
          Segment sg;
          sg.row_ptr = (int64_t)rec[1];
          sg.b = t == 0 ? 0 : type_end[t - 1];
          sg.e = type_end[t] - 1;
          sg.lo = sg.row_ptr + sg.b; sg.hi = sg.row_ptr + sg.e;
          sg.wb_lo = rec[0];
          sg.row_deg = (uint32_t)type_end[T - 1];
          sg.row_total = lim[T - 1];
          sg.limit_end = lim[t];
          sg.limit_begin = t == 0 ? 0.f : lim[t - 1];
          BlockPivotSample(a.g, sg, u_nb, &id, &w);
        }
        SetsStore(a, set, r, j, id, w, t);
      }
    }
    __syncthreads();            // the records are rewritten by the next round
  }
}

// The kernel for a call the single-type pivot kernels (sample_kernels.hip) do not serve:
// type draws on the block pivots where the graph allows it, the reference loop otherwise
// (grid = workgroups for one sample per lane).
int LaunchK1Variant(const euler_gpu_graph* g, hipStream_t stream, const SampleNbArgs& a,
                    int grid) {
  const int32_t layout = a.layout, k = a.k;
  const int block = 256;
  const bool tf_zero = layout == EULER_GPU_LAYOUT_TF && g->view.has_zero_nbr != 0;
  if (g_k1_variant == 6 && g_k1_typed_pivot != 0 && k != 1 && g->view.monotone &&
      !tf_zero && HasBlockSearch(a.g)) {
    auto kern = layout == EULER_GPU_LAYOUT_TF ? SampleNeighborTypedPivotKernel<true>
                                              : SampleNeighborTypedPivotKernel<false>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, stream, a);
  } else {
    hipLaunchKernelGGL(SampleNeighborKernel, dim3(grid), dim3(block), 0, stream, a);
  }
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

// n_sets edge-type sets over the same roots; false = this graph / tuning takes the separate calls
bool LaunchSampleNeighborSets(const euler_gpu_graph* g, hipStream_t stream, uint64_t seed,
                              uint32_t call_id, const uint64_t* roots, int64_t n,
                              const int32_t* edge_types, const int32_t* set_k, int32_t n_sets,
                              int32_t count, int64_t default_node, uint64_t* out_id, float* out_w,
                              int32_t* out_t, int* rc_out, int32_t* packed) {
  *rc_out = EULER_GPU_OK;
  if (g_k1_variant != 6 || g_k1_typed_pivot == 0 || !g->view.monotone || g->view.has_zero_nbr != 0 ||
      n_sets > kMaxTypeSets)
    return false;
  int32_t total_k = 0;
  for (int32_t s = 0; s < n_sets; ++s) total_k += set_k[s];
  if (total_k > kMaxListedTypes) return false;
  SampleSetsArgs a{};
  *rc_out = SamplingView(g, &a.g);
  if (*rc_out != EULER_GPU_OK) return true;
  a.seed = seed; a.call_id = call_id; a.roots = roots; a.n = n; a.count = count;
  a.default_node = default_node; a.n_sets = n_sets;
  a.out_id = out_id; a.out_w = out_w; a.out_t = out_t;
  a.packed = packed;
  int32_t off = 0;
  int64_t poff = 0;
  for (int32_t s = 0; s < n_sets; ++s) {
    a.packed_off[s] = poff;
    poff += n * (int64_t)PackedWords(count, set_k[s] == 1 ? 0 : 1);
    a.set_k[s] = set_k[s]; a.set_off[s] = off;
    for (int32_t i = 0; i < set_k[s]; ++i) a.et[off + i] = edge_types[off + i];
    off += set_k[s];
  }
  const int64_t total = n * (int64_t)count * n_sets;
  const int32_t rpb = count <= 256 ? 256 / count : 0;
  const size_t lds = (size_t)rpb * (8 + 4 * (2 + 3 * (size_t)a.g.T));
  if (a.g.wbg != nullptr && a.g.wb != nullptr && rpb > 0 && lds <= 48 * 1024 && g_k1_sets_lds != 0) {
    int64_t blocks = (n + rpb - 1) / rpb;
    if (blocks > kK1GridCap) blocks = kK1GridCap;
    // (key 47 = 2: the build for 6 waves per SIMD - 80 registers, nothing spilled; 1: 8 waves, 72 bytes
    // of scratch per lane)
    auto kern = g_k1_sets_lds == 2 ? SampleNeighborSetsLdsKernel<6> : SampleNeighborSetsLdsKernel<8>;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, a, rpb);
    if (hipGetLastError() != hipSuccess) *rc_out = Fail(EULER_GPU_EHIP, "sample_neighbor_sets: launch failed");
    return true;
  }
  int64_t blocks = (total + 255) / 256;
  if (blocks > kK1GridCap) blocks = kK1GridCap;
  hipLaunchKernelGGL(SampleNeighborSetsKernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
  if (hipGetLastError() != hipSuccess) *rc_out = Fail(EULER_GPU_EHIP, "sample_neighbor_sets: launch failed");
  return true;
}

}  // namespace euler_gpu
