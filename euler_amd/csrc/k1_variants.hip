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
// Round 3: the typed launches of the hetero workload are 2.5 rounds of waves on a chain of
// dependent loads (root -> record -> 3-4 probes of the type draw, each an L1 round trip ->
// the type's bounds -> its two limits -> levels -> leaf -> id).  Two cuts:
//  * REG8 (graphs of <= 8 edge-type groups): the record comes in ONE trip - row_ptr and
//    all type_end / type_prefix entries as independent loads - and the type draw (the
//    reference's bisection over the type sums, or over the listed sub-collection's)
//    runs on registers;
//  * a segment that lies inside ONE EdgeBlock (the usual case for typed rows: a few edges
//    per type) takes its limits from the block's sums, which are the leaf's sums too:
//    block -> id instead of limits -> leaf -> id.
__device__ __forceinline__ float Sel8(const float (&v)[8], int32_t i) {
  const bool b0 = (i & 1) != 0, b1 = (i & 2) != 0, b2 = (i & 4) != 0;
  const float a = b0 ? v[1] : v[0], b = b0 ? v[3] : v[2], c = b0 ? v[5] : v[4], d = b0 ? v[7] : v[6];
  const float e = b1 ? b : a, f = b1 ? d : c;
  return b2 ? f : e;
}
__device__ __forceinline__ int32_t Sel8i(const int32_t (&v)[8], int32_t i) {
  const bool b0 = (i & 1) != 0, b1 = (i & 2) != 0, b2 = (i & 4) != 0;
  const int32_t a = b0 ? v[1] : v[0], b = b0 ? v[3] : v[2], c = b0 ? v[5] : v[4], d = b0 ? v[7] : v[6];
  const int32_t e = b1 ? b : a, f = b1 ? d : c;
  return b2 ? f : e;
}
struct RegTypeSum {              // edge_group_collection running sums out of registers
  const float (&tp)[8];
  __device__ __forceinline__ float operator()(uint64_t i) const { return Sel8(tp, (int32_t)i); }
};
struct RegSubTypeSum {           // SubTypeSum (device_fns.h) out of registers
  const float (&tp)[8];
  const int32_t* et;
  __device__ __forceinline__ float operator()(uint64_t i) const {
    float s = 0.f;
    for (uint64_t x = 0; x <= i; ++x) {
      const int32_t t = et[x];
      s = __fadd_rn(s, __fsub_rn(Sel8(tp, t), t > 0 ? Sel8(tp, t - 1) : 0.f));
    }
    return s;
  }
};

// One draw on a segment [lo, hi] that lies inside one EdgeBlock; false = Q3 (r rounded up
// to the segment's end): the caller takes the general search, which replays the reference.
__device__ __forceinline__ bool OneBlockSample(const GraphView& g, int64_t row_ptr, int64_t lo,
                                               int64_t hi, int32_t b_idx, double u, uint64_t* id,
                                               float* w) {
  const int64_t x = lo / kEdgesPerBlock;
  const EdgeBlock* bk = g.blk + x;
  const int64_t base = x * kEdgesPerBlock;
  const int32_t i_lo = (int32_t)(lo - base), i_hi = (int32_t)(hi - base);      // inclusive
  const float4 a0 = *reinterpret_cast<const float4*>(bk->pw);
  const float4 a1 = *reinterpret_cast<const float4*>(bk->pw + 4);
  const float4 a2 = *reinterpret_cast<const float4*>(bk->pw + 8);   // pw[8], pw[9], prev_last, pad
  const float v[10] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y};
  float limit_end = v[0], limit_begin = 0.f;
#pragma unroll
  for (int j = 0; j < kEdgesPerBlock; ++j) {
    if (j == i_hi) limit_end = v[j];
    if (j + 1 == i_lo) limit_begin = v[j];
  }
  if (i_lo == 0) limit_begin = a2.z;                 // the edge before the block
  if (b_idx == 0) limit_begin = 0.f;                 // the segment starts the row
  const double rr = ScaleDraw(u, limit_begin, limit_end);
  if (!((double)limit_end > rr)) return false;
  int32_t i = i_lo;
#pragma unroll
  for (int j = 0; j < kEdgesPerBlock - 1; ++j)
    i += (j >= i_lo && j < i_hi && !((double)v[j] > rr)) ? 1 : 0;
  float nw_m = v[0], prev = a2.z;
#pragma unroll
  for (int j = 0; j < kEdgesPerBlock; ++j) {
    if (j == i) nw_m = v[j];
    if (j + 1 == i) prev = v[j];
  }
  if (base + i == row_ptr) prev = 0.f;               // `mid ? nw[mid-1] : 0`, row-relative
  *id = bk->nbr[i];
  *w = __fsub_rn(nw_m, prev);
  return true;
}

template <bool TF_LAYOUT, bool REG8>
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
      const RowMeta m = LoadRowMeta(a.g, row);
      int32_t te[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      float tp[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (REG8) {
#pragma unroll
        for (int x = 0; x < 8; ++x)
          if (x < T) { te[x] = m.type_end[x]; tp[x] = m.type_prefix[x]; }
      }
      // node.cc:106-121,137-148: which rows have nothing to draw from
      if (mode == kTypeSub) {
        valid = true;
        for (int32_t i = 0; i < a.k; ++i) valid = valid && a.et[i] >= 0 && a.et[i] < T;
        if (valid) {
          if (REG8) valid = RegSubTypeSum{tp, a.et}((uint64_t)(a.k - 1)) != 0.f;
          else valid = SubTypeSum{m.type_prefix, a.et}((uint64_t)(a.k - 1)) != 0.f;
        }
      } else {
        valid = (REG8 ? Sel8(tp, T - 1) : m.type_prefix[T - 1]) != 0.f;
      }
      if (valid) {
        const Philox4 b = RngBlock(a.seed, a.call_id, kDomainNeighbor, node, (uint32_t)j);
        const double u_type = UnitFromWords(b.w[0], b.w[1]);
        const double u_nb = UnitFromWords(b.w[2], b.w[3]);
        if (mode == kTypeSub) {
          if (REG8) t = a.et[RandomSelectT(RegSubTypeSum{tp, a.et}, 0, (uint64_t)(a.k - 1), u_type)];
          else t = a.et[RandomSelectT(SubTypeSum{m.type_prefix, a.et}, 0, (uint64_t)(a.k - 1), u_type)];
        } else {
          if (REG8) t = (int32_t)RandomSelectT(RegTypeSum{tp}, 0, (uint64_t)(T - 1), u_type);
          else t = (int32_t)RandomSelect(m.type_prefix, 0, (uint64_t)(T - 1), u_type);
        }
        const int32_t b_idx = t == 0 ? 0 : (REG8 ? Sel8i(te, t - 1) : m.type_end[t - 1]);
        const int32_t e_idx = (REG8 ? Sel8i(te, t) : m.type_end[t]) - 1;
        if (e_idx < b_idx) {
          // an empty group is only reachable through an out-of-range read in the
          // reference: the sentinel, as SampleAt (device_fns.h)
          id = 0; w = 0.f; t = 0;
        } else {
          const int64_t lo = m.row_ptr + b_idx, hi = m.row_ptr + e_idx;
          bool done = false;
          if (lo / kEdgesPerBlock == hi / kEdgesPerBlock && a.g.uniform_w == 0)
            done = OneBlockSample(a.g, m.row_ptr, lo, hi, b_idx, u_nb, &id, &w);
          if (!done) {
            Segment sg;
            sg.row_ptr = m.row_ptr; sg.b = b_idx; sg.e = e_idx;
            sg.lo = lo; sg.hi = hi;
            sg.limit_end = BlockedPw(a.g, sg.hi);
            sg.limit_begin = b_idx == 0 ? 0.f : BlockedPw(a.g, sg.lo - 1);
            BlockPivotSample(a.g, sg, u_nb, &id, &w);
          }
        }
      }
    }
    a.out_id[s] = id;
    a.out_w[s] = w;
    a.out_t[s] = t;
    if (j == 0 && a.out_row_mask != nullptr) a.out_row_mask[r] = valid ? 0 : 1;
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
      !tf_zero && g->view.blk != nullptr) {
    const bool reg8 = g->view.T <= 8 && g_k1_typed_pivot != 2;
    auto kern = layout == EULER_GPU_LAYOUT_TF
                    ? (reg8 ? SampleNeighborTypedPivotKernel<true, true> : SampleNeighborTypedPivotKernel<true, false>)
                    : (reg8 ? SampleNeighborTypedPivotKernel<false, true> : SampleNeighborTypedPivotKernel<false, false>);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, stream, a);
  } else {
    hipLaunchKernelGGL(SampleNeighborKernel, dim3(grid), dim3(block), 0, stream, a);
  }
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

}  // namespace euler_gpu
