// The default search of sample_neighbor on gfx950: pivot levels over the flat
// running sums and block pivots over the 128-byte EdgeBlock index (common.h).
// Device code shared by the sampling kernels (sample_kernels.hip) and the
// random walk (walk_kernels.hip).
#ifndef EULER_AMD_CSRC_K1_SEARCH_H_
#define EULER_AMD_CSRC_K1_SEARCH_H_

#include <hip/hip_runtime.h>

#include "device_fns.h"
#include "wb_index.h"

namespace euler_gpu {

// running sum of edge m out of its EdgeBlock
__device__ __forceinline__ float BlockedPw(const GraphView& g, int64_t m) {
  // (a graph whose EdgeBlocks were declined - graph_build.hip: EnsureBlockedIndex - and that has
  // no weight-bucket index either: the flat sums hold the same value)
  if (g.blk == nullptr) return g.prefix_w[m];
  const int64_t bi = m / kEdgesPerBlock;
  return g.blk[bi].pw[(int32_t)(m - bi * kEdgesPerBlock)];
}

// wave-level barrier with LDS visibility (lanes of one wave exchanging staged data)
__device__ __forceinline__ void WaveSync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------
// K1 pivot path (default).
//
// tools/k1_phases.py (s_memtime stamps around every phase of the flat kernel on
// the metric workload) shows that each vector-memory round trip of a wave -
// the coalesced root-id load as much as a divergent probe - takes the same
// ~3000 ticks: the CU's memory pipeline is a queue, and the kernel's time is
// (vector-memory instructions per wave) x (queue service time).  Fewer bytes
// (blocked index), more loads in flight (ILP) or whole-line staging in LDS do
// not change it; fewer memory INSTRUCTIONS do.  A binary search spends
// ceil(log2 deg) of them; this kernel spends ~log5(deg):
//
//   level 1 entry q = nw[4q+3]; level k+1 entry q = level k entry 5q+4.
//   The answer's possible positions at level k are [lo/D_k, hi/D_k] (D_1 = 4,
//   D_k = 4*5^(k-1)).  Start at the first level K where that range has <= 4
//   candidates, and walk down: each step is ONE unaligned 16-byte load of the
//   <= 4 candidate entries below the chosen entry ("found" entries bound their
//   last child, so a step never needs a fifth key).  The leaf window is
//   shifted by one so that it also holds nw[m-1]; nw[m] is a loaded key, the
//   bounding key carried down, or the segment's limit.  The result index is
//   the first m of [b, e] with nw[m] > r, i.e. RandomSelect's answer on
//   non-decreasing rows; Q3 lanes replay the reference loop.
// ------------------------------------------------------------------------
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ float Pick4(const float4u& v, int32_t i) {
  return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
}

struct Segment {
  int64_t row_ptr;      // first edge of the row
  int64_t lo, hi;       // searched edges [lo, hi] (global indices)
  float limit_begin, limit_end;
  int32_t b, e;         // the same segment, row-relative
  // weight-bucket index (g.wbg != nullptr): the row's first block, its edges, its total
  uint32_t wb_lo, row_deg;
  float row_total;
};

// The row's weight-bucket record (common.h: GraphView::wbg): first block, total, and the
// limits of the edge-type group [b, e] - out of ONE small record (the pivot path reads the
// limits from two block lines, a dependent trip).
__device__ __forceinline__ void LoadWbSegment(const GraphView& g, int64_t row, int32_t t,
                                              int32_t row_deg, Segment* sg) {
  const uint8_t* rec = g.wbg + row * (int64_t)g.wbg_stride;
  const float* lim = reinterpret_cast<const float*>(rec + 8 + 4 * g.T);
  sg->wb_lo = *reinterpret_cast<const uint32_t*>(rec);
  sg->row_deg = (uint32_t)row_deg;
  sg->row_total = lim[g.T - 1];
  sg->limit_end = lim[t];
  sg->limit_begin = t == 0 ? 0.f : lim[t - 1];
}

// The row record as the typed kernels read it (first edge, group ends, type sums) - out of the
// weight-bucket record when the graph has one with type sums (T > 1): the same line the
// segment's limits come from, instead of the row record's line beside it.
__device__ __forceinline__ RowMeta LoadRowMetaWb(const GraphView& g, int64_t row) {
  if (g.wbg == nullptr || g.T == 1) return LoadRowMeta(g, row);
  const uint8_t* rec = g.wbg + row * (int64_t)g.wbg_stride;
  RowMeta m;
  m.row_ptr = (int64_t)reinterpret_cast<const uint32_t*>(rec)[1];
  m.type_end = reinterpret_cast<const int32_t*>(rec + 8);
  m.type_prefix = reinterpret_cast<const float*>(rec + 8 + 8 * g.T);
  return m;
}

// ... and the whole segment of listed type t from that record alone: the row record is not
// read at all (false: the row has no edge of the type)
__device__ __forceinline__ bool LoadWbSegmentOnly(const GraphView& g, int64_t row, int32_t t,
                                                  Segment* sg) {
  const uint8_t* rec = g.wbg + row * (int64_t)g.wbg_stride;
  const int32_t* te = reinterpret_cast<const int32_t*>(rec + 8);
  const float* lim = reinterpret_cast<const float*>(rec + 8 + 4 * g.T);
  if (g.T == 1) {                        // one 16-byte load
    const uint4 q = *reinterpret_cast<const uint4*>(rec);
    sg->wb_lo = q.x; sg->row_ptr = (int64_t)q.y;
    sg->b = 0; sg->e = (int32_t)q.z - 1;
    sg->row_deg = q.z;
    sg->row_total = __uint_as_float(q.w);
    sg->limit_begin = 0.f; sg->limit_end = sg->row_total;
  } else {
    sg->wb_lo = reinterpret_cast<const uint32_t*>(rec)[0];
    sg->row_ptr = (int64_t)reinterpret_cast<const uint32_t*>(rec)[1];
    sg->b = t == 0 ? 0 : te[t - 1];
    sg->e = te[t] - 1;
    sg->row_deg = (uint32_t)te[g.T - 1];
    sg->row_total = lim[g.T - 1];
    sg->limit_end = lim[t];
    sg->limit_begin = t == 0 ? 0.f : lim[t - 1];
  }
  sg->lo = sg->row_ptr + sg->b;
  sg->hi = sg->row_ptr + sg->e;
  return sg->e >= sg->b;
}

// One draw u on a segment: the neighbour RandomSelect picks and its weight.
__device__ __forceinline__ void PivotSample(const GraphView& g, const Segment& sg,
                                            double u, uint64_t* id, float* w) {
  const float* __restrict__ A0 = g.prefix_w;
  const int64_t lo = sg.lo, hi = sg.hi;
  const double rr = ScaleDraw(u, sg.limit_begin, sg.limit_end);
  if (!((double)sg.limit_end > rr)) {
    // Q3: r rounded up to the end of the segment - replay the reference
    const float* nw = A0 + sg.row_ptr;
    const int32_t m = (int32_t)RandomSelect(nw, (uint64_t)sg.b, (uint64_t)sg.e, u);
    *id = g.nbr[sg.row_ptr + m];
    *w = __fsub_rn(nw[m], m == 0 ? 0.f : nw[m - 1]);
    return;
  }
  if (g.uniform_w) {
    // H1 (uniform weights, configs[1]): nw[m] = m + 1 exactly, so the first m with
    // nw[m] > r is floor(r) (r < limit_end was just checked; r >= limit_begin = b)
    // and the weight nw[m] - nw[m-1] is 1.0f
    *id = g.nbr[sg.row_ptr + (int64_t)rr];
    *w = 1.0f;
    return;
  }
  // candidate ranges of every level; K = first level with <= 4 of them
  uint32_t l[kPivotLevels + 1], h[kPivotLevels + 1];
  l[1] = (uint32_t)(lo >> 2);
  h[1] = (uint32_t)(hi >> 2);
#pragma unroll
  for (int k = 2; k <= kPivotLevels; ++k) { l[k] = l[k - 1] / 5u; h[k] = h[k - 1] / 5u; }
  int32_t K = 0;
  if (hi - lo > 3) {
    K = kPivotLevels + 1;
#pragma unroll
    for (int k = kPivotLevels; k >= 1; --k)
      if (h[k] - l[k] <= 4u) K = k;
  }
  if (K > kPivotLevels) {
    // rows beyond the pivot levels' reach: plain upper-bound search
    int64_t lo2 = lo, hi2 = hi;
    while (lo2 < hi2) {
      const int64_t mid = (lo2 + hi2) >> 1;
      if ((double)A0[mid] > rr) hi2 = mid; else lo2 = mid + 1;
    }
    *id = g.nbr[lo2];
    *w = __fsub_rn(A0[lo2], lo2 == sg.row_ptr ? 0.f : A0[lo2 - 1]);
    return;
  }
  uint32_t x = 0;        // chosen entry of the level above
  bool found = false;    // its key was compared (> r): it bounds its children
  float kv = 0.f;        // that key
#pragma unroll
  for (int k = kPivotLevels; k >= 1; --k) {
    if (k <= K) {
      uint32_t c_lo, c_hi;
      if (k == K) { c_lo = l[k]; c_hi = h[k]; }
      else {
        c_lo = max(l[k], 5u * x);
        c_hi = found ? 5u * x + 4u : h[k];
      }
      const int32_t cnt = (int32_t)(c_hi - c_lo);
      const float4u kw =
          *reinterpret_cast<const float4u*>(g.pivots + g.piv_off[k] + c_lo);
      int32_t pos = 0;
      pos += (0 < cnt && !((double)kw.x > rr)) ? 1 : 0;
      pos += (1 < cnt && !((double)kw.y > rr)) ? 1 : 0;
      pos += (2 < cnt && !((double)kw.z > rr)) ? 1 : 0;
      pos += (3 < cnt && !((double)kw.w > rr)) ? 1 : 0;
      x = c_lo + (uint32_t)pos;
      if (pos < cnt) { found = true; kv = Pick4(kw, pos); }
    }
  }
  // leaf: candidates among the flat elements below level-1 entry x
  int64_t c_lo = lo, c_hi = hi;
  if (K >= 1) {
    c_lo = max(lo, (int64_t)x * 4);
    c_hi = found ? (int64_t)x * 4 + 3 : hi;
  }
  const int32_t cnt = (int32_t)(c_hi - c_lo);       // <= 3
  int64_t ws = c_lo - 1;                            // window start
  if (ws > g.n_edges - 4) ws = g.n_edges - 4;
  if (ws < 0) ws = 0;
  const int32_t sh = (int32_t)(c_lo - ws);          // key i sits at sh + i
  const float4u wv = *reinterpret_cast<const float4u*>(A0 + ws);
  int32_t pos = 0;
  pos += (0 < cnt && !((double)Pick4(wv, sh) > rr)) ? 1 : 0;
  pos += (1 < cnt && !((double)Pick4(wv, sh + 1) > rr)) ? 1 : 0;
  pos += (2 < cnt && !((double)Pick4(wv, sh + 2) > rr)) ? 1 : 0;
  const int64_t m = c_lo + pos;
  const float nw_m = pos < cnt ? Pick4(wv, sh + pos) : (found ? kv : sg.limit_end);
  // `mid ? nw[mid-1] : 0` is row-relative
  const float prev = m == sg.row_ptr ? 0.f : Pick4(wv, sh + pos - 1);
  *id = g.nbr[m];
  *w = __fsub_rn(nw_m, prev);
}

// ------------------------------------------------------------------------
// Block-pivot search (K1 variant 6).  Over the distinct roots of a dedup'ed
// hop every row is cold and the launch runs at the chip's random-line rate
// (46 of ~54 G L2 misses/s), touching ~3 cold lines per sample: a level-1
// pivot window, the leaf window of prefix_w, and the id in nbr.  Here the
// pivots index 128-byte EdgeBlocks (10 edges: sums + ids + the previous block's
// last sum in ONE line) instead of 4-element groups of the flat array: level 1
// = one entry per block (skip1), level k+1 entry q = level k entry 5q+4.  A
// sample then touches a level-1 window (a row of degree d has d/320 lines of
// them, shared by its samples) and one block line.
// Same contract as PivotSample: first m in [lo, hi] with nw[m] > r.
// ------------------------------------------------------------------------
// USE_WB = false: the pivot levels whatever the view holds (the lean kernels' second chance for a
// draw its weight-bucket block did not bracket).  m_out: the flat index of the drawn edge.
template <bool USE_WB = true>
__device__ __forceinline__ void BlockPivotSample(const GraphView& g, const Segment& sg,
                                                 double u, uint64_t* id, float* w,
                                                 int64_t* m_out = nullptr) {
  const int64_t lo = sg.lo, hi = sg.hi;
  const double rr = ScaleDraw(u, sg.limit_begin, sg.limit_end);
  if (!((double)sg.limit_end > rr)) {
    // Q3: r rounded up to the end of the segment - replay the reference
    const float* nw = g.prefix_w + sg.row_ptr;
    const int32_t m = (int32_t)RandomSelect(nw, (uint64_t)sg.b, (uint64_t)sg.e, u);
    *id = g.nbr[sg.row_ptr + m];
    *w = __fsub_rn(nw[m], m == 0 ? 0.f : nw[m - 1]);
    if (m_out != nullptr) *m_out = sg.row_ptr + m;
    return;
  }
  if (g.uniform_w) {
    // H1 (uniform weights, configs[1]): nw[m] = m + 1 exactly, so the first m with
    // nw[m] > r is floor(r) (r < limit_end was just checked; r >= limit_begin = b)
    // and the weight nw[m] - nw[m-1] is 1.0f
    *id = g.nbr[sg.row_ptr + (int64_t)rr];
    *w = 1.0f;
    if (m_out != nullptr) *m_out = sg.row_ptr + (int64_t)rr;
    return;
  }
  if (USE_WB && g.wbg != nullptr) {
    // weight-bucket index (wb_index.h): the bucket of r in the ROW's range names one
    // 128-byte line; its keys decide.  The first m of the row with nw[m] > r lies in
    // [lo, hi] because limit_begin <= r < limit_end and the sums do not decrease.
    const float f = WbFloorToFloat(rr);
    const uint32_t nbk = WbBuckets(sg.row_deg);
    const uint32_t j = nbk <= 1u ? 0u : WbBucketOf(f, nbk, WbScale(nbk, sg.row_total));
    const EdgeBlock* wbk = g.wb + sg.wb_lo + j;
    // (requesting the line's ids beside its keys - 8 loads instead of 3 + 1 dependent one -
    // was measured and is SLOWER: the CU's memory pipe is bound by load instructions x
    // divergent lanes, not by the dependent trip; profiles/r4_ab_wb_full.txt)
    uint32_t m = 0;
    const int32_t i = WbDraw(wbk, f, w, &m);
    if (i >= 0) {
      *id = wbk->nbr[i];
      if (m_out != nullptr) *m_out = (int64_t)m;
      return;
    }
    // the block does not bracket r - a row whose weights are far from even (i.i.d. uniform:
    // 1e-4 of the draws; lognormal sigma 2 or Pareto alpha 0.7: ~5 %): the pivot levels
    // below find it in ~log5(deg / 10) steps, not the bisection's log2(deg)
  }
  if (g.blk == nullptr) {
    // a graph served by the weight-bucket index alone (common.h: HasBlockSearch; at most 2
    // buckets in a thousand overflow): the missed draw bisects the flat running sums - the
    // first m of [lo, hi] with nw[m] > r, as the levels would find
    const float* __restrict__ nw = g.prefix_w + sg.row_ptr;      // row-relative, 32-bit positions
    uint32_t a = (uint32_t)sg.b, b = (uint32_t)sg.e;
    while (a < b) {
      const uint32_t mid = (a + b) >> 1;
      if ((double)nw[mid] > rr) b = mid; else a = mid + 1u;
    }
    *id = g.nbr[sg.row_ptr + a];
    *w = __fsub_rn(nw[a], a == 0u ? 0.f : nw[a - 1u]);
    if (m_out != nullptr) *m_out = sg.row_ptr + (int64_t)a;
    return;
  }
  // ranges of the levels, bottom up, only as far as needed: K = first level
  // with <= 4 candidates (most rows stop at level 1 or 2, and a wave whose
  // lanes have all stopped skips the remaining divisions)
  uint32_t l[kPivotLevels + 1], h[kPivotLevels + 1];
  l[1] = (uint32_t)(lo / kEdgesPerBlock);
  h[1] = (uint32_t)(hi / kEdgesPerBlock);
  int32_t K = 0;                     // 0: the segment lies inside one block
  if (h[1] != l[1]) {
    K = h[1] - l[1] <= 4u ? 1 : kPivotLevels + 1;
#pragma unroll
    for (int k = 2; k <= kPivotLevels; ++k) {
      l[k] = 0; h[k] = 0;
      if (K > kPivotLevels) {
        l[k] = l[k - 1] / 5u;
        h[k] = h[k - 1] / 5u;
        if (h[k] - l[k] <= 4u) K = k;
      }
    }
  }
  uint32_t x = l[1];
  bool found = false;
  if (K > kPivotLevels) {
    // beyond the levels' reach: bisect the block entries
    uint32_t a = l[1], b = h[1];
    while (a < b) {
      const uint32_t mid = (a + b) >> 1;
      if ((double)g.skip1[mid] > rr) b = mid; else a = mid + 1;
    }
    x = a;
    found = a < h[1];
  } else {
#pragma unroll
    for (int k = kPivotLevels; k >= 1; --k) {
      if (k <= K) {
        uint32_t c_lo, c_hi;
        if (k == K) { c_lo = l[k]; c_hi = h[k]; }
        else {
          c_lo = max(l[k], 5u * x);
          c_hi = found ? 5u * x + 4u : h[k];
        }
        const int32_t cnt = (int32_t)(c_hi - c_lo);
        const float* lvl = k == 1 ? g.skip1 : g.bpiv + g.bpiv_off[k];
        const float4u kw = *reinterpret_cast<const float4u*>(lvl + c_lo);
        int32_t pos = 0;
        pos += (0 < cnt && !((double)kw.x > rr)) ? 1 : 0;
        pos += (1 < cnt && !((double)kw.y > rr)) ? 1 : 0;
        pos += (2 < cnt && !((double)kw.z > rr)) ? 1 : 0;
        pos += (3 < cnt && !((double)kw.w > rr)) ? 1 : 0;
        x = c_lo + (uint32_t)pos;
        if (pos < cnt) found = true;
      }
    }
  }
  // leaf: block x holds the answer (its last sum exceeds r when `found`,
  // otherwise it is the block of hi, whose sum exceeds r)
  const EdgeBlock* bk = g.blk + x;
  const int64_t base = (int64_t)x * kEdgesPerBlock;
  const int32_t i_lo = lo > base ? (int32_t)(lo - base) : 0;
  const int32_t i_hi = found ? kEdgesPerBlock - 1 : (int32_t)(hi - base);   // inclusive
  const float4 a0 = *reinterpret_cast<const float4*>(bk->pw);
  const float4 a1 = *reinterpret_cast<const float4*>(bk->pw + 4);
  const float4 a2 = *reinterpret_cast<const float4*>(bk->pw + 8);   // pw[8], pw[9], prev_last, pad
  const float v[10] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y};
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
  if (base + i == sg.row_ptr) prev = 0.f;          // `mid ? nw[mid-1] : 0`, row-relative
  *id = bk->nbr[i];
  *w = __fsub_rn(nw_m, prev);
  if (m_out != nullptr) *m_out = base + i;
}

// Row record -> searched segment of the listed type; false = empty / invalid
// (node.cc:127-136).
template <bool BLOCKED = false>
__device__ __forceinline__ bool LoadSegment(const GraphView& g, int64_t row,
                                            int32_t t, Segment* sg) {
  if (row < 0 || t < 0 || t >= g.T) return false;
  if (BLOCKED && g.wbg != nullptr) return LoadWbSegmentOnly(g, row, t, sg);
  const uint8_t* rec = g.row_meta + row * (int64_t)g.meta_stride;
  if (g.T == 1) {
    const uint4 q = *reinterpret_cast<const uint4*>(rec);
    sg->row_ptr = (int64_t)(((uint64_t)q.y << 32) | q.x);
    sg->b = 0;
    sg->e = (int32_t)q.z - 1;
    if (g.total_in_meta) {
      // the record's type sum IS the row's last running sum (verified at build):
      // one dependent load less per root
      if (sg->e < 0) return false;
      sg->lo = sg->row_ptr;
      sg->hi = sg->row_ptr + sg->e;
      sg->limit_begin = 0.f;
      sg->limit_end = __uint_as_float(q.w);
      return true;
    }
  } else {
    sg->row_ptr = *reinterpret_cast<const int64_t*>(rec);
    const int32_t* te = reinterpret_cast<const int32_t*>(rec + 8);
    sg->b = t == 0 ? 0 : te[t - 1];
    sg->e = te[t] - 1;
  }
  if (sg->e < sg->b) return false;
  sg->lo = sg->row_ptr + sg->b;
  sg->hi = sg->row_ptr + sg->e;
  if (BLOCKED) {     // same values, read from the block lines the search will touch
    sg->limit_end = BlockedPw(g, sg->hi);
    sg->limit_begin = sg->b == 0 ? 0.f : BlockedPw(g, sg->lo - 1);
  } else {
    sg->limit_end = g.prefix_w[sg->hi];
    sg->limit_begin = sg->b == 0 ? 0.f : g.prefix_w[sg->lo - 1];
  }
  return true;
}

}  // namespace euler_gpu

#endif  // EULER_AMD_CSRC_K1_SEARCH_H_
