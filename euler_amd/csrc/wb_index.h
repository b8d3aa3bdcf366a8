// Weight-bucket index ("WB"): CDF inversion by direct address (gfx950).
//
// RandomSelect (common/compact_weighted_collection.h:30-52) finds, for a draw
// r = u * total of a row, the first edge m with nw[m] > r by bisection: log2(deg)
// dependent probes.  Rounds 1-3 replaced the bisection with fanout-5 pivot levels over
// 128-byte EdgeBlocks: log5(deg / 10) dependent 16-byte windows, each a cold line on a hub
// row, then the leaf line.  The answer only depends on WHERE r falls in [0, total) - so
// the row's range is cut into nbk equal buckets and bucket j gets its own 128-byte line
// holding the (up to) 10 consecutive edges that begin with the first edge reaching into
// the bucket:
//
//     j     = floor(f * nbk / total)              f = the largest float <= r
//     block = wb[wb_lo(row) + j]                  ONE cold line per draw, no levels
//     i     = #{k : pw[k] <= f}                   -> edge base + i, weight pw[i] - pw[i-1]
//
// nbk = 1 for rows of <= 10 edges, else ceil(deg / 4): a bucket is as wide as 4 average
// edges, so its answers span ~5 consecutive edges - the block's 10 hold them unless ten
// consecutive weights sum to less than four average ones (for i.i.d. uniform [0.5, 8)
// weights: 1e-4 of the buckets).  The keys decide, not the layout: a draw whose block does
// not bracket it (prev > f, or every key <= f) replays the reference's bisection over
// the flat arrays - any weight distribution gets RandomSelect's index, smooth ones get
// it with one line.  Memory: 128 B per 4 edges of a large row, one line per small row
// (42 GB for the 100M / 1B metric graph, beside the 28 GB of the flat arrays and EdgeBlocks
// the other kernels read).
//
// Everything here is __host__ __device__ per item: tests/csrc/host_check.hip runs the
// same source on the CPU against the oracle (`pytest -m "not gpu"`).
#ifndef EULER_AMD_CSRC_WB_INDEX_H_
#define EULER_AMD_CSRC_WB_INDEX_H_

#include "device_fns.h"

namespace euler_gpu {

constexpr uint32_t kWbShift = 2;      // a bucket per 4 edges of a large row

// 16-byte row record of the WB path: everything a draw needs before its leaf.  (The general
// record of common.h - {wb_lo, row_lo, type_end[T], lim[T]} - at T = 1: plain graphs keep ONE
// array and read it through either view.)
struct alignas(16) WbRec {
  uint32_t wb_lo;     // first block of the row in `wb`
  uint32_t lo;        // first edge in the flat arrays (cold path, edge numbers)
  uint32_t deg;       // edges of the row (single edge-type group)
  float total;        // last running sum of the row
};
static_assert(sizeof(WbRec) == 16, "WbRec must be 16 bytes");

EG_HD uint32_t WbBuckets(uint32_t deg) {
  return deg <= (uint32_t)kEdgesPerBlock ? (deg > 0u ? 1u : 0u)
                                         : (deg + ((1u << kWbShift) - 1u)) >> kWbShift;
}

// buckets per unit of running sum; the SAME float operations in the builder and in the
// sampler (one correctly rounded division)
EG_HD float WbScale(uint32_t nbk, float total) { return EG_FDIV((float)nbk, total); }

// bucket of the (rounded-down) draw f >= 0
EG_HD uint32_t WbBucketOf(float f, uint32_t nbk, float scale) {
  const float x = EG_FMUL(f, scale);
  // x < nbk fails for NaN too (denormal totals: scale = inf, f = 0): the last bucket - its
  // block then does not bracket the draw and the draw goes the cold way
  if (!(x < (float)nbk)) return nbk - 1u;
  return (uint32_t)x;
}

// First edge of bucket j's block: the first edge m of the row with nw[m] > L, where L is
// safely below every f that WbBucketOf maps to j (f * scale >= j up to one rounding of
// the product: f >= (j / scale) (1 - 2^-24); 2^-20 leaves room for the division's own).
// nw = running sums of the row (row-relative), deg > 0.  Bucket 0 starts at edge 0.
template <typename SumAt>
EG_HD uint32_t WbBlockStart(const SumAt& nw, uint32_t deg, uint32_t nbk, float scale,
                            uint32_t j) {
  if (j == 0u || nbk <= 1u) return 0u;
  const double L = ((double)j / (double)scale) * (1.0 - 1.0 / 1048576.0);
  if (!(L > 0.0)) return 0u;               // scale = inf / NaN: every block starts at 0
  uint32_t lo = 0u, hi = deg - 1u;         // answer in [lo, hi]; hi if no sum exceeds L
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if ((double)nw(mid) > L) hi = mid; else lo = mid + 1u;
  }
  return lo;
}

#if defined(__HIP_DEVICE_COMPILE__)
#define EG_POPC(x) __popc(x)
#else
#define EG_POPC(x) __builtin_popcount(x)
#endif

// Block of bucket j of a row (lo = first flat edge, deg > 0 edges, total = last running sum):
// up to 10 consecutive edges from the first one that reaches into the bucket; +inf keys and
// id 0 past the row's end; prev_last = the sum before the block's first edge (0 at the
// row's start: `mid ? nw[mid-1] : 0` is row-relative); pad = that edge's flat index.
// Returns true when the bucket OVERFLOWS its block: some draw that maps to bucket j has its
// answer beyond the block's ten edges (more than ten edges' intervals reach into the bucket -
// dust among giants) and will take the cold path.  The builder counts them: a graph where
// more than a few buckets in a thousand overflow keeps the pivot levels for its lean kernels.
EG_HD bool WbBuildBlock(const float* prefix_w, const uint64_t* nbr, uint32_t lo, uint32_t deg,
                        float total, uint32_t j, EdgeBlock* out) {
  const uint32_t nbk = WbBuckets(deg);
  const ArraySum nw{prefix_w + lo};
  const float scale = WbScale(nbk, total);
  const uint32_t s = WbBlockStart(nw, deg, nbk, scale, j);
  bool overflow = false;
  if (nbk > 1u && s + (uint32_t)kEdgesPerBlock < deg) {
    // the last edge a draw of this bucket can select: the first one whose sum exceeds the
    // bucket's upper bound (the row's last edge for the last bucket)
    if (j + 1u >= nbk) {
      overflow = true;                       // the row goes on past the block
    } else {
      const double U = ((double)(j + 1u) / (double)scale) * (1.0 + 1.0 / 1048576.0);
      overflow = !((double)nw(s + (uint32_t)kEdgesPerBlock - 1u) > U);
    }
  }
  for (int k = 0; k < kEdgesPerBlock; ++k) {
    const uint32_t m = s + (uint32_t)k;
    const bool in = m < deg;
    out->pw[k] = in ? prefix_w[lo + m] : __builtin_huge_valf();
    out->nbr[k] = in ? nbr[lo + m] : 0ull;
  }
  out->prev_last = s == 0u ? 0.f : prefix_w[lo + s - 1u];
  out->pad = lo + s;
  return overflow;
}

// One draw on a block: f = the largest float <= r, r < the row's total.  Returns the
// position i of the first key > f, its weight and flat edge index - or -1 when the block
// does not bracket f (first key's predecessor > f, or every key <= f): the caller replays
// the reference's search.  The three 16-byte loads are the whole line's key half; nw[i] and
// nw[i-1] come out of those registers (a second load of the line costs the CU's memory
// pipe more than a dozen selects cost a SIMD).
struct WbKeys { float4 a0, a1, a2; };     // pw[0..3], pw[4..7], {pw[8], pw[9], prev, base}

EG_HD WbKeys WbLoadKeys(const EdgeBlock* bk) {
  WbKeys k;
  k.a0 = *reinterpret_cast<const float4*>(bk->pw);
  k.a1 = *reinterpret_cast<const float4*>(bk->pw + 4);
  k.a2 = *reinterpret_cast<const float4*>(bk->pw + 8);
  return k;
}

EG_HD int32_t WbPickKeys(const WbKeys& k, float f, float* w_out, uint32_t* m_out) {
  const float4 a0 = k.a0, a1 = k.a1, a2 = k.a2;
  uint32_t le = 0;           // bit k = [pw[k] <= f]
  le = 2u * le + (!(a2.y > f) ? 1u : 0u);
  le = 2u * le + (!(a2.x > f) ? 1u : 0u);
  le = 2u * le + (!(a1.w > f) ? 1u : 0u);
  le = 2u * le + (!(a1.z > f) ? 1u : 0u);
  le = 2u * le + (!(a1.y > f) ? 1u : 0u);
  le = 2u * le + (!(a1.x > f) ? 1u : 0u);
  le = 2u * le + (!(a0.w > f) ? 1u : 0u);
  le = 2u * le + (!(a0.z > f) ? 1u : 0u);
  le = 2u * le + (!(a0.y > f) ? 1u : 0u);
  le = 2u * le + (!(a0.x > f) ? 1u : 0u);
  // the keys of a monotone row are non-decreasing: the set bits are a prefix
  const uint32_t i = (uint32_t)EG_POPC(le);
  if (i >= (uint32_t)kEdgesPerBlock || a2.z > f) return -1;
  const float v0 = a0.x, v1 = a0.y, v2 = a0.z, v3 = a0.w, v4 = a1.x, v5 = a1.y, v6 = a1.z,
              v7 = a1.w, v8 = a2.x, v9 = a2.y;
  const bool b0 = (i & 1u) != 0, b1 = (i & 2u) != 0, b2 = (i & 4u) != 0, b3 = (i & 8u) != 0;
  const float s01 = b0 ? v1 : v0, s23 = b0 ? v3 : v2, s45 = b0 ? v5 : v4, s67 = b0 ? v7 : v6,
              s89 = b0 ? v9 : v8;
  const float q03 = b1 ? s23 : s01, q47 = b1 ? s67 : s45;
  const float nw_m = b3 ? s89 : (b2 ? q47 : q03);
  // v[i - 1] for i >= 1, prev for i == 0: index (i + 15) & 15 over {v0 .. v8}, 15 = prev
  const uint32_t ip = (i + 15u) & 15u;
  const bool c0 = (ip & 1u) != 0, c1 = (ip & 2u) != 0, c2 = (ip & 4u) != 0, c3 = (ip & 8u) != 0;
  const float r01 = c0 ? v1 : v0, r23 = c0 ? v3 : v2, r45 = c0 ? v5 : v4, r67 = c0 ? v7 : v6;
  const float u03 = c1 ? r23 : r01, u47 = c1 ? r67 : r45;
  const float lowp = c2 ? u47 : u03;
  const float prev = c3 ? (ip == 8u ? v8 : a2.z) : lowp;
  *w_out = EG_FSUB(nw_m, prev);
  uint32_t base;
#if defined(__HIP_DEVICE_COMPILE__)
  base = __float_as_uint(a2.w);
#else
  __builtin_memcpy(&base, &a2.w, 4);
#endif
  *m_out = base + i;
  return (int32_t)i;
}

EG_HD int32_t WbDraw(const EdgeBlock* bk, float f, float* w_out, uint32_t* m_out) {
  return WbPickKeys(WbLoadKeys(bk), f, w_out, m_out);
}

// largest float <= r (r >= 0): for a float v, v > r <=> v > f - one conversion per draw
// instead of one per key
EG_HD float WbFloorToFloat(double r) {
  float f = (float)r;
  if ((double)f > r) {
#if defined(__HIP_DEVICE_COMPILE__)
    f = __uint_as_float(__float_as_uint(f) - 1u);
#else
    uint32_t b;
    __builtin_memcpy(&b, &f, 4);
    b -= 1u;
    __builtin_memcpy(&f, &b, 4);
#endif
  }
  return f;
}

// The hot part of one draw: r = u * total (the subtraction and the addition of the
// segment's zero begin are exact), its bucket's block, the key count.  false: the draw is
// cold - r rounded up to the row's total (Q3) or the block does not bracket it - and the
// caller replays RandomSelect over the flat arrays.
EG_HD bool WbSampleHot(const EdgeBlock* wb, const WbRec& rec, double u, uint64_t* id,
                       float* w, uint32_t* m) {
  const double r = EG_DMUL(u, (double)rec.total);
  if (!((double)rec.total > r)) return false;
  const float f = WbFloorToFloat(r);
  const uint32_t nbk = WbBuckets(rec.deg);
  const uint32_t j = nbk <= 1u ? 0u : WbBucketOf(f, nbk, WbScale(nbk, rec.total));
  const EdgeBlock* bk = wb + rec.wb_lo + j;
  const int32_t i = WbDraw(bk, f, w, m);
  if (i < 0) return false;
  *id = bk->nbr[i];
  return true;
}

}  // namespace euler_gpu

#endif  // EULER_AMD_CSRC_WB_INDEX_H_
