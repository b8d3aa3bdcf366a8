// The 2-hop SampleFanout of fanout_local.h rebuilt for ONE shape - the metric's: a plain graph
// (one edge-type group per node, strided identity id map, no neighbour id 0, weighted, served
// by the weight-bucket index) and one listed type per hop - and for instruction count.
//
// Same contract and the same algorithm as SampleFanoutLeanKernel<.., WB = 1> (reference:
// tf_euler/kernels/sample_fanout_op.cc:60-145 over Node::SampleNeighbor, core/graph/node.cc:
// 123-159; duplicates as parser/compiler.cc:76-90 resolves them): a wave owns `gr` roots, hop 1
// is a lane per PAIR of samples, duplicate children are found by drawn edge inside the wave,
// hop 2 runs once per distinct child, finished rows leave LDS as 16-byte stores.  What round 5's
// counters said about that kernel (profiles/r5_final_pmc_summary.json): 1 556 VALU + 611 SALU
// instructions per tile, 351 of the static instructions v_readlane / v_writelane (115 SGPRs of
// the by-value GraphView + option fields spilled), 22 % of the wave-cycles spent waiting for an
// issue slot and 37 % of every SIMD's cycles in VALU issue - the chain of cold loads is long
// BECAUSE the instructions between them queue behind other waves' instructions.  So here:
//   * the arguments are the 21 words this shape needs (nothing spilled);
//   * every lane's role - (root, pair) in hop 1, (slot, pair) in hop 2, (row, 16-byte chunk)
//     in the copy-out - is computed ONCE per wave, outside the tile loop: the loops advance by
//     whole rows, so no division and no modulo is left inside them (24-bit multiplies only);
//   * the slot pass works on the hop-1 lanes' own registers (one pass over 52 lanes instead of
//     two over 100 positions with the edge offsets parked in LDS);
//   * slot numbers are bytes, a slot's child is read through its representative sample (no
//     second id array): 5.3 KB of LDS per wave at 32 slots per pass (9-10 KB before);
//   * weights / types leave as 16-byte stores over PAIRS of rows (2 c2 floats = c2 / 2 chunks,
//     the chunk -> (row, column) pattern a per-lane constant).
// COOP = true (tuning key 54; NOT the shipped build): a draw's block is fetched by THREE lanes, 16 bytes
// each, as one coalesced request per line, staged in LDS and read back by the lane that owns the
// draw.  Why it was built (tools/ubench_block.hip, profiles/r6_ubench_block.txt): a lane that asks
// for its own line with three 16-byte loads and then a dependent 8-byte one - what every build
// does - makes FOUR requests per line, and the chip completes 19.5 G such lines/s at ANY residency
// (8 to 32 waves per CU): the rate the kernel's read side runs at however its geometry is tuned.
// One request per line: 48 G lines/s; lanes sharing a line + a dependent id request: 34-36 G.
// (What a request costs is its address translation: inside ~2 GiB three loads of a line cost what
// one costs - profiles/r6_ubench_block_footprint.txt)  In the kernel the staging
// gives the saving back: 0.237-0.244 ms against 0.222-0.229 (profiles/r6_sweep3_plain_coop.txt).
// LITE2 = true (key 57): hop 2 asks for two key chunks and the third only at a block's ends -
// 0.247 against 0.233 ms (profiles/r6_sweep4_plain_lite.txt).  Both stay as parity-tested variants.
// Bit-identical outputs (tests/test_gpu_parity.py runs every fanout test through both builds).
#ifndef EULER_AMD_CSRC_FANOUT_PLAIN_H_
#define EULER_AMD_CSRC_FANOUT_PLAIN_H_

#include <hip/hip_runtime.h>

#include "fanout_local.h"

namespace euler_gpu {

struct FanoutPlainArgs {
  const WbRec* wrec;
  const EdgeBlock* wb;
  const float* prefix_w;        // cold draws only (Q3, a bucket that overflows its block)
  const uint64_t* nbr;
  const uint64_t* roots;
  uint64_t* id1; float* w1; int32_t* ty1;
  uint64_t* id2; float* w2; int32_t* ty2;
  uint32_t* row_index;          // not null: the (unique rows, index) form
  const uint32_t* call_ids;     // several minibatches per launch (fanout_local.h: TileCallId)
  uint64_t seed, id_base, id_stride;
  int64_t n, n_rows, default_node, mb_n;
  uint32_t call_id, call_stride;
  int32_t c1, c2, gr, cap, wave_lds;
};

constexpr uint32_t kStageLines = 128;      // two blocks per lane
constexpr uint32_t kStageBytes = kStageLines * 48 + kStageLines * 4;   // keys of 128 blocks + their numbers
constexpr uint32_t kSkipLine = 0xFFFFFFFFu;

struct FanoutPlainLds {
  uint32_t o_stage, o_blk, o_sid, o_c1, o_mask, o_sw, o_w1, o_slot, o_rep, o_st, o_rvalid, bytes;
};
// coop: the key staging area [128][48 B] + block numbers [128] (cooperative fetch).  The hop-2
// results (sid / sw / st) may lie over the staged keys when a pass of hop 2 is ONE sampling step
// (cap <= the slots a step takes): the keys are in registers before a result is written.
__host__ __device__ inline FanoutPlainLds FanoutPlainLayout(int32_t gr, int32_t c1, int32_t c2,
                                                            int32_t cap, bool coop = false) {
  FanoutPlainLds L;
  const uint32_t p = (uint32_t)gr * (uint32_t)c1;
  const uint32_t s = (uint32_t)cap * (uint32_t)c2;
  const uint32_t hp2 = (uint32_t)c2 >> 1;
  const uint32_t rpi = hp2 <= 1u ? 64u : 64u / hp2;
  uint32_t o = 0;
  // what a tile keeps from hop 1 to its end
  L.o_c1 = o; o += (p + (p & 1)) * 8;       // u64 [gr][c1]   hop-1 ids (0 for a row without samples)
  L.o_w1 = o; o += ((p + 3) & ~3u) * 4;     // f32 [gr][c1]   (16-byte aligned)
  L.o_mask = o; o += (((uint32_t)gr + 1u) & ~1u) * 8;   // u64 [gr]  drawn edge offsets of a root
  L.o_slot = o; o += (p + 3) & ~3u;         // u8  [gr][c1]   slot of the sample's child
  L.o_rep = o; o += (p + 3) & ~3u;          // u8  [slots]    a sample that drew the slot's child
  L.o_rvalid = o; o += ((uint32_t)gr + 3) & ~3u;
  o = (o + 15) & ~15u;
  // the finished hop-2 rows of one pass ...
  const uint32_t o_res = o;
  L.o_sid = o; o += s * 8;                  // u64 [cap][c2]  ids
  L.o_sw = o; o += ((s + 3) & ~3u) * 4;     // f32 [cap][c2]  weights (16-byte aligned)
  L.o_st = o; o += ((uint32_t)cap + 3) & ~3u;    // i8 [cap]  type of the slot's row: 0, or -1 (no samples)
  o = (o + 15) & ~15u;
  // ... and the staged keys: over them when a pass is one sampling step, else behind them
  L.o_stage = 0; L.o_blk = 0;
  if (coop) {
    const uint32_t at = (uint32_t)cap <= rpi ? o_res : o;
    L.o_stage = at; L.o_blk = at + kStageLines * 48;
    if (at + kStageBytes > o) o = at + kStageBytes;
  }
  L.bytes = (o + 15) & ~15u;
  return L;
}

// n / d for n < 4096, 2 <= d <= 128 with a 24-bit multiply (full rate; v_mul_hi_u32 is not)
struct TinyDiv {
  uint32_t m;
  __host__ __device__ void Set(uint32_t d) { m = d <= 1u ? 0u : (1u << 20) / d + 1u; }
  __device__ __forceinline__ uint32_t operator()(uint32_t n) const {
    return m == 0u ? n : (__umul24(n, m) >> 20);
  }
};

// the two draws of one Philox block on one row through the weight-bucket index: WbSamplePair
// (fanout_local.h) over four pointers instead of a GraphView
__device__ __forceinline__ void PlainSamplePair(const FanoutPlainArgs& a, const WbRec rec,
                                                const bool live0, const bool live1, const double u0,
                                                const double u1, uint64_t id[2], float w[2],
                                                uint32_t m[2]) {
  const double r0 = __dmul_rn(u0, (double)rec.total), r1 = __dmul_rn(u1, (double)rec.total);
  bool cold0 = live0 && !((double)rec.total > r0);
  bool cold1 = live1 && !((double)rec.total > r1);
  const float f0 = WbFloorToFloat(r0), f1 = WbFloorToFloat(r1);
  const uint32_t nbk = WbBuckets(rec.deg);
  uint32_t j0 = 0u, j1 = 0u;
  if (nbk > 1u) {
    const float scale = WbScale(nbk, rec.total);
    j0 = WbBucketOf(f0, nbk, scale);
    j1 = WbBucketOf(f1, nbk, scale);
  }
  // (a dead lane's record is all zeros: block 0, a valid line nobody uses)
  const EdgeBlock* b0 = a.wb + rec.wb_lo + j0;
  const EdgeBlock* b1 = a.wb + rec.wb_lo + j1;
  const WbKeys k0 = WbLoadKeys(b0);
  const WbKeys k1 = WbLoadKeys(b1);
  id[0] = 0; id[1] = 0; w[0] = 0.f; w[1] = 0.f; m[0] = rec.lo; m[1] = rec.lo;
  const int32_t i0 = WbPickKeys(k0, f0, &w[0], &m[0]);
  const int32_t i1 = WbPickKeys(k1, f1, &w[1], &m[1]);
  const bool hot0 = live0 && !cold0 && i0 >= 0;
  const bool hot1 = live1 && !cold1 && i1 >= 0;
  if (hot0) id[0] = b0->nbr[i0];
  if (hot1) id[1] = b1->nbr[i1];
  cold0 = live0 && !hot0;
  cold1 = live1 && !hot1;
  if (__ballot(cold0 || cold1) != 0ull) {
    // the reference's own bisection over the flat running sums (RandomSelect,
    // common/compact_weighted_collection.h:30-52): right on every row, slow, rare
#pragma nounroll
    for (int s = 0; s < 2; ++s) {
      if (s == 0 ? cold0 : cold1) {
        const float* nw = a.prefix_w + rec.lo;
        const uint32_t mid = (uint32_t)RandomSelect(nw, 0, (uint64_t)(rec.deg - 1u), s == 0 ? u0 : u1);
        const uint64_t ci = a.nbr[rec.lo + mid];
        const float cw = __fsub_rn(nw[mid], mid == 0u ? 0.f : nw[mid - 1]);
        if (s == 0) { id[0] = ci; w[0] = cw; m[0] = rec.lo + mid; }
        else { id[1] = ci; w[1] = cw; m[1] = rec.lo + mid; }
      }
    }
  }
}

// The same two draws with the blocks' keys fetched COOPERATIVELY: every lane names its two
// blocks in LDS, three lanes fetch each block's key half (16 bytes each: one request per line
// instead of three per lane), the keys go through LDS to the lane that owns the draw; only the
// id of the drawn edge is a private (dependent) load.  All 64 lanes call; lanes [0, ntask) own
// draws (ntask wave-uniform).  A pair whose draws fall into one block names it once.
__device__ __forceinline__ void WaveSamplePairs(const FanoutPlainArgs& a, const uint32_t lane,
                                                const uint32_t ntask, float4* s_stage, uint32_t* s_blk,
                                                const WbRec rec, const bool live0, const bool live1,
                                                const double u0, const double u1, uint64_t id[2],
                                                float w[2], uint32_t m[2]) {
  const double r0 = __dmul_rn(u0, (double)rec.total), r1 = __dmul_rn(u1, (double)rec.total);
  bool cold0 = live0 && !((double)rec.total > r0);
  bool cold1 = live1 && !((double)rec.total > r1);
  const float f0 = WbFloorToFloat(r0), f1 = WbFloorToFloat(r1);
  const uint32_t nbk = WbBuckets(rec.deg);
  uint32_t j0 = 0u, j1 = 0u;
  if (nbk > 1u) {
    const float scale = WbScale(nbk, rec.total);
    j0 = WbBucketOf(f0, nbk, scale);
    j1 = WbBucketOf(f1, nbk, scale);
  }
  const bool same = j0 == j1;
  const uint32_t bi0 = rec.wb_lo + j0, bi1 = rec.wb_lo + j1;
  if (lane < ntask) {
    s_blk[2u * lane] = live0 ? bi0 : kSkipLine;
    s_blk[2u * lane + 1u] = (live1 && !same) ? bi1 : kSkipLine;
  }
  WaveSync();
  // ---- fetch: lane = (line fl of a group of 21, chunk fc); every load of the wave is issued
  // before the first is waited for
  const uint32_t nlines = 2u * ntask;
  const uint32_t fl = (lane * 21846u) >> 16, fc = lane - 3u * fl;      // lane / 3, lane % 3
  // LDS-DMA: the 16 bytes of lane L land at (step's base) + 16 L - and (line, chunk) = (21 it +
  // L / 3, L % 3) makes that the staged layout [line][3] itself; no register holds them
  {
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t*)s_stage;
#pragma unroll
    for (int it = 0; it < 7; ++it) {
      if ((uint32_t)it * 21u < nlines) {                                  // (wave-uniform)
        const uint32_t line = (uint32_t)it * 21u + fl;
        const uint32_t bi = (fl < 21u && line < nlines) ? s_blk[line] : kSkipLine;
        if (bi != kSkipLine) {
          const uint8_t* src = reinterpret_cast<const uint8_t*>(a.wb) + (size_t)bi * 128u + fc * 16u;
          const uint32_t dst = lds0 + (uint32_t)it * (63u * 16u);
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  WaveSync();
  WbKeys k0, k1;
  k0.a0 = s_stage[6u * lane]; k0.a1 = s_stage[6u * lane + 1u]; k0.a2 = s_stage[6u * lane + 2u];
  k1 = k0;
  if (!same) { k1.a0 = s_stage[6u * lane + 3u]; k1.a1 = s_stage[6u * lane + 4u]; k1.a2 = s_stage[6u * lane + 5u]; }
  id[0] = 0; id[1] = 0; w[0] = 0.f; w[1] = 0.f; m[0] = rec.lo; m[1] = rec.lo;
  const int32_t i0 = WbPickKeys(k0, f0, &w[0], &m[0]);
  const int32_t i1 = WbPickKeys(k1, f1, &w[1], &m[1]);
  const bool hot0 = live0 && !cold0 && i0 >= 0;
  const bool hot1 = live1 && !cold1 && i1 >= 0;
  if (hot0) id[0] = a.wb[bi0].nbr[i0];
  if (hot1) id[1] = a.wb[bi1].nbr[i1];
  cold0 = live0 && !hot0;
  cold1 = live1 && !hot1;
  if (__ballot(cold0 || cold1) != 0ull) {
#pragma nounroll
    for (int s = 0; s < 2; ++s) {
      if (s == 0 ? cold0 : cold1) {
        const float* nw = a.prefix_w + rec.lo;
        const uint32_t mid = (uint32_t)RandomSelect(nw, 0, (uint64_t)(rec.deg - 1u), s == 0 ? u0 : u1);
        const uint64_t ci = a.nbr[rec.lo + mid];
        const float cw = __fsub_rn(nw[mid], mid == 0u ? 0.f : nw[mid - 1]);
        if (s == 0) { id[0] = ci; w[0] = cw; m[0] = rec.lo + mid; }
        else { id[1] = ci; w[1] = cw; m[1] = rec.lo + mid; }
      }
    }
  }
}

// Hop 2 does not need the drawn edge's number, and most draws do not need the third key chunk:
// the answer i = #{keys <= f} is known from keys 0 .. 7 unless all eight are <= f, and for
// 1 <= i <= 7 both nw[i] and nw[i - 1] are among them and key 0 <= f proves the block brackets
// the draw from below.  So a draw asks for TWO key chunks, then - together with its id -
// for the third only when i == 0 (prev_last: the weight and the bracket check; the id is
// nbr[0] either way) or i == 8 (keys 8, 9 and the two candidate ids in the same trip):
// ~3.3 requests per line instead of 4 on a kernel whose read side is bound by requests
// (tools/ubench_block.hip).  LITE2 of the sampler below.
__device__ __forceinline__ float Sel8(const float4 a0, const float4 a1, const uint32_t i) {
  const bool b0 = (i & 1u) != 0, b1 = (i & 2u) != 0, b2 = (i & 4u) != 0;
  const float s01 = b0 ? a0.y : a0.x, s23 = b0 ? a0.w : a0.z, s45 = b0 ? a1.y : a1.x, s67 = b0 ? a1.w : a1.z;
  const float q03 = b1 ? s23 : s01, q47 = b1 ? s67 : s45;
  return b2 ? q47 : q03;
}

__device__ __forceinline__ void PlainSamplePairLite(const FanoutPlainArgs& a, const WbRec rec,
                                                    const bool live0, const bool live1, const double u0,
                                                    const double u1, uint64_t id[2], float w[2]) {
  const double r0 = __dmul_rn(u0, (double)rec.total), r1 = __dmul_rn(u1, (double)rec.total);
  bool cold0 = live0 && !((double)rec.total > r0);
  bool cold1 = live1 && !((double)rec.total > r1);
  const float f0 = WbFloorToFloat(r0), f1 = WbFloorToFloat(r1);
  const uint32_t nbk = WbBuckets(rec.deg);
  uint32_t j0 = 0u, j1 = 0u;
  if (nbk > 1u) {
    const float scale = WbScale(nbk, rec.total);
    j0 = WbBucketOf(f0, nbk, scale);
    j1 = WbBucketOf(f1, nbk, scale);
  }
  const EdgeBlock* b0 = a.wb + rec.wb_lo + j0;
  const EdgeBlock* b1 = a.wb + rec.wb_lo + j1;
  const float4 p0 = *reinterpret_cast<const float4*>(b0->pw), p1 = *reinterpret_cast<const float4*>(b0->pw + 4);
  const float4 q0 = *reinterpret_cast<const float4*>(b1->pw), q1 = *reinterpret_cast<const float4*>(b1->pw + 4);
  uint32_t i0 = 0, i1 = 0;
  i0 += !(p0.x > f0) ? 1u : 0u; i0 += !(p0.y > f0) ? 1u : 0u; i0 += !(p0.z > f0) ? 1u : 0u; i0 += !(p0.w > f0) ? 1u : 0u;
  i0 += !(p1.x > f0) ? 1u : 0u; i0 += !(p1.y > f0) ? 1u : 0u; i0 += !(p1.z > f0) ? 1u : 0u; i0 += !(p1.w > f0) ? 1u : 0u;
  i1 += !(q0.x > f1) ? 1u : 0u; i1 += !(q0.y > f1) ? 1u : 0u; i1 += !(q0.z > f1) ? 1u : 0u; i1 += !(q0.w > f1) ? 1u : 0u;
  i1 += !(q1.x > f1) ? 1u : 0u; i1 += !(q1.y > f1) ? 1u : 0u; i1 += !(q1.z > f1) ? 1u : 0u; i1 += !(q1.w > f1) ? 1u : 0u;
  const bool t0 = live0 && !cold0, t1 = live1 && !cold1;
  const bool edge0 = t0 && (i0 == 0u || i0 == 8u), edge1 = t1 && (i1 == 0u || i1 == 8u);
  // the second trip: the id (nbr[min(i, 8)]: for i == 8 the pair nbr[8], nbr[9]) and, at the
  // two ends, the third chunk
  fl_u64x2 n0, n1;
  n0.x = 0; n0.y = 0; n1 = n0;
  float4 p2 = make_float4(0.f, 0.f, 0.f, 0.f), q2 = p2;
  if (t0) {
    if (i0 == 8u) n0 = *reinterpret_cast<const fl_u64x2*>(b0->nbr + 8);
    else n0.x = b0->nbr[i0];
  }
  if (t1) {
    if (i1 == 8u) n1 = *reinterpret_cast<const fl_u64x2*>(b1->nbr + 8);
    else n1.x = b1->nbr[i1];
  }
  if (edge0) p2 = *reinterpret_cast<const float4*>(b0->pw + 8);
  if (edge1) q2 = *reinterpret_cast<const float4*>(b1->pw + 8);
  id[0] = 0; id[1] = 0; w[0] = 0.f; w[1] = 0.f;
  bool hot0 = t0, hot1 = t1;
  if (t0) {
    if (!edge0) { id[0] = n0.x; w[0] = __fsub_rn(Sel8(p0, p1, i0), Sel8(p0, p1, i0 - 1u)); }
    else if (i0 == 0u) { hot0 = !(p2.z > f0); id[0] = n0.x; w[0] = __fsub_rn(p0.x, p2.z); }
    else {              // all of keys 0 .. 7 <= f: the answer is edge 8 or 9 of the block, or beyond it
      const bool k8 = p2.x > f0, k9 = p2.y > f0;
      hot0 = k8 || k9;
      id[0] = k8 ? n0.x : n0.y;
      w[0] = k8 ? __fsub_rn(p2.x, p1.w) : __fsub_rn(p2.y, p2.x);
    }
  }
  if (t1) {
    if (!edge1) { id[1] = n1.x; w[1] = __fsub_rn(Sel8(q0, q1, i1), Sel8(q0, q1, i1 - 1u)); }
    else if (i1 == 0u) { hot1 = !(q2.z > f1); id[1] = n1.x; w[1] = __fsub_rn(q0.x, q2.z); }
    else {
      const bool k8 = q2.x > f1, k9 = q2.y > f1;
      hot1 = k8 || k9;
      id[1] = k8 ? n1.x : n1.y;
      w[1] = k8 ? __fsub_rn(q2.x, q1.w) : __fsub_rn(q2.y, q2.x);
    }
  }
  cold0 = live0 && !hot0;
  cold1 = live1 && !hot1;
  if (__ballot(cold0 || cold1) != 0ull) {
#pragma nounroll
    for (int s = 0; s < 2; ++s) {
      if (s == 0 ? cold0 : cold1) {
        const float* nw = a.prefix_w + rec.lo;
        const uint32_t mid = (uint32_t)RandomSelect(nw, 0, (uint64_t)(rec.deg - 1u), s == 0 ? u0 : u1);
        const uint64_t ci = a.nbr[rec.lo + mid];
        const float cw = __fsub_rn(nw[mid], mid == 0u ? 0.f : nw[mid - 1]);
        if (s == 0) { id[0] = ci; w[0] = cw; }
        else { id[1] = ci; w[1] = cw; }
      }
    }
  }
}

__device__ __forceinline__ uint32_t OpaqueLane(uint32_t lane) {
  asm volatile("" : "+v"(lane));
  return lane;
}

__device__ __forceinline__ int64_t PlainFindRow(const FanoutPlainArgs& a, const uint64_t id) {
  const uint64_t d = id - a.id_base;
  if (id < a.id_base) return -1;
  if (a.id_stride == 1) return d < (uint64_t)a.n_rows ? (int64_t)d : -1;
  const uint64_t r = d / a.id_stride;
  return (r * a.id_stride == d && r < (uint64_t)a.n_rows) ? (int64_t)r : -1;
}

__device__ __forceinline__ WbRec PlainLoadRec(const FanoutPlainArgs& a, const uint64_t node) {
  WbRec wr{0u, 0u, 0u, 0.f};
  const int64_t row = PlainFindRow(a, node);
  if (row >= 0) wr = a.wrec[row];
  return wr;
}

template <int WPS, bool COOP, bool LITE2 = false>
__global__ __launch_bounds__(256, WPS) void SampleFanoutPlainKernel(const FanoutPlainArgs a) {
  extern __shared__ __align__(16) uint8_t fp_smem[];
  const uint32_t lane = threadIdx.x & 63u;
  const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int waves_per_block = blockDim.x >> 6;
  const uint32_t c1 = (uint32_t)a.c1, c2 = (uint32_t)a.c2, gr = (uint32_t)a.gr, cap = (uint32_t)a.cap;
  const FanoutPlainLds L = FanoutPlainLayout(a.gr, a.c1, a.c2, a.cap, COOP);
  uint8_t* base = fp_smem + (size_t)wave_in_block * a.wave_lds;
  float4* s_stage = reinterpret_cast<float4*>(base + L.o_stage);
  uint32_t* s_blk = reinterpret_cast<uint32_t*>(base + L.o_blk);
  uint64_t* s_sid = reinterpret_cast<uint64_t*>(base + L.o_sid);
  uint64_t* s_c1 = reinterpret_cast<uint64_t*>(base + L.o_c1);
  unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(base + L.o_mask);
  float* s_sw = reinterpret_cast<float*>(base + L.o_sw);
  float* s_w1 = reinterpret_cast<float*>(base + L.o_w1);
  uint8_t* s_slot = base + L.o_slot;
  uint8_t* s_rep = base + L.o_rep;
  int8_t* s_st = reinterpret_cast<int8_t*>(base + L.o_st);
  uint8_t* s_rvalid = base + L.o_rvalid;
  const uint32_t hp1 = (c1 + 1u) >> 1, hp2 = c2 >> 1;     // pair-lanes per row (c2 is even)
  TinyDiv d_hp1, d_hp2, d_c1, d_c2;
  d_hp1.Set(hp1); d_hp2.Set(hp2); d_c1.Set(c1); d_c2.Set(c2);
  // The lane's roles - (root, pair) in hop 1, (slot / row, chunk) in hop 2 and the copy-out - are
  // functions of the lane number alone; they are recomputed at the head of each phase from a
  // lane number the compiler cannot see through (a dozen 24-bit multiplies per tile) instead
  // of living in a dozen registers across the whole tile loop.
  const uint32_t RPI = hp2 <= 1u ? 64u : d_hp2(64u);         // rows / slots a 64-lane step takes
  const int64_t n_tiles = (a.n + gr - 1) / gr;
  const int64_t wave0 = (int64_t)blockIdx.x * waves_per_block + wave_in_block;
  const int64_t wave_stride = (int64_t)gridDim.x * waves_per_block;
  for (int64_t tile = wave0; tile < n_tiles; tile += wave_stride) {
    const int64_t r0 = tile * gr;
    const uint32_t nr = (uint32_t)(a.n - r0 < (int64_t)gr ? a.n - r0 : (int64_t)gr);
    const uint32_t p1 = __umul24(nr, c1);
    const int64_t out1 = r0 * (int64_t)c1, out2 = out1 * (int64_t)c2;
    uint32_t call = a.call_id;
    if (a.mb_n > 0) {
      const uint32_t b = (uint32_t)((uint64_t)r0 / (uint64_t)a.mb_n);
      call = a.call_ids != nullptr ? a.call_ids[b] : a.call_id + b * a.call_stride;
    }
    if (lane < gr) s_mask[lane] = 0ull;
    WaveSync();
    // ---- P1: hop 1, a lane per pair of samples: pair jp1 of root q1 (gr * hp1 <= 64: one pass)
    const uint32_t lane1 = OpaqueLane(lane);
    const uint32_t q1 = d_hp1(lane1), jp1 = lane1 - __umul24(q1, hp1);
    const bool two1 = 2u * jp1 + 1u < c1;
    const uint32_t e1 = __umul24(q1, c1) + 2u * jp1;           // the pair's first sample in the tile
    const bool in1 = lane < gr * hp1 && q1 < nr;
    uint64_t node = 0;
    WbRec wr{0u, 0u, 0u, 0.f};
    if (in1) { node = a.roots[r0 + q1]; wr = PlainLoadRec(a, node); }
    const bool live = in1 && wr.deg > 0u;
    uint64_t id[2]; float w[2]; uint32_t m[2];
    {
      const Philox4 pb = RngBlock(a.seed, call, kDomainNeighbor, node, jp1);
      if (COOP) WaveSamplePairs(a, lane, __umul24(nr, hp1), s_stage, s_blk, wr, live, live && two1,
                                UnitFromWords(pb.w[0], pb.w[1]), UnitFromWords(pb.w[2], pb.w[3]), id, w, m);
      else PlainSamplePair(a, wr, live, live && two1, UnitFromWords(pb.w[0], pb.w[1]),
                           UnitFromWords(pb.w[2], pb.w[3]), id, w, m);
    }
    const bool by_edge = __ballot(in1 && wr.deg > 64u) == 0ull;   // every root of the tile has <= 64 edges
    const uint32_t off0 = live ? (m[0] - wr.lo) & 63u : 0u;
    const uint32_t off1 = live && two1 ? (m[1] - wr.lo) & 63u : off0;
    if (in1) {
      s_c1[e1] = live ? id[0] : 0;          // a row without samples hands node id 0 on
      s_w1[e1] = live ? w[0] : 0.f;
      if (two1) {
        s_c1[e1 + 1] = live ? id[1] : 0;
        s_w1[e1 + 1] = live ? w[1] : 0.f;
      }
      atomicOr(&s_mask[q1], (1ull << off0) | (1ull << off1));
      if (jp1 == 0) s_rvalid[q1] = live ? 1 : 0;
    }
    WaveSync();
    // ---- P2: slots of the distinct children ----------------------------------------------
    uint32_t n_slots = 0;
    if (by_edge) {
      // slot = rank of the sample's (root, drawn edge) among the tile's: the hop-1 lanes still
      // hold their edges
      uint32_t sbase = 0;
      unsigned long long mine = 0ull;
      for (uint32_t x = 0; x < nr; ++x) {
        const unsigned long long mk = s_mask[x];
        const uint32_t pc = (uint32_t)__popcll(mk);
        if (x < q1) sbase += pc;
        if (x == q1) mine = mk;
        n_slots += pc;
      }
      if (in1) {
        const uint32_t sl0 = sbase + (uint32_t)__popcll(mine & ((1ull << off0) - 1ull));
        s_slot[e1] = (uint8_t)sl0;
        s_rep[sl0] = (uint8_t)e1;           // (every sample of a slot names the same child)
        if (two1) {
          const uint32_t sl1 = sbase + (uint32_t)__popcll(mine & ((1ull << off1) - 1ull));
          s_slot[e1 + 1] = (uint8_t)sl1;
          s_rep[sl1] = (uint8_t)(e1 + 1u);
        }
      }
    } else {
      // some root has more than 64 edges: first occurrence by id among the root's samples
      const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
      for (uint32_t b = 0; b < p1; b += 64) {
        const uint32_t tk = b + lane;
        const bool in = tk < p1;
        const uint32_t q = d_c1(tk);
        const uint32_t j = tk - __umul24(q, c1);
        const uint64_t me = in ? s_c1[tk] : 0;
        uint32_t first = j;
        const uint64_t* row = s_c1 + __umul24(q, c1);
        for (uint32_t i = 0; i + 1 < c1; ++i) {          // wave-uniform trip count
          const uint64_t v = in ? row[i] : 0;
          if (in && i < j && first == j && v == me) first = i;
        }
        const bool rep = in && first == j;
        const uint64_t bal = __ballot(rep);
        if (rep) {
          const uint32_t slot = n_slots + (uint32_t)__popcll(bal & lt_mask);
          s_slot[tk] = (uint8_t)slot;
          s_rep[slot] = (uint8_t)tk;
        } else if (in) {
          s_slot[tk] = (uint8_t)(0x80u | first);
        }
        n_slots += (uint32_t)__popcll(bal);
      }
      WaveSync();
      for (uint32_t b = 0; b < p1; b += 64) {
        const uint32_t tk = b + lane;
        if (tk < p1) {
          const uint32_t v = s_slot[tk];
          if (v & 0x80u) s_slot[tk] = s_slot[__umul24(d_c1(tk), c1) + (v & 0x7Fu)];
        }
      }
    }
    WaveSync();
    n_slots = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_slots);     // (the same on every lane)
    if (a.row_index != nullptr) {
      for (uint32_t b = 0; b < p1; b += 64) {
        const uint32_t tk = b + lane;
        if (tk < p1) a.row_index[out1 + tk] = (uint32_t)(out1 + s_slot[tk]);
      }
    }
    // ---- P3 / P4 per chunk of `cap` slots -------------------------------------------------
    for (uint32_t s0 = 0; s0 < n_slots; s0 += cap) {
      const uint32_t ns = n_slots - s0 < cap ? n_slots - s0 : cap;
      // chunk x2 (a pair of draws / two ids) of slot / row rs of a step
      const uint32_t lane2 = OpaqueLane(lane);
      const uint32_t rs = d_hp2(lane2), x2 = lane2 - __umul24(rs, hp2);
      const bool act2 = rs < RPI;
      for (uint32_t sb = 0; sb < ns; sb += RPI) {
        const uint32_t sl = sb + rs;
        const bool in = act2 && sl < ns;
        uint64_t child = 0;
        WbRec cr{0u, 0u, 0u, 0.f};
        if (in) { child = s_c1[s_rep[s0 + sl]]; cr = PlainLoadRec(a, child); }
        const bool lv = in && cr.deg > 0u;
        uint64_t i2[2]; float w2[2]; uint32_t m2[2];
        const Philox4 pb = RngBlock(a.seed, call + 1u, kDomainNeighbor, child, x2);
        if (COOP) {
          const uint32_t left = ns - sb < RPI ? ns - sb : RPI;        // slots of this step
          WaveSamplePairs(a, lane, __umul24(left, hp2), s_stage, s_blk, cr, lv, lv, UnitFromWords(pb.w[0], pb.w[1]),
                          UnitFromWords(pb.w[2], pb.w[3]), i2, w2, m2);
        } else if (LITE2) {
          PlainSamplePairLite(a, cr, lv, lv, UnitFromWords(pb.w[0], pb.w[1]), UnitFromWords(pb.w[2], pb.w[3]), i2, w2);
        } else {
          PlainSamplePair(a, cr, lv, lv, UnitFromWords(pb.w[0], pb.w[1]), UnitFromWords(pb.w[2], pb.w[3]),
                          i2, w2, m2);
        }
        if (in) {
          fl_u64x2 iv;
          iv.x = lv ? i2[0] : (uint64_t)a.default_node;
          iv.y = lv ? i2[1] : (uint64_t)a.default_node;
          const uint32_t at = __umul24(sl, c2) + 2u * x2;
          *reinterpret_cast<fl_u64x2*>(s_sid + at) = iv;
          *reinterpret_cast<float2*>(s_sw + at) = make_float2(lv ? w2[0] : 0.f, lv ? w2[1] : 0.f);
          if (x2 == 0) s_st[sl] = lv ? 0 : -1;
        }
      }
      WaveSync();
      if (a.row_index != nullptr) {
        // the (unique rows, index) form: the chunk's rows as they are, once
        const int64_t row0 = (out1 + (int64_t)s0) * (int64_t)c2;
        const uint32_t tot = __umul24(ns, c2);
        for (uint32_t b = 0; b < tot; b += 128) {
          const uint32_t e = b + 2 * lane;
          if (e < tot) {
            *reinterpret_cast<fl_u64x2*>(a.id2 + row0 + e) = *reinterpret_cast<const fl_u64x2*>(s_sid + e);
            *reinterpret_cast<float2*>(a.w2 + row0 + e) = *reinterpret_cast<const float2*>(s_sw + e);
            const int32_t tv = s_st[d_c2(e)];
            *reinterpret_cast<int2*>(a.ty2 + row0 + e) = make_int2(tv, tv);
          }
        }
      } else {
        // -- P4 ids: RPI rows per pass, chunk x2 of row g ------------------------------------
        {
          uint64_t* dst = a.id2 + out2 + 2u * x2;
          for (uint32_t g0 = 0; g0 < p1; g0 += RPI) {
            const uint32_t g = g0 + rs;
            if (act2 && g < p1) {
              const uint32_t sl = (uint32_t)s_slot[g] - s0;
              if (sl < ns)
                *reinterpret_cast<fl_u64x2*>(dst + __umul24(g, c2)) =
                    *reinterpret_cast<const fl_u64x2*>(s_sid + __umul24(sl, c2) + 2u * x2);
            }
          }
        }
        // -- P4 weights / types: RPI pairs of rows per pass ----------------------------------
        {
          // chunk x2 of the PAIR of rows (A, B) = floats [4 x2, 4 x2 + 4) of their 2 c2
          const uint32_t ea = 4u * x2, eb = 4u * x2 + 2u;            // the chunk's two halves
          const uint32_t selb0 = ea >= c2 ? 1u : 0u, selb1 = eb >= c2 ? 1u : 0u;
          const uint32_t col0 = ea - (selb0 ? c2 : 0u), col1 = eb - (selb1 ? c2 : 0u);
          float* wdst = a.w2 + out2 + 4u * x2;
          int32_t* tdst = a.ty2 + out2 + 4u * x2;
          for (uint32_t g0 = 0; g0 < p1; g0 += 2u * RPI) {
            const uint32_t ga = g0 + 2u * rs;           // row A of the pair (B = A + 1)
            if (act2 && ga < p1) {
              const uint32_t g_h0 = ga + selb0, g_h1 = ga + selb1;
              const uint32_t sl0 = g_h0 < p1 ? (uint32_t)s_slot[g_h0] - s0 : 0xFFFFFFFFu;
              const uint32_t sl1 = g_h1 < p1 ? (uint32_t)s_slot[g_h1] - s0 : 0xFFFFFFFFu;
              const bool in0 = sl0 < ns, in1b = sl1 < ns;
              float2 wa = make_float2(0.f, 0.f), wb = wa;
              int32_t ta = -1, tb = -1;
              if (in0) { wa = *reinterpret_cast<const float2*>(s_sw + __umul24(sl0, c2) + col0); ta = s_st[sl0]; }
              if (in1b) { wb = *reinterpret_cast<const float2*>(s_sw + __umul24(sl1, c2) + col1); tb = s_st[sl1]; }
              float* wp = wdst + __umul24(ga, c2);
              int32_t* tp = tdst + __umul24(ga, c2);
              if (in0 && in1b) {
                *reinterpret_cast<float4*>(wp) = make_float4(wa.x, wa.y, wb.x, wb.y);
                *reinterpret_cast<int4*>(tp) = make_int4(ta, ta, tb, tb);
              } else if (in0) {
                *reinterpret_cast<float2*>(wp) = wa;
                *reinterpret_cast<int2*>(tp) = make_int2(ta, ta);
              } else if (in1b) {
                *reinterpret_cast<float2*>(wp + 2) = wb;
                *reinterpret_cast<int2*>(tp + 2) = make_int2(tb, tb);
              }
            }
          }
        }
      }
      WaveSync();              // the next chunk rewrites the slot rows
    }
    // ---- hop-1 outputs (contiguous over the tile) -----------------------------------------
    if ((p1 & 3u) == 0u) {
      for (uint32_t b = 0; b < p1; b += 128) {
        const uint32_t e = b + 2u * lane;
        if (e < p1) {
          const uint32_t qa = d_c1(e), qb = d_c1(e + 1u);
          fl_u64x2 v;
          v.x = s_rvalid[qa] ? s_c1[e] : (uint64_t)a.default_node;
          v.y = s_rvalid[qb] ? s_c1[e + 1] : (uint64_t)a.default_node;
          *reinterpret_cast<fl_u64x2*>(a.id1 + out1 + e) = v;
        }
      }
      for (uint32_t b = 0; b < p1; b += 256) {
        const uint32_t e = b + 4u * lane;
        if (e < p1) {
          *reinterpret_cast<float4*>(a.w1 + out1 + e) = *reinterpret_cast<const float4*>(s_w1 + e);
          int4 t;
          t.x = s_rvalid[d_c1(e)] ? 0 : -1;
          t.y = s_rvalid[d_c1(e + 1u)] ? 0 : -1;
          t.z = s_rvalid[d_c1(e + 2u)] ? 0 : -1;
          t.w = s_rvalid[d_c1(e + 3u)] ? 0 : -1;
          *reinterpret_cast<int4*>(a.ty1 + out1 + e) = t;
        }
      }
    } else {
      for (uint32_t b = 0; b < p1; b += 64) {
        const uint32_t tk = b + lane;
        if (tk < p1) {
          const bool ok = s_rvalid[d_c1(tk)] != 0;
          a.id1[out1 + tk] = ok ? s_c1[tk] : (uint64_t)a.default_node;
          a.w1[out1 + tk] = s_w1[tk];
          a.ty1[out1 + tk] = ok ? 0 : -1;
        }
      }
    }
    WaveSync();
  }
}

}  // namespace euler_gpu

#endif  // EULER_AMD_CSRC_FANOUT_PLAIN_H_
