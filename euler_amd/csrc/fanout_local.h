// 2-hop SampleFanout as ONE kernel with duplicate children found LOCALLY (gfx950).
//
// What the reference does (tf_euler/kernels/sample_fanout_op.cc:37-42,60-145 over the
// GQL rewrite of parser/compiler.cc:76-90): hop 2's roots are hop 1's samples; the
// rewrite samples every DISTINCT root once (ID_UNIQUE), and DATA_GATHER copies the
// row to every position that asked for it.  Rows are a pure function of
// (seed, call_id, node id) here (philox.h), so WHERE a duplicate is detected is
// free: any two positions that hold the same node get the same row.
//
// Rounds 1-2 found the duplicates of hop 2 globally (owner table over all rows,
// numbering, resolve), sampled the distinct nodes into scratch rows and expanded
// them: four kernel boundaries, a 0.4 GB table every XCD's L2 pulled, 46 MB of
// scratch written and 122 MB re-read - 49 % of the step was a copy.  But the
// duplicates are local: a uniformly drawn root of the metric graph has 1-2
// edges (76 %), so its 25 hop-1 samples are 2.8 distinct children on average
// (371 K (root, child) pairs per 131 072 roots against 285 K globally distinct
// children) - a wave can find them among its own lanes.
//
// One wave owns GR consecutive roots:
//   P1  hop 1: one lane per (root, j) sample - row record -> block-pivot search
//       (k1_search.h); ids / weights stay in LDS.
//   P2  local dedup: lane (root, j) looks for the first j' < j of its root with the
//       same child; first occurrences take a slot number (ballot rank).
//   P3  hop 2: one lane per (slot, x) sample, the slots of the wave pooled over its
//       lanes, results into LDS.  CAP slots per pass; a wave with more distinct
//       children (hub roots) repeats P3 / P4 per chunk of CAP slots.
//   P4  the wave streams its roots' output rows - contiguous in every output
//       array - from LDS: 16-byte id stores, 8- or 16-byte weight / type stores.
// All global loads of a wave precede all its stores (vector-memory operations
// retire in order on gfx950: a load issued behind a store waits for it).
//
// Same draws as the chained kernels: hop h uses call_id + h, stream = node id,
// draw x of a row = Philox block x >> 1, half x & 1.  A hop-1 row without samples
// (unknown root, empty type group) is default-filled and hands node id 0 to hop 2; a row
// the TF repack drops because its first sample is id 0 (Q1) is default-filled too, but
// hop 2 samples its real ids: the reference chains the hops on the core tensors.
#ifndef EULER_AMD_CSRC_FANOUT_LOCAL_H_
#define EULER_AMD_CSRC_FANOUT_LOCAL_H_

#include <hip/hip_runtime.h>

#include "k1_search.h"
#include "wb_index.h"

namespace euler_gpu {

// n / d for n * d < 2^32 (n: a position inside one wave's tile)
struct SmallDiv {
  uint32_t d, m;
  __host__ void Set(uint32_t div) { d = div; m = div <= 1 ? 0u : 0xFFFFFFFFu / div + 1u; }
  __device__ __forceinline__ uint32_t operator()(uint32_t n) const {
    return d <= 1 ? n : __umulhi(n, m);
  }
};

struct FanoutLocalArgs {
  GraphView g;
  uint64_t seed;
  const uint64_t* roots;
  int64_t n;
  int64_t default_node;
  uint32_t call_id;
  int32_t c1, c2, t1, t2;
  int32_t gr;               // roots per wave
  int32_t cap;              // hop-2 slots (distinct children) sampled per pass
  int32_t wide;             // 1: weights / types leave as 16-byte stores
  int32_t vec;              // 1: c2 even and the outputs 16-byte aligned (two ids per lane)
  int32_t wave_lds;         // bytes of LDS per wave
  SmallDiv div_c1, div_c2;
  SmallDiv div_h1, div_h2;  // by the pairs per row, (c1 + 1) / 2 and c2 / 2 (lean kernel)
  uint64_t* id1; float* w1; int32_t* ty1; uint8_t* mask0;
  uint64_t* id2; float* w2; int32_t* ty2; uint8_t* mask1;
  // several minibatches in one launch (euler_gpu_sample_fanout_multi): roots [mb_n * M],
  // minibatch b = root / mb_n draws with call id call_ids[b] (device array) or, when that
  // is null, call_id + b * call_stride.  mb_n == 0: one minibatch.  A tile never straddles
  // two minibatches (the launcher requires mb_n % gr == 0).
  int64_t mb_n;
  const uint32_t* call_ids;
  uint32_t call_stride;
  // WB == 3 (every hop lists k > 1 edge types: a type draw, then the neighbour draw): the
  // lists of the two hops and how Node::__SampleNeighbor reads a list of that length
  // (device_fns.h: kTypeSub / kTypeAll)
  int32_t k, type_mode;
  int32_t et1[kMaxListedTypes], et2[kMaxListedTypes];
  uint32_t* row_index;      // lean kernel, not null: the (unique rows, index) form - id2 / w2 / ty2
                            // receive each tile's DISTINCT hop-2 rows (row r0 * c1 + slot), row_index
                            // the row of every hop-1 sample; nothing is expanded
#ifdef EULER_GPU_MEASURE
  // measurement builds only (make MEASURE=1): the shipped kernel carries neither field nor branch
  unsigned long long* dbg;  // euler_gpu_set_debug_buffer: [tiles][8] phase stamps
  int32_t ablate;           // tuning key 36, lean kernel: 1 = no hop-2 stores, 2 = no hop-2 sampling,
                            // 4 = no hop-1 stores, 8 = no hop-1 sampling
#endif
};

#ifdef EULER_GPU_MEASURE
#define EG_FL_ABLATE(a, bits) (((a).ablate & (bits)) != 0)
#define EG_FL_DBG(a) ((a).dbg != nullptr)
#else
#define EG_FL_ABLATE(a, bits) (false)
#define EG_FL_DBG(a) false
#endif
#ifdef EULER_GPU_MEASURE
#define EG_FL_ABLATE_BITS(a) ((a).ablate)
#else
#define EG_FL_ABLATE_BITS(a) 0
#endif

// call id of the tile that starts at root r0
__device__ __forceinline__ uint32_t TileCallId(const FanoutLocalArgs& a, int64_t r0) {
  if (a.mb_n <= 0) return a.call_id;
  const uint32_t b = (uint32_t)((uint64_t)r0 / (uint64_t)a.mb_n);
  return a.call_ids != nullptr ? a.call_ids[b] : a.call_id + b * a.call_stride;
}

// LDS of one wave (bytes), and the offsets of its arrays
struct FanoutLocalLds {
  uint32_t o_sid, o_c1, o_sw, o_w1, o_slot, o_rep, o_svalid, o_rvalid, bytes;
};
__host__ __device__ inline FanoutLocalLds FanoutLocalLayout(int32_t gr, int32_t c1, int32_t c2,
                                                            int32_t cap) {
  FanoutLocalLds L;
  const uint32_t p = (uint32_t)gr * (uint32_t)c1;       // hop-1 samples of the wave
  const uint32_t s = (uint32_t)cap * (uint32_t)c2;      // hop-2 samples of one pass
  uint32_t o = 0;
  L.o_sid = o; o += s * 8;                  // u64 [cap][c2]  sampled ids of the slots
  L.o_c1 = o; o += p * 8;                   // u64 [gr][c1]   hop-1 ids (0 for a masked row)
  L.o_sw = o; o += s * 4;                   // f32 [cap][c2]
  L.o_w1 = o; o += p * 4;                   // f32 [gr][c1]   hop-1 weights
  L.o_slot = o; o += (p * 2 + 3) & ~3u;     // u16 [gr][c1]   slot of the sample's child
  L.o_rep = o; o += (p * 2 + 3) & ~3u;      // u16 [slots]    pooled index of a slot's first occurrence
  L.o_svalid = o; o += ((uint32_t)cap + 3) & ~3u;   // u8 [cap] the slot's row has samples
  L.o_rvalid = o; o += ((uint32_t)gr + 3) & ~3u;    // u8 [gr]  the root's row has samples
  L.bytes = (o + 15) & ~15u;
  return L;
}

typedef unsigned long long fl_u64x2 __attribute__((ext_vector_type(2)));

// PLAIN: the graph is the common case - one edge-type group per node with the row's total
// in its record, weighted, identity id map, no neighbour id 0 - and the
// kernel is compiled with those as constants (fewer live registers: 8 waves per SIMD).
// WPS: waves per SIMD the register allocation targets (8: 64 VGPRs with a few cold spills
// at the phase boundaries; 5: 96, none).
template <bool WIDE, bool PLAIN, int WPS>
__global__ __launch_bounds__(256, WPS) void SampleFanoutLocalKernel(
    const FanoutLocalArgs a_in) {
  FanoutLocalArgs a = a_in;
  if (PLAIN) {
    a.g.T = 1; a.g.meta_stride = 16; a.g.total_in_meta = 1; a.g.uniform_w = 0;
    a.g.map_mode = 0; a.g.has_zero_nbr = 0; a.g.monotone = 1;
  }
  extern __shared__ __align__(16) uint8_t fl_smem[];
  const int lane = threadIdx.x & 63;
  // wave-uniform, and the compiler should know it: everything derived from it (the LDS
  // pointers, the tile, the output offsets) then lives in SGPRs
  const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int waves_per_block = blockDim.x >> 6;
  const FanoutLocalLds L = FanoutLocalLayout(a.gr, a.c1, a.c2, a.cap);
  uint8_t* base = fl_smem + (size_t)wave_in_block * a.wave_lds;
  uint64_t* s_sid = reinterpret_cast<uint64_t*>(base + L.o_sid);
  uint64_t* s_c1 = reinterpret_cast<uint64_t*>(base + L.o_c1);
  float* s_sw = reinterpret_cast<float*>(base + L.o_sw);
  float* s_w1 = reinterpret_cast<float*>(base + L.o_w1);
  uint16_t* s_slot = reinterpret_cast<uint16_t*>(base + L.o_slot);
  uint16_t* s_rep = reinterpret_cast<uint16_t*>(base + L.o_rep);
  uint8_t* s_svalid = base + L.o_svalid;
  uint8_t* s_rvalid = base + L.o_rvalid;
  const uint32_t c1 = (uint32_t)a.c1, c2 = (uint32_t)a.c2, c12 = c1 * c2;
  const uint32_t gr = (uint32_t)a.gr, cap = (uint32_t)a.cap;
  const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const bool zero_rule = a.g.has_zero_nbr != 0;
  const int64_t n_tiles = (a.n + gr - 1) / gr;
  const int64_t wave0 = (int64_t)blockIdx.x * waves_per_block + wave_in_block;
  const int64_t wave_stride = (int64_t)gridDim.x * waves_per_block;
  for (int64_t tile = wave0; tile < n_tiles; tile += wave_stride) {
    const int64_t r0 = tile * gr;
    const uint32_t nr = (uint32_t)(a.n - r0 < (int64_t)gr ? a.n - r0 : (int64_t)gr);
    const uint32_t p1 = nr * c1;               // hop-1 samples of this tile
    const uint32_t p2 = nr * c12;              // hop-2 samples (output positions) of the tile
    const int64_t out1 = r0 * (int64_t)c1;     // first hop-1 output position of the tile
    const int64_t out2 = out1 * (int64_t)c2;
    const uint32_t tile_call = TileCallId(a, r0);
    uint32_t n_slots = 0, s0 = 0;
    // One sampling loop serves both hops (phase 0: the tile's hop-1 samples, phase 1: the
    // hop-2 samples of the slots [s0, s0 + cap)): the search is inlined once.
    int phase = 0;
#pragma nounroll
    while (true) {
      const uint32_t ns = phase == 0 ? 0u : (n_slots - s0 < cap ? n_slots - s0 : cap);
      const uint32_t tasks = phase == 0 ? p1 : ns * c2;
      const uint32_t cx = phase == 0 ? c1 : c2;
      const SmallDiv dv = phase == 0 ? a.div_c1 : a.div_c2;
      const int32_t et = phase == 0 ? a.t1 : a.t2;
      const uint32_t call = tile_call + (uint32_t)phase;
#pragma nounroll
      for (uint32_t b = 0; b < tasks; b += 64) {
        const uint32_t tk = b + lane;
        const bool live = tk < tasks;
        const uint32_t q = dv(tk);             // root of the tile / slot of the chunk
        const uint32_t x = tk - q * cx;        // draw of the row
        uint64_t id = 0;
        float w = 0.f;
        bool valid = false;
        if (live) {
          const uint64_t node = phase == 0 ? a.roots[r0 + q] : s_c1[s_rep[s0 + q]];
          Segment sg;
          valid = LoadSegment<true>(a.g, FindRow(a.g, node), et, &sg);
          if (valid) {
            const Philox4 pb = RngBlock(a.seed, call, kDomainNeighbor, node, x >> 1);
            const double u = (x & 1) ? UnitFromWords(pb.w[2], pb.w[3])
                                     : UnitFromWords(pb.w[0], pb.w[1]);
            BlockPivotSample(a.g, sg, u, &id, &w);
          }
          // the row is masked iff it has no samples or (TF sentinel rule, Q1) its FIRST
          // sample is id 0
          if (x == 0) (phase == 0 ? s_rvalid : s_svalid)[q] = (valid && !(zero_rule && id == 0)) ? 1 : 0;
        }
        if (phase == 0) {
          if (live) {
            // hop 2 is chained on the CORE ids (sample_fanout_op.cc:37-42: one GQL): a row
            // WITH samples hands them on even when the TF repack drops it for starting
            // with id 0; a row without samples hands on the core fill, node id 0
            s_c1[tk] = valid ? id : 0;
            s_w1[tk] = valid ? w : 0.f;
          }
        } else if (live) {
          s_sid[tk] = id;
          s_sw[tk] = w;
        }
      }
      WaveSync();
      if (phase == 0) {
        // ---- P2: first occurrence of every child among its root's samples -> slots ---
        for (uint32_t b = 0; b < p1; b += 64) {
          const uint32_t tk = b + lane;
          const bool live = tk < p1;
          const uint32_t g = a.div_c1(tk);
          const uint32_t j = tk - g * c1;
          const uint64_t mine = live ? s_c1[tk] : 0;
          uint32_t first = j;
          const uint64_t* row = s_c1 + g * c1;
          for (uint32_t i = 0; i + 1 < c1; ++i) {          // wave-uniform trip count
            const uint64_t v = live ? row[i] : 0;
            if (live && i < j && first == j && v == mine) first = i;
          }
          const bool rep = live && first == j;
          const uint64_t bal = __ballot(rep);
          if (rep) {
            const uint32_t slot = n_slots + (uint32_t)__popcll(bal & lt_mask);
            s_slot[tk] = (uint16_t)slot;
            s_rep[slot] = (uint16_t)tk;
          } else if (live) {
            s_slot[tk] = (uint16_t)(0x8000u | first);
          }
          n_slots += (uint32_t)__popcll(bal);
        }
        WaveSync();
        for (uint32_t b = 0; b < p1; b += 64) {
          const uint32_t tk = b + lane;
          if (tk < p1) {
            const uint32_t v = s_slot[tk];
            if (v & 0x8000u) {
              const uint32_t g = a.div_c1(tk);
              // the first occurrence's entry is final (no flag) and never rewritten here
              s_slot[tk] = s_slot[g * c1 + (v & 0x7FFFu)];
            }
          }
        }
        WaveSync();
        phase = 1;
        continue;
      }
      // ---- P4: the tile's hop-2 output rows whose slot is in this chunk ---------------
      if (a.vec) {
        // ids: two per lane (c2 is even: a pair never leaves its row)
        for (uint32_t b = 0; b < p2; b += 128) {
          const uint32_t p = b + 2 * lane;
          if (p < p2) {
            const uint32_t gj = a.div_c2(p);
            const uint32_t x = p - gj * c2;
            const uint32_t sl = (uint32_t)s_slot[gj] - s0;
            if (sl < ns) {
              const bool ok = s_svalid[sl] != 0;
              fl_u64x2 v;
              v.x = ok ? s_sid[sl * c2 + x] : (uint64_t)a.default_node;
              v.y = ok ? s_sid[sl * c2 + x + 1] : (uint64_t)a.default_node;
              *reinterpret_cast<fl_u64x2*>(a.id2 + out2 + p) = v;
              if (x == 0) a.mask1[out1 + gj] = ok ? 0 : 1;
              if (!WIDE) {
                float2 wv;
                wv.x = ok ? s_sw[sl * c2 + x] : 0.f;
                wv.y = ok ? s_sw[sl * c2 + x + 1] : 0.f;
                *reinterpret_cast<float2*>(a.w2 + out2 + p) = wv;
                const int32_t tv = ok ? a.t2 : -1;
                *reinterpret_cast<int2*>(a.ty2 + out2 + p) = make_int2(tv, tv);
              }
            }
          }
        }
        if (WIDE) {
          // weights and types: four per lane = two pairs, each inside one row
          for (uint32_t b = 0; b < p2; b += 256) {
            const uint32_t p = b + 4 * lane;
            if (p < p2) {
              const uint32_t gja = a.div_c2(p);
              const uint32_t xa = p - gja * c2;
              const uint32_t sla = (uint32_t)s_slot[gja] - s0;
              const bool ina = sla < ns;
              const bool oka = ina && s_svalid[sla] != 0;
              const bool hasb = p + 2 < p2;
              uint32_t gjb = gja, xb = xa + 2;
              if (xb >= c2) { xb -= c2; ++gjb; }
              const uint32_t slb = hasb ? (uint32_t)s_slot[gjb] - s0 : 0xFFFFFFFFu;
              const bool inb = hasb && slb < ns;
              const bool okb = inb && s_svalid[slb] != 0;
              float4 wv;
              wv.x = oka ? s_sw[sla * c2 + xa] : 0.f;
              wv.y = oka ? s_sw[sla * c2 + xa + 1] : 0.f;
              wv.z = okb ? s_sw[slb * c2 + xb] : 0.f;
              wv.w = okb ? s_sw[slb * c2 + xb + 1] : 0.f;
              const int32_t ta = oka ? a.t2 : -1, tb = okb ? a.t2 : -1;
              float* wp = a.w2 + out2 + p;
              int32_t* tp = a.ty2 + out2 + p;
              if (ina && inb) {
                *reinterpret_cast<float4*>(wp) = wv;
                *reinterpret_cast<int4*>(tp) = make_int4(ta, ta, tb, tb);
              } else if (ina) {
                *reinterpret_cast<float2*>(wp) = make_float2(wv.x, wv.y);
                *reinterpret_cast<int2*>(tp) = make_int2(ta, ta);
              } else if (inb) {
                *reinterpret_cast<float2*>(wp + 2) = make_float2(wv.z, wv.w);
                *reinterpret_cast<int2*>(tp + 2) = make_int2(tb, tb);
              }
            }
          }
        }
      } else {
        for (uint32_t b = 0; b < p2; b += 64) {
          const uint32_t p = b + lane;
          if (p < p2) {
            const uint32_t gj = a.div_c2(p);
            const uint32_t x = p - gj * c2;
            const uint32_t sl = (uint32_t)s_slot[gj] - s0;
            if (sl < ns) {
              const bool ok = s_svalid[sl] != 0;
              a.id2[out2 + p] = ok ? s_sid[sl * c2 + x] : (uint64_t)a.default_node;
              a.w2[out2 + p] = ok ? s_sw[sl * c2 + x] : 0.f;
              a.ty2[out2 + p] = ok ? a.t2 : -1;
              if (x == 0) a.mask1[out1 + gj] = ok ? 0 : 1;
            }
          }
        }
      }
      WaveSync();              // the next chunk rewrites the slot arrays
      s0 += cap;
      if (s0 >= n_slots) break;
    }
    // ---- hop-1 outputs (contiguous over the tile) -------------------------------------
    for (uint32_t b = 0; b < p1; b += 64) {
      const uint32_t tk = b + lane;
      if (tk < p1) {
        const uint32_t g = a.div_c1(tk);
        const uint32_t j = tk - g * c1;
        // s_rvalid was written by the root's j == 0 lane in phase 0
        const bool ok = s_rvalid[g] != 0;
        a.id1[out1 + tk] = ok ? s_c1[tk] : (uint64_t)a.default_node;
        a.w1[out1 + tk] = ok ? s_w1[tk] : 0.f;
        a.ty1[out1 + tk] = ok ? a.t1 : -1;
        if (j == 0) a.mask0[r0 + g] = ok ? 0 : 1;
      }
    }
    WaveSync();
  }
}


// ------------------------------------------------------------------------
// The lean build of the same kernel for PLAIN graphs: one edge-type group per node with
// the row total in its record, identity id map, no neighbour id 0, fewer than 2^31 edges
// (weighted: the search below; all weights 1.0f: LeanSamplePairUniform).  rocprofv3 on the general kernel above (profiles/r3_fl_v1_pmc.json): 3 400
// VALU instructions per wave and tile, every one of them a quad-cycle - 185 us of the
// step's 350 are VALU issue, the memory pipes idle half the time.  The search is the
// cost, so here it is rewritten for instruction count (same contract: the first m of
// the row with nw[m] > r, i.e. RandomSelect's index on a non-decreasing row):
//   * a lane draws the PAIR of samples one Philox block yields (x, x + 1): root id, row
//     record, the level ranges and the Philox block are per pair, and the two searches
//     run in lockstep (two loads in flight);
//   * compares are f32: for a float v and the f64 draw r, v > r  <=>  v > f with
//     f = the largest float <= r (round-down of r) - one conversion per draw instead of
//     one per key, full-rate compares;
//   * the candidate range of level k is carried as base-5 DIGITS of the first / last
//     block index (l[k] = 5 l[k+1] + digit), 3 bits each in one register, instead of
//     twenty registers of quotients; levels are walked in a scalar loop from the
//     wave's deepest level down, so the level's array offset is an SGPR;
//   * the leaf counts with a bit mask (v_cmp + v_addc per key, one v_bcnt) and fetches
//     nw[m-1], nw[m] and the id by index from the block line it just read;
//   * duplicate children by EDGE: a root of <= 64 edges ORs its drawn edge offsets into
//     one 64-bit LDS word; the slot of a sample is the rank of its bit (two distinct
//     edges with the same neighbour are two slots - the rows are equal anyway);
//   * hop 2 writes finished rows (default-filled when the child has none) into LDS
//     and the write phase is a copy.
// Draws whose r rounds up to the row total (Q3) and rows beyond the pivot levels' reach
// replay the reference's bisection over the flat arrays (cold).
// ------------------------------------------------------------------------
// FindRow for the strided identity map
__device__ __forceinline__ int64_t LeanFindRow(const GraphView& g, uint64_t id) {
  const uint64_t d = id - g.id_base;
  if (id < g.id_base) return -1;
  if (g.id_stride == 1) return d < (uint64_t)g.n_rows ? (int64_t)d : -1;
  const uint64_t r = d / g.id_stride;
  return (r * g.id_stride == d && r < (uint64_t)g.n_rows) ? (int64_t)r : -1;
}

// largest float <= r (r >= 0)
__device__ __forceinline__ float FloorToFloat(double r) {
  float f = (float)r;
  if ((double)f > r) f = __uint_as_float(__float_as_uint(f) - 1u);
  return f;
}

// Both draws of one Philox block on one row.  lo = first edge, deg > 0, total = the row's
// last running sum.  m[] = the edge drawn (global index).  TWO = false: draw 0 only (the
// walks take one sample per node and step).
template <bool TWO = true>
__device__ __forceinline__ void LeanSamplePair(const GraphView& g, const uint32_t lo,
                                               const int32_t deg, const float total,
                                               const bool live, const double u0, const double u1,
                                               uint64_t id[2], float w[2], uint32_t m[2],
                                               const int32_t ablate = 0) {
  const uint32_t hi = lo + (uint32_t)deg - 1u;
  const uint32_t l1 = lo / 10u, h1 = hi / 10u;
  const uint32_t lo_off = lo - 10u * l1, hi_off = hi - 10u * h1;
  // r = u * (total - 0) + 0: the subtraction and the addition of zero are exact
  const double r0 = __dmul_rn(u0, (double)total), r1 = __dmul_rn(u1, (double)total);
  const float f[2] = {FloorToFloat(r0), FloorToFloat(r1)};
  // level ranges as base-5 digits; K = first level with <= 4 candidates
  uint32_t lt = l1, ht = h1, ldig = 0, hdig = 0;
  int32_t K = 0;
  // Interpolation start.  A row of i.i.d. weights has nearly linear running sums: the
  // entry that holds r is within a fraction of sqrt(deg) edges of (r / total) * deg.
  // Level kg is the lowest whose entries span more than that; ONE window of four keys
  // around the guessed entry there either brackets r (then the walk continues below
  // kg, having skipped the levels above it) or it does not, and the walk starts at
  // the top as if nothing had been tried.  Same answer either way: the keys decide.
  const int32_t kg = deg <= 1600 ? 1 : deg <= 40000 ? 2 : deg <= 1000000 ? 3 : 4;
  uint32_t lg = l1, hg = h1;
  if (live && h1 != l1) {
    K = 1;
    while (ht - lt > 4u && K <= kPivotLevels) {
      const uint32_t l5 = lt / 5u, h5 = ht / 5u;
      ldig = (ldig << 3) | (lt - 5u * l5);
      hdig = (hdig << 3) | (ht - 5u * h5);
      lt = l5; ht = h5; ++K;
      if (K == kg) { lg = lt; hg = ht; }
    }
  }
  // cold: Q3 (r rounded up to the row's end) or a row beyond the levels' reach
  const bool cold0 = live && (!((double)total > r0) || K > kPivotLevels);
  const bool cold1 = TWO && live && (!((double)total > r1) || K > kPivotLevels);
  if (K > kPivotLevels) K = 0;
  uint32_t x[2] = {l1, l1};
  bool found[2] = {false, false}, onl[2] = {true, true};
  int32_t ks[2] = {K, TWO ? K : 0};     // the level a draw's walk starts at
  const bool guess = K > kg && !(ablate & 128);
  if (__ballot(guess) != 0ull) {
    const float* gl = kg == 1 ? g.skip1 : g.bpiv + (kg == 2 ? g.bpiv_off[2] : kg == 3 ? g.bpiv_off[3]
                                                                                   : g.bpiv_off[4]);
    const float inv = __frcp_rn(total) * (float)(hg - lg + 1u);
#pragma unroll
    for (int s = 0; s < (TWO ? 2 : 1); ++s) {
      if (guess) {
        uint32_t e = lg + (uint32_t)(f[s] * inv);
        // window of entries wq .. wq + 3 (all inside the row: hg - lg >= 5 here),
        // candidates wq .. wq + 4
        uint32_t wq = (e > lg + 2u ? e : lg + 2u) - 2u;
        wq = wq < hg - 4u ? wq : hg - 4u;
        const float4u kw = *reinterpret_cast<const float4u*>(gl + wq);
        int32_t pos = 0;
        pos += !(kw.x > f[s]) ? 1 : 0;
        pos += !(kw.y > f[s]) ? 1 : 0;
        pos += !(kw.z > f[s]) ? 1 : 0;
        pos += !(kw.w > f[s]) ? 1 : 0;
        const bool ok = (pos > 0 || wq == lg) && (pos < 4 || wq + 4u == hg);
        if (ok) {
          x[s] = wq + (uint32_t)pos;
          found[s] = pos < 4;
          onl[s] = x[s] == lg;
          ks[s] = kg - 1;
        }
      }
    }
  }
  int32_t kmax = 0;
#pragma unroll
  for (int k = 1; k <= kPivotLevels; ++k)
    kmax = __ballot(ks[0] >= k || ks[1] >= k) != 0ull ? k : kmax;
  if (ablate & 32) {          // measurement only: no level walk, some block of the row
    kmax = 0;
    x[0] = l1 + (__float_as_uint(f[0]) * 2654435761u >> 8) % (h1 - l1 + 1u);
    x[1] = l1 + (__float_as_uint(f[1]) * 2654435761u >> 8) % (h1 - l1 + 1u);
  }
  if (K - 1 > kmax) {         // the digits of the levels nobody of this wave walks
    const uint32_t sh = 3u * (uint32_t)(K - 1 - kmax);
    ldig >>= sh; hdig >>= sh;
  }
  for (int k = kmax; k >= 1; --k) {
    const float* lvl = k == 1 ? g.skip1 : g.bpiv + g.bpiv_off[k];
    if (k <= K) {
      uint32_t dl = 0, dh = 0;
      if (k < K) { dl = ldig & 7u; dh = hdig & 7u; ldig >>= 3; hdig >>= 3; }
#pragma unroll
      for (int s = 0; s < (TWO ? 2 : 1); ++s) {
        if (k <= ks[s]) {
          uint32_t c_lo, c_hi;
          if (k == K) { c_lo = lt; c_hi = ht; }
          else {
            c_lo = 5u * x[s] + (onl[s] ? dl : 0u);
            c_hi = 5u * x[s] + (found[s] ? 4u : dh);
          }
          const int32_t cnt = (int32_t)(c_hi - c_lo);
          const float4u kw = *reinterpret_cast<const float4u*>(lvl + c_lo);
          int32_t pos = 0;
          pos += (0 < cnt && !(kw.x > f[s])) ? 1 : 0;
          pos += (1 < cnt && !(kw.y > f[s])) ? 1 : 0;
          pos += (2 < cnt && !(kw.z > f[s])) ? 1 : 0;
          pos += (3 < cnt && !(kw.w > f[s])) ? 1 : 0;
          x[s] = c_lo + (uint32_t)pos;
          onl[s] = onl[s] && pos == 0;
          if (pos < cnt) found[s] = true;
        }
      }
    }
  }
  // leaf: block x[s] holds the answer
  if (!TWO) { id[1] = 0; w[1] = 0.f; m[1] = lo; }
#pragma unroll
  for (int s = 0; s < (TWO ? 2 : 1); ++s) {
    const EdgeBlock* bk = g.blk + x[s];
    const uint32_t i_lo = x[s] == l1 ? lo_off : 0u;
    const uint32_t i_hi = found[s] ? (uint32_t)(kEdgesPerBlock - 1) : hi_off;   // inclusive
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
    if (live && !(ablate & 64)) {
      a0 = *reinterpret_cast<const float4*>(bk->pw);
      a1 = *reinterpret_cast<const float4*>(bk->pw + 4);
      a2 = *reinterpret_cast<const float4*>(bk->pw + 8);   // pw[8], pw[9], prev_last, pad
    }
    // bit j = [pw[j] <= r] for j = 0 .. 8, built from the top: mask = 2 mask + bit
    uint32_t le = 0;
    le = 2u * le + (!(a2.x > f[s]) ? 1u : 0u);
    le = 2u * le + (!(a1.w > f[s]) ? 1u : 0u);
    le = 2u * le + (!(a1.z > f[s]) ? 1u : 0u);
    le = 2u * le + (!(a1.y > f[s]) ? 1u : 0u);
    le = 2u * le + (!(a1.x > f[s]) ? 1u : 0u);
    le = 2u * le + (!(a0.w > f[s]) ? 1u : 0u);
    le = 2u * le + (!(a0.z > f[s]) ? 1u : 0u);
    le = 2u * le + (!(a0.y > f[s]) ? 1u : 0u);
    le = 2u * le + (!(a0.x > f[s]) ? 1u : 0u);
    const uint32_t range = ((1u << i_hi) - 1u) & ~((1u << i_lo) - 1u);     // bits [i_lo, i_hi)
    const uint32_t i = i_lo + (uint32_t)__popc(le & range);
    m[s] = 10u * x[s] + i;
    id[s] = m[s]; w[s] = 0.f;
    if (live && !(ablate & 16)) {
      id[s] = bk->nbr[i];
      // nw[m] and nw[m-1] out of the registers the leaf was read into (a second load of
      // the line costs the CU's memory pipe more than a dozen selects cost a SIMD)
      const float v0 = a0.x, v1 = a0.y, v2 = a0.z, v3 = a0.w, v4 = a1.x, v5 = a1.y, v6 = a1.z,
                  v7 = a1.w, v8 = a2.x, v9 = a2.y;
      const bool b0 = (i & 1u) != 0, b1 = (i & 2u) != 0, b2 = (i & 4u) != 0, b3 = (i & 8u) != 0;
      // v[i]
      const float s01 = b0 ? v1 : v0, s23 = b0 ? v3 : v2, s45 = b0 ? v5 : v4, s67 = b0 ? v7 : v6,
                  s89 = b0 ? v9 : v8;
      const float q03 = b1 ? s23 : s01, q47 = b1 ? s67 : s45;
      const float nw_m = b3 ? s89 : (b2 ? q47 : q03);
      // v[i - 1] (i >= 1), prev_last for i == 0: index i - 1 = (i + 15) & 15 over
      // {v0 .. v8}, with slot 15 = prev_last
      const uint32_t ip = (i + 15u) & 15u;
      const bool c0 = (ip & 1u) != 0, c1b = (ip & 2u) != 0, c2b = (ip & 4u) != 0, c3 = (ip & 8u) != 0;
      const float r01 = c0 ? v1 : v0, r23 = c0 ? v3 : v2, r45 = c0 ? v5 : v4, r67 = c0 ? v7 : v6;
      const float u03 = c1b ? r23 : r01, u47 = c1b ? r67 : r45;
      const float lowp = c2b ? u47 : u03;                    // ip in 0 .. 7
      float prev = c3 ? (ip == 8u ? v8 : a2.z) : lowp;       // 8, or 15 = prev_last
      if (m[s] == lo) prev = 0.f;                  // `mid ? nw[mid-1] : 0` is row-relative
      w[s] = __fsub_rn(nw_m, prev);
    }
  }
  if (__ballot(cold0 || cold1) != 0ull) {
    // the reference's own bisection over the flat running sums (RandomSelect,
    // compact_weighted_collection.h:30-52) - right on every row, slow, rare; once in
    // the code for both draws
#pragma nounroll
    for (int s = 0; s < (TWO ? 2 : 1); ++s) {
      if (s == 0 ? cold0 : cold1) {
        const float* nw = g.prefix_w + lo;
        const uint32_t mid = (uint32_t)RandomSelect(nw, 0, (uint64_t)(deg - 1), s == 0 ? u0 : u1);
        const uint64_t ci = g.nbr[lo + mid];
        const float cw = __fsub_rn(nw[mid], mid == 0u ? 0.f : nw[mid - 1]);
        if (s == 0) { id[0] = ci; w[0] = cw; m[0] = lo + mid; }
        else { id[1] = ci; w[1] = cw; m[1] = lo + mid; }
      }
    }
  }
}

// The same pair of draws on a graph whose weights are all 1.0f (H1, configs[1]): the running
// sums of a row are 1, 2, 3, ..., so the first m with nw[m] > r is floor(r) and the weight
// is 1.0f - no search, no sums read, one id load per draw (k1_search.h: uniform_w).
__device__ __forceinline__ void LeanSamplePairUniform(const GraphView& g, const uint32_t lo,
                                                      const int32_t deg, const float total,
                                                      const bool live, const double u0,
                                                      const double u1, uint64_t id[2], float w[2],
                                                      uint32_t m[2]) {
  const double r0 = __dmul_rn(u0, (double)total), r1 = __dmul_rn(u1, (double)total);
  const bool cold0 = live && !((double)total > r0), cold1 = live && !((double)total > r1);
  m[0] = lo + (uint32_t)r0; m[1] = lo + (uint32_t)r1;
  id[0] = 0; id[1] = 0; w[0] = 1.0f; w[1] = 1.0f;
  if (live && !cold0) id[0] = g.nbr[m[0]];
  if (live && !cold1) id[1] = g.nbr[m[1]];
  if (__ballot(cold0 || cold1) != 0ull) {
#pragma nounroll
    for (int s = 0; s < 2; ++s) {
      if (s == 0 ? cold0 : cold1) {           // Q3: r rounded up to the row's end
        const float* nw = g.prefix_w + lo;
        const uint32_t mid = (uint32_t)RandomSelect(nw, 0, (uint64_t)(deg - 1), s == 0 ? u0 : u1);
        const uint64_t ci = g.nbr[lo + mid];
        const float cw = __fsub_rn(nw[mid], mid == 0u ? 0.f : nw[mid - 1]);
        if (s == 0) { id[0] = ci; w[0] = cw; m[0] = lo + mid; }
        else { id[1] = ci; w[1] = cw; m[1] = lo + mid; }
      }
    }
  }
}

// Both draws of one Philox block on one row through the weight-bucket index (wb_index.h):
// per draw ONE 128-byte line - the bucket's block - instead of a walk over pivot levels and
// a leaf.  Same contract as LeanSamplePair: m[] = the flat edge drawn; draws that round up to
// the row's total (Q3) and draws whose block does not bracket them replay the reference's
// bisection.  The two blocks' loads are issued before either is examined.
template <bool TWO = true>
__device__ __forceinline__ void WbSamplePair(const GraphView& g, const WbRec rec, const bool live,
                                             const double u0, const double u1, uint64_t id[2],
                                             float w[2], uint32_t m[2]) {
  const double r0 = __dmul_rn(u0, (double)rec.total), r1 = __dmul_rn(u1, (double)rec.total);
  bool cold0 = live && !((double)rec.total > r0);
  bool cold1 = TWO && live && !((double)rec.total > r1);
  const float f0 = WbFloorToFloat(r0), f1 = WbFloorToFloat(r1);
  const uint32_t nbk = WbBuckets(rec.deg);
  uint32_t j0 = 0u, j1 = 0u;
  if (nbk > 1u) {
    const float scale = WbScale(nbk, rec.total);
    j0 = WbBucketOf(f0, nbk, scale);
    j1 = WbBucketOf(f1, nbk, scale);
  }
  // (a dead lane's record is all zeros: block 0, a valid line nobody uses)
  const EdgeBlock* b0 = g.wb + rec.wb_lo + j0;
  const EdgeBlock* b1 = g.wb + rec.wb_lo + (TWO ? j1 : j0);
  const WbKeys k0 = WbLoadKeys(b0);
  WbKeys k1 = k0;
  if (TWO) k1 = WbLoadKeys(b1);
  id[0] = 0; id[1] = 0; w[0] = 0.f; w[1] = 0.f; m[0] = rec.lo; m[1] = rec.lo;
  const int32_t i0 = WbPickKeys(k0, f0, &w[0], &m[0]);
  const int32_t i1 = TWO ? WbPickKeys(k1, f1, &w[1], &m[1]) : 0;
  const bool hot0 = live && !cold0 && i0 >= 0;
  const bool hot1 = TWO && live && !cold1 && i1 >= 0;
  if (hot0) id[0] = b0->nbr[i0];
  if (hot1) id[1] = b1->nbr[i1];
  cold0 = live && !hot0;
  cold1 = TWO && live && !hot1;
  if (!TWO) { id[1] = 0; w[1] = 0.f; m[1] = rec.lo; }
  if (__ballot(cold0 || cold1) != 0ull) {
#pragma nounroll
    for (int s = 0; s < (TWO ? 2 : 1); ++s) {
      if (s == 0 ? cold0 : cold1) {
        const float* nw = g.prefix_w + rec.lo;
        const uint32_t mid = (uint32_t)RandomSelect(nw, 0, (uint64_t)(rec.deg - 1u), s == 0 ? u0 : u1);
        const uint64_t ci = g.nbr[rec.lo + mid];
        const float cw = __fsub_rn(nw[mid], mid == 0u ? 0.f : nw[mid - 1]);
        if (s == 0) { id[0] = ci; w[0] = cw; m[0] = rec.lo + mid; }
        else { id[1] = ci; w[1] = cw; m[1] = rec.lo + mid; }
      }
    }
  }
}

// The same pair of draws on ANY graph the weight-bucket index serves - several edge-type
// groups per node, hashed ids - for one listed type: the segment's limits and the row's first
// block come out of the row's weight-bucket record (common.h: GraphView::wbg) alone.
struct WbSeg {
  uint32_t wb_lo, row_deg;      // the row: first block, edges (the buckets are the ROW's)
  uint32_t lo, deg;             // the listed type's segment: first flat edge, edges
  float row_total, lim_b, lim_e;
  uint32_t row_lo;              // the row's first flat edge: the cold path replays RandomSelect over
                                // prefix_w + row_lo (the sums restart at every row)
};

// Graphs with a hash id map and at most two edge-type groups (what a converted dataset looks
// like: arbitrary ids, 'train' / 'train_removed'): the hash slot CARRIES the row's record
// (common.h: GraphView::fat - 64 bytes: key | wb_lo row_lo | te[2] lim[2] | ts[2] row), so a
// cold root / child costs one line where the 16-byte slot + the record cost two dependent
// ones.  Same slot index as the 16-byte table (same probe sequence).  Returns false on a miss.
__device__ __forceinline__ bool FatFind(const GraphView& g, const uint64_t id, uint4* hd, uint4* tl,
                                        const uint4** slot) {
  uint64_t h = Mix64(id) & g.hash_mask;
  for (uint64_t probes = 0; probes <= g.hash_mask; ++probes) {
    const uint4* s = reinterpret_cast<const uint4*>(g.fat + h * 64);
    const uint4 a = s[0], b = s[1];
    if ((int32_t)b.y == -1) return false;                       // empty slot: no such node
    if (a.x == (uint32_t)id && a.y == (uint32_t)(id >> 32)) { *hd = a; *tl = b; *slot = s; return true; }
    h = (h + 1) & g.hash_mask;
  }
  return false;
}

__device__ __forceinline__ void LoadWbSeg(const GraphView& g, uint64_t node, int32_t t, WbSeg* s) {
  s->wb_lo = 0; s->row_deg = 0; s->lo = 0; s->deg = 0; s->row_total = 0.f; s->lim_b = 0.f; s->lim_e = 0.f;
  s->row_lo = 0;
  if (g.fat != nullptr) {
    uint4 hd, tl;
    const uint4* slot;
    if (t < 0 || t >= g.T || !FatFind(g, node, &hd, &tl, &slot)) return;
    const int32_t te0 = (int32_t)tl.x, te1 = (int32_t)tl.y;
    const float lim0 = __uint_as_float(tl.z), lim1 = __uint_as_float(tl.w);
    const int32_t b = t == 0 ? 0 : te0, e = t == 0 ? te0 : te1;
    s->wb_lo = hd.z;
    s->row_lo = hd.w;
    s->row_deg = (uint32_t)(g.T == 1 ? te0 : te1);
    s->row_total = g.T == 1 ? lim0 : lim1;
    s->lim_e = t == 0 ? lim0 : lim1;
    s->lim_b = t == 0 ? 0.f : lim0;
    s->lo = hd.w + (uint32_t)b;
    s->deg = e > b ? (uint32_t)(e - b) : 0u;
    return;
  }
  const int64_t row = FindRow(g, node);
  if (row < 0 || t < 0 || t >= g.T) return;
  const uint8_t* wrec = g.trec + row * (int64_t)g.trec_stride;
  const uint32_t* hd = reinterpret_cast<const uint32_t*>(wrec);
  const int32_t* te = reinterpret_cast<const int32_t*>(wrec + 8);
  const float* lim = reinterpret_cast<const float*>(wrec + 8 + 4 * g.T);
  const int32_t b = t == 0 ? 0 : te[t - 1], e = te[t];
  s->wb_lo = hd[0];
  s->row_deg = (uint32_t)te[g.T - 1];
  s->row_total = lim[g.T - 1];
  s->lim_e = lim[t];
  s->lim_b = t == 0 ? 0.f : lim[t - 1];
  s->lo = hd[1] + (uint32_t)b;
  s->row_lo = hd[1];
  s->deg = e > b ? (uint32_t)(e - b) : 0u;
}

// Two draws of one row, each on its own segment (sg0 / sg1 of types t0 / t1: the same one when
// the hop lists one type, the drawn ones when it lists several).
template <bool TWO = true>
__device__ __forceinline__ void WbSamplePairG2(const GraphView& g, const WbSeg sg0, const WbSeg sg1,
                                               const int32_t t0, const int32_t t1, const bool live0,
                                               const bool live1, const double u0, const double u1,
                                               uint64_t id[2], float w[2], uint32_t m[2]) {
  // r = u * (limit_end - limit_begin) + limit_begin as the reference rounds it (ScaleDraw)
  const double span0 = (double)__fsub_rn(sg0.lim_e, sg0.lim_b);
  const double span1 = (double)__fsub_rn(sg1.lim_e, sg1.lim_b);
  const double r0 = __dadd_rn(__dmul_rn(u0, span0), (double)sg0.lim_b);
  const double r1 = __dadd_rn(__dmul_rn(u1, span1), (double)sg1.lim_b);
  bool cold0 = live0 && !((double)sg0.lim_e > r0);
  bool cold1 = TWO && live1 && !((double)sg1.lim_e > r1);
  const float f0 = WbFloorToFloat(r0), f1 = WbFloorToFloat(r1);
  const uint32_t nbk = WbBuckets(sg0.row_deg);          // (the buckets are the ROW's)
  uint32_t j0 = 0u, j1 = 0u;
  if (nbk > 1u) {
    const float scale = WbScale(nbk, sg0.row_total);
    j0 = WbBucketOf(f0, nbk, scale);
    j1 = WbBucketOf(f1, nbk, scale);
  }
  const EdgeBlock* b0 = g.wb + sg0.wb_lo + j0;
  const EdgeBlock* b1 = g.wb + sg0.wb_lo + (TWO ? j1 : j0);
  const WbKeys k0 = WbLoadKeys(b0);
  WbKeys k1 = k0;
  if (TWO) k1 = WbLoadKeys(b1);
  id[0] = 0; id[1] = 0; w[0] = 0.f; w[1] = 0.f; m[0] = sg0.lo; m[1] = sg1.lo;
  const int32_t i0 = WbPickKeys(k0, f0, &w[0], &m[0]);
  const int32_t i1 = TWO ? WbPickKeys(k1, f1, &w[1], &m[1]) : 0;
  const bool hot0 = live0 && !cold0 && i0 >= 0;
  const bool hot1 = TWO && live1 && !cold1 && i1 >= 0;
  if (hot0) id[0] = b0->nbr[i0];
  if (hot1) id[1] = b1->nbr[i1];
  cold0 = live0 && !hot0;
  cold1 = TWO && live1 && !hot1;
  if (!TWO) { id[1] = 0; w[1] = 0.f; m[1] = sg1.lo; }
  if (__ballot(cold0 || cold1) != 0ull) {
#pragma nounroll
    for (int s = 0; s < (TWO ? 2 : 1); ++s) {
      if (s == 0 ? cold0 : cold1) {
        // the reference's bisection over the group's running sums (they restart at the row's
        // first edge: positions are row-relative, b = the group's first)
        const WbSeg& sg = s == 0 ? sg0 : sg1;
        const float* nw = g.prefix_w + sg.row_lo;
        const uint32_t b = sg.lo - sg.row_lo;
        const uint32_t mid = (uint32_t)RandomSelect(nw, (uint64_t)b, (uint64_t)(b + sg.deg - 1u), s == 0 ? u0 : u1);
        const uint64_t ci = g.nbr[sg.row_lo + mid];
        const float cw = __fsub_rn(nw[mid], mid == 0u ? 0.f : nw[mid - 1]);
        if (s == 0) { id[0] = ci; w[0] = cw; m[0] = sg.row_lo + mid; }
        else { id[1] = ci; w[1] = cw; m[1] = sg.row_lo + mid; }
      }
    }
  }
}

template <bool TWO = true>
__device__ __forceinline__ void WbSamplePairG(const GraphView& g, const WbSeg sg, const int32_t t,
                                              const bool live, const double u0, const double u1,
                                              uint64_t id[2], float w[2], uint32_t m[2]) {
  WbSamplePairG2<TWO>(g, sg, sg, t, t, live, live, u0, u1, id, w, m);
}

// ---- hops that list SEVERAL edge types (WB == 3) ----------------------------------------
// Node::__SampleNeighbor (core/graph/node.cc:98-167) then draws the TYPE first - a CDF over the
// listed types in the listed order, or over all groups when the list is as long as the graph
// has types (device_fns.h: TypeModeOf) - and the neighbour inside that type's group with the
// second half of the sample's Philox block (one block per SAMPLE here, not per pair).  Both
// come out of the row's weight-bucket record, which carries the type sums for that.
struct WbRowT {
  const uint32_t* hd;           // {wb_lo, row_lo}
  const int32_t* te;            // [T] ends of the type groups
  const float* lim;             // [T] running sums at those ends
  const float* tsum;            // [T] cumulative type sums
  int64_t row;
  uint32_t row_lo, row_deg;
  bool valid;                   // the listed types have weight: the row yields samples
};

__device__ __forceinline__ void LoadWbRowT(const GraphView& g, const uint64_t node, const int32_t mode,
                                           const int32_t* et, const int32_t k, WbRowT* r) {
  r->hd = nullptr; r->te = nullptr; r->lim = nullptr; r->tsum = nullptr;
  r->row_lo = 0; r->row_deg = 0; r->valid = false;
  r->row = FindRow(g, node);
  if (r->row < 0) return;
  const uint8_t* rec = g.trec + r->row * (int64_t)g.trec_stride;
  r->hd = reinterpret_cast<const uint32_t*>(rec);
  r->te = reinterpret_cast<const int32_t*>(rec + 8);
  r->lim = reinterpret_cast<const float*>(rec + 8 + 4 * g.T);
  r->tsum = reinterpret_cast<const float*>(rec + 8 + 8 * g.T);
  r->row_lo = r->hd[1];
  r->row_deg = (uint32_t)r->te[g.T - 1];
  if (mode == kTypeSub) {
    bool ok = true;
    for (int32_t i = 0; i < k; ++i) ok = ok && et[i] >= 0 && et[i] < g.T;      // node.cc:109-118
    r->valid = ok && SubTypeSum{r->tsum, et}((uint64_t)(k - 1)) != 0.f;        // node.cc:139-141
  } else {
    r->valid = r->tsum[g.T - 1] != 0.f;
  }
}

// the type of one sample and that type's segment; false = a group without edges was drawn
// (only through an inconsistent row: the reference reads out of range there, SampleAt in
// device_fns.h answers the sentinel {0, 0, type 0})
__device__ __forceinline__ bool WbTypedSeg(const GraphView& g, const WbRowT& r, const int32_t mode,
                                           const int32_t* et, const int32_t k, const double u_type,
                                           int32_t* t_out, WbSeg* s) {
  const int32_t t = mode == kTypeSub
      ? et[RandomSelectT(SubTypeSum{r.tsum, et}, 0, (uint64_t)(k - 1), u_type)]
      : (int32_t)RandomSelect(r.tsum, 0, (uint64_t)(g.T - 1), u_type);
  const int32_t b = t == 0 ? 0 : r.te[t - 1], e = r.te[t];
  s->wb_lo = r.hd[0];
  s->row_deg = r.row_deg;
  s->row_total = r.lim[g.T - 1];
  s->lim_e = r.lim[t];
  s->lim_b = t == 0 ? 0.f : r.lim[t - 1];
  s->lo = r.row_lo + (uint32_t)b;
  s->deg = e > b ? (uint32_t)(e - b) : 0u;
  s->row_lo = r.row_lo;
  *t_out = t;
  return e > b;
}

// The two samples 2 jp and 2 jp + 1 of `node` on a typed hop.  tt[] = their types (-1: the row
// yields nothing, 0 with id 0 for the sentinel); *sentinel = a sentinel was drawn (the caller
// must not tell children apart by edge offset then).
__device__ __forceinline__ void WbSampleTypedPair(const GraphView& g, const WbRowT& r, const int32_t mode,
                                                  const int32_t* et, const int32_t k, const uint64_t seed,
                                                  const uint32_t call, const uint64_t node,
                                                  const uint32_t jp, const bool live, const bool two,
                                                  uint64_t id[2], float w[2], uint32_t m[2],
                                                  int32_t tt[2], bool* sentinel) {
  const Philox4 pa = RngBlock(seed, call, kDomainNeighbor, node, 2u * jp);
  const Philox4 pb = RngBlock(seed, call, kDomainNeighbor, node, 2u * jp + 1u);
  WbSeg sg0, sg1;
  sg0.wb_lo = 0; sg0.row_deg = 0; sg0.lo = 0; sg0.deg = 0; sg0.row_total = 0.f; sg0.lim_b = 0.f; sg0.lim_e = 0.f; sg0.row_lo = 0;
  sg1 = sg0;
  tt[0] = -1; tt[1] = -1;
  bool ok0 = false, ok1 = false;
  if (live) {
    ok0 = WbTypedSeg(g, r, mode, et, k, UnitFromWords(pa.w[0], pa.w[1]), &tt[0], &sg0);
    if (two) ok1 = WbTypedSeg(g, r, mode, et, k, UnitFromWords(pb.w[0], pb.w[1]), &tt[1], &sg1);
    else sg1 = sg0;
  }
  WbSamplePairG2<true>(g, sg0, sg1, tt[0], tt[1], ok0, ok1, UnitFromWords(pa.w[2], pa.w[3]),
                       UnitFromWords(pb.w[2], pb.w[3]), id, w, m);
  *sentinel = live && (!ok0 || (two && !ok1));
  if (live && !ok0) { id[0] = 0; w[0] = 0.f; m[0] = r.row_lo; tt[0] = 0; }
  if (live && two && !ok1) { id[1] = 0; w[1] = 0.f; m[1] = r.row_lo; tt[1] = 0; }
}

// Graphs with at most 4 edge-type groups (every dataset the reference ships has 2 or 3): the
// row record's three arrays are loaded ONCE into registers, side by side, and the type draw,
// the group's ends and its limits are register selects - the general form above walks the
// record with dependent loads (hits, but four round trips in a row per sample).  WB == 4.
struct WbRowT4 {
  uint32_t wb_lo, row_lo;
  int32_t te[4];
  float lim[4], ts[4];
  int64_t row;
  uint32_t row_deg;
  float row_total;
  bool valid;
};

template <typename V>
__device__ __forceinline__ V At4(const V a[4], const int32_t i) {
  return i == 0 ? a[0] : i == 1 ? a[1] : i == 2 ? a[2] : a[3];
}

struct RegSum4 {
  const float* ts;          // registers (fully unrolled selects)
  __device__ __forceinline__ float operator()(uint64_t i) const { return At4(ts, (int32_t)i); }
};

struct SubTypeSum4 {         // SubTypeSum (device_fns.h) over register sums: same sequential f32 adds
  const float* ts;
  const int32_t* edge_types;
  __device__ __forceinline__ float operator()(uint64_t i) const {
    float s = 0.f;
    for (uint64_t x = 0; x <= i; ++x) {
      const int32_t t = edge_types[x];
      s = EG_FADD(s, EG_FSUB(At4(ts, t), t > 0 ? At4(ts, t - 1) : 0.f));
    }
    return s;
  }
};

__device__ __forceinline__ void LoadWbRowT4(const GraphView& g, const uint64_t node, const int32_t mode,
                                            const int32_t* et, const int32_t k, WbRowT4* r) {
  r->wb_lo = 0; r->row_lo = 0; r->row_deg = 0; r->row_total = 0.f; r->valid = false;
#pragma unroll
  for (int i = 0; i < 4; ++i) { r->te[i] = 0; r->lim[i] = 0.f; r->ts[i] = 0.f; }
  const int32_t T = g.T;
  if (g.fat != nullptr) {              // T <= 2: the record rides in the hash slot
    uint4 hd, tl;
    const uint4* slot;
    if (!FatFind(g, node, &hd, &tl, &slot)) { r->row = -1; return; }
    const uint4 x = slot[2];
    r->row = 0;
    r->wb_lo = hd.z; r->row_lo = hd.w;
    const int32_t te0 = (int32_t)tl.x, te1 = (int32_t)tl.y;
    const float lim0 = __uint_as_float(tl.z), lim1 = __uint_as_float(tl.w);
    const float ts0 = __uint_as_float(x.x), ts1 = __uint_as_float(x.y);
    r->te[0] = te0; r->lim[0] = lim0; r->ts[0] = ts0;
#pragma unroll
    for (int i = 1; i < 4; ++i) { r->te[i] = te1; r->lim[i] = lim1; r->ts[i] = ts1; }
  } else {
  r->row = FindRow(g, node);
  if (r->row < 0) return;
  const uint8_t* rec = g.trec + r->row * (int64_t)g.trec_stride;
  const uint32_t* hd = reinterpret_cast<const uint32_t*>(rec);
  const int32_t* te = reinterpret_cast<const int32_t*>(rec + 8);
  const float* lim = reinterpret_cast<const float*>(rec + 8 + 4 * T);
  const float* ts = reinterpret_cast<const float*>(rec + 8 + 8 * T);
  r->wb_lo = hd[0]; r->row_lo = hd[1];
#pragma unroll
  for (int i = 0; i < 4; ++i) {           // entries past T repeat the last one (never selected)
    const int x = i < T ? i : T - 1;
    r->te[i] = te[x]; r->lim[i] = lim[x]; r->ts[i] = ts[x];
  }
  }
  r->row_deg = (uint32_t)r->te[3];
  r->row_total = r->lim[3];
  if (mode == kTypeSub) {
    bool ok = true;
    for (int32_t i = 0; i < k; ++i) ok = ok && et[i] >= 0 && et[i] < T;
    r->valid = ok && SubTypeSum4{r->ts, et}((uint64_t)(k - 1)) != 0.f;
  } else {
    r->valid = r->ts[3] != 0.f;
  }
}

__device__ __forceinline__ bool WbTypedSeg4(const GraphView& g, const WbRowT4& r, const int32_t mode,
                                            const int32_t* et, const int32_t k, const double u_type,
                                            int32_t* t_out, WbSeg* s) {
  const int32_t t = mode == kTypeSub
      ? et[RandomSelectT(SubTypeSum4{r.ts, et}, 0, (uint64_t)(k - 1), u_type)]
      : (int32_t)RandomSelectT(RegSum4{r.ts}, 0, (uint64_t)(g.T - 1), u_type);
  const int32_t b = t == 0 ? 0 : At4(r.te, t - 1), e = At4(r.te, t);
  s->wb_lo = r.wb_lo;
  s->row_deg = r.row_deg;
  s->row_total = r.row_total;
  s->lim_e = At4(r.lim, t);
  s->lim_b = t == 0 ? 0.f : At4(r.lim, t - 1);
  s->lo = r.row_lo + (uint32_t)b;
  s->deg = e > b ? (uint32_t)(e - b) : 0u;
  s->row_lo = r.row_lo;
  *t_out = t;
  return e > b;
}

// The two neighbour draws on a graph of UNIFORM weights: the running sum after edge m is m + 1
// exactly, so the first sum above r is edge floor(r) of the row (k1_search.h: PivotSample, H1) -
// one id load, no search.  r rounded up to the end of the group: the reference's loop replayed.
__device__ __forceinline__ void UniformSamplePairG2(const GraphView& g, const WbSeg sg0, const WbSeg sg1,
                                                    const int32_t t0, const int32_t t1, const bool live0,
                                                    const bool live1, const double u0, const double u1,
                                                    uint64_t id[2], float w[2], uint32_t m[2]) {
  const double r0 = __dadd_rn(__dmul_rn(u0, (double)__fsub_rn(sg0.lim_e, sg0.lim_b)), (double)sg0.lim_b);
  const double r1 = __dadd_rn(__dmul_rn(u1, (double)__fsub_rn(sg1.lim_e, sg1.lim_b)), (double)sg1.lim_b);
  const bool cold0 = live0 && !((double)sg0.lim_e > r0);
  const bool cold1 = live1 && !((double)sg1.lim_e > r1);
  id[0] = 0; id[1] = 0; w[0] = 0.f; w[1] = 0.f; m[0] = sg0.lo; m[1] = sg1.lo;
  if (live0 && !cold0) {                     // (lim_b = the group's first edge, row-relative, as a float)
    m[0] = sg0.lo + ((uint32_t)r0 - (uint32_t)sg0.lim_b);
    id[0] = g.nbr[m[0]]; w[0] = 1.0f;
  }
  if (live1 && !cold1) {
    m[1] = sg1.lo + ((uint32_t)r1 - (uint32_t)sg1.lim_b);
    id[1] = g.nbr[m[1]]; w[1] = 1.0f;
  }
  if (__ballot(cold0 || cold1) != 0ull) {
#pragma nounroll
    for (int s = 0; s < 2; ++s) {
      if (s == 0 ? cold0 : cold1) {
        // the reference's bisection over the group's running sums (they restart at the row's
        // first edge: positions are row-relative, b = the group's first)
        const WbSeg& sg = s == 0 ? sg0 : sg1;
        const float* nw = g.prefix_w + sg.row_lo;
        const uint32_t b = sg.lo - sg.row_lo;
        const uint32_t mid = (uint32_t)RandomSelect(nw, (uint64_t)b, (uint64_t)(b + sg.deg - 1u), s == 0 ? u0 : u1);
        const uint64_t ci = g.nbr[sg.row_lo + mid];
        const float cw = __fsub_rn(nw[mid], mid == 0u ? 0.f : nw[mid - 1]);
        if (s == 0) { id[0] = ci; w[0] = cw; m[0] = sg.row_lo + mid; }
        else { id[1] = ci; w[1] = cw; m[1] = sg.row_lo + mid; }
      }
    }
  }
}

template <bool UNI = false>
__device__ __forceinline__ void WbSampleTypedPair4(const GraphView& g, const WbRowT4& r, const int32_t mode,
                                                   const int32_t* et, const int32_t k, const uint64_t seed,
                                                   const uint32_t call, const uint64_t node,
                                                   const uint32_t jp, const bool live, const bool two,
                                                   uint64_t id[2], float w[2], uint32_t m[2],
                                                   int32_t tt[2], bool* sentinel) {
  const Philox4 pa = RngBlock(seed, call, kDomainNeighbor, node, 2u * jp);
  const Philox4 pb = RngBlock(seed, call, kDomainNeighbor, node, 2u * jp + 1u);
  WbSeg sg0, sg1;
  sg0.wb_lo = 0; sg0.row_deg = 0; sg0.lo = 0; sg0.deg = 0; sg0.row_total = 0.f; sg0.lim_b = 0.f; sg0.lim_e = 0.f; sg0.row_lo = 0;
  sg1 = sg0;
  tt[0] = -1; tt[1] = -1;
  bool ok0 = false, ok1 = false;
  if (live) {
    ok0 = WbTypedSeg4(g, r, mode, et, k, UnitFromWords(pa.w[0], pa.w[1]), &tt[0], &sg0);
    if (two) ok1 = WbTypedSeg4(g, r, mode, et, k, UnitFromWords(pb.w[0], pb.w[1]), &tt[1], &sg1);
    else sg1 = sg0;
  }
  if (UNI) UniformSamplePairG2(g, sg0, sg1, tt[0], tt[1], ok0, ok1, UnitFromWords(pa.w[2], pa.w[3]),
                               UnitFromWords(pb.w[2], pb.w[3]), id, w, m);
  else WbSamplePairG2<true>(g, sg0, sg1, tt[0], tt[1], ok0, ok1, UnitFromWords(pa.w[2], pa.w[3]),
                            UnitFromWords(pb.w[2], pb.w[3]), id, w, m);
  *sentinel = live && (!ok0 || (two && !ok1));
  if (live && !ok0) { id[0] = 0; w[0] = 0.f; m[0] = r.row_lo; tt[0] = 0; }
  if (live && two && !ok1) { id[1] = 0; w[1] = 0.f; m[1] = r.row_lo; tt[1] = 0; }
}

struct FanoutLeanLds {
  uint32_t o_sid, o_c1, o_slotid, o_mask, o_sw, o_w1, o_st, o_slot, o_rvalid, o_t1, o_t2, bytes;
};
__host__ __device__ inline FanoutLeanLds FanoutLeanLayout(int32_t gr, int32_t c1, int32_t c2,
                                                          int32_t cap, bool typed = false) {
  FanoutLeanLds L;
  const uint32_t p = (uint32_t)gr * (uint32_t)c1;
  const uint32_t s = (uint32_t)cap * (uint32_t)c2;
  uint32_t o = 0;
  L.o_sid = o; o += s * 8;                  // u64 [cap][c2]  finished hop-2 rows: ids
  L.o_c1 = o; o += (p + (p & 1)) * 8;       // u64 [gr][c1]   hop-1 ids (0 for a row without samples)
  L.o_slotid = o; o += p * 8;               // u64 [slots]    the child of a slot
  L.o_mask = o; o += (uint32_t)gr * 8;      // u64 [gr]       drawn edge offsets of a root
  L.o_sw = o; o += s * 4;                   // f32 [cap][c2]  ... weights
  L.o_w1 = o; o += p * 4;                   // f32 [gr][c1]
  L.o_st = o; o += (uint32_t)cap * 4;       // i32 [cap]      ... type (or -1)
  L.o_slot = o; o += (p * 2 + 3) & ~3u;     // u16 [gr][c1]   slot of the sample's child
  L.o_rvalid = o; o += ((uint32_t)gr + 3) & ~3u;
  L.o_t1 = o; L.o_t2 = o;
  if (typed) {                              // i8: the type of every sample (WB == 3)
    o += (p + 3) & ~3u;                     //   [gr][c1]
    L.o_t2 = o; o += (s + 3) & ~3u;         //   [cap][c2]
  }
  L.bytes = (o + 15) & ~15u;
  return L;
}

// WPS: waves per SIMD the register allocation targets (6: 80 VGPRs, nothing spilled; 8: 64
// with ~17 registers spilled - 140 MB of scratch stores and as much again re-read per
// step, profiles/r3_fl_v2_pmc.json).  a.dbg (measurement only): per tile, s_memtime at
// the phase boundaries.
// WB: 1 = draws through the weight-bucket index (WbSamplePair) instead of the pivot levels;
// 2 = the same on graphs with several edge-type groups / hashed ids (WbSamplePairG: one listed
// type per hop, no neighbour id 0); 3 = ... and hops that list several types (a type draw per
// sample, WbSampleTypedPair; at most 127 types); 4 = 3 with the row record in registers (at most 4
// type groups); 5 = 4 on a graph of uniform weights (the neighbour draw is an index computation);
// 6 = 2 on a graph of uniform weights.
template <bool WIDE, int WPS, bool UNIFORM = false, int WB = 0>
__global__ __launch_bounds__(256, WPS) void SampleFanoutLeanKernel(
    const FanoutLocalArgs a) {
  extern __shared__ __align__(16) uint8_t fl_smem[];
  const int lane = threadIdx.x & 63;
  const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int waves_per_block = blockDim.x >> 6;
  const FanoutLeanLds L = FanoutLeanLayout(a.gr, a.c1, a.c2, a.cap, WB >= 3 && WB <= 5);
  uint8_t* base = fl_smem + (size_t)wave_in_block * a.wave_lds;
  uint64_t* s_sid = reinterpret_cast<uint64_t*>(base + L.o_sid);
  uint64_t* s_c1 = reinterpret_cast<uint64_t*>(base + L.o_c1);
  uint64_t* s_slotid = reinterpret_cast<uint64_t*>(base + L.o_slotid);
  unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(base + L.o_mask);
  float* s_sw = reinterpret_cast<float*>(base + L.o_sw);
  float* s_w1 = reinterpret_cast<float*>(base + L.o_w1);
  int32_t* s_st = reinterpret_cast<int32_t*>(base + L.o_st);
  uint16_t* s_slot = reinterpret_cast<uint16_t*>(base + L.o_slot);
  uint8_t* s_rvalid = base + L.o_rvalid;
  int8_t* s_t1 = reinterpret_cast<int8_t*>(base + L.o_t1);      // (WB == 3)
  int8_t* s_t2 = reinterpret_cast<int8_t*>(base + L.o_t2);
  // (no local copy of the view: its level offsets are indexed by a runtime level, and a
  // private copy indexed that way lives in scratch memory)
  const GraphView& g = a.g;
  const uint32_t c1 = (uint32_t)a.c1, c2 = (uint32_t)a.c2, c12 = c1 * c2;
  const uint32_t hp1 = (c1 + 1u) >> 1, hp2 = c2 >> 1;     // pairs per row (c2 is even)
  const uint32_t gr = (uint32_t)a.gr, cap = (uint32_t)a.cap;
  const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int64_t n_tiles = (a.n + gr - 1) / gr;
  const int64_t wave0 = (int64_t)blockIdx.x * waves_per_block + wave_in_block;
  const int64_t wave_stride = (int64_t)gridDim.x * waves_per_block;
  for (int64_t tile = wave0; tile < n_tiles; tile += wave_stride) {
    const int64_t r0 = tile * gr;
    const uint32_t nr = (uint32_t)(a.n - r0 < (int64_t)gr ? a.n - r0 : (int64_t)gr);
    const uint32_t p1 = nr * c1, p2 = nr * c12;
    const int64_t out1 = r0 * (int64_t)c1, out2 = out1 * (int64_t)c2;
    const uint32_t tile_call = TileCallId(a, r0);
    if ((uint32_t)lane < gr) s_mask[lane] = 0ull;
    WaveSync();
    unsigned long long t_s[6] = {0, 0, 0, 0, 0, 0};
    if (EG_FL_DBG(a)) t_s[0] = __builtin_readcyclecounter();
    // ---- P1: hop 1, a lane per pair of samples -----------------------------------
    bool by_edge = true;                  // every root of the tile has <= 64 edges
    const uint32_t t1n = nr * hp1;
    for (uint32_t b = 0; b < t1n; b += 64) {
      const uint32_t tk = b + lane;
      const bool in = tk < t1n;
      const uint32_t q = a.div_h1(tk);
      const uint32_t jp = tk - q * hp1;
      uint32_t lo = 0;
      int32_t deg = 0;
      float total = 0.f;
      uint64_t node = 0;
      WbRec wr{0u, 0u, 0u, 0.f};
      WbSeg ws;
      if (WB == 2 || WB == 6) { ws.wb_lo = 0; ws.row_deg = 0; ws.lo = 0; ws.deg = 0; ws.row_total = 0.f; ws.lim_b = 0.f; ws.lim_e = 0.f; ws.row_lo = 0; }
      WbRowT wt;
      WbRowT4 w4;
      if (WB == 3) { wt.hd = nullptr; wt.te = nullptr; wt.lim = nullptr; wt.tsum = nullptr; wt.row = -1; wt.row_lo = 0; wt.row_deg = 0; wt.valid = false; }
      if (WB == 4 || WB == 5) { w4.wb_lo = 0; w4.row_lo = 0; w4.row = -1; w4.row_deg = 0; w4.row_total = 0.f; w4.valid = false;
                     for (int i = 0; i < 4; ++i) { w4.te[i] = 0; w4.lim[i] = 0.f; w4.ts[i] = 0.f; } }
      if (in) {
        node = a.roots[r0 + q];
        if (WB == 4 || WB == 5) {
          LoadWbRowT4(g, node, a.type_mode, a.et1, a.k, &w4);
          lo = w4.row_lo; deg = w4.valid ? (int32_t)w4.row_deg : 0;
        } else if (WB == 3) {
          LoadWbRowT(g, node, a.type_mode, a.et1, a.k, &wt);
          lo = wt.row_lo; deg = wt.valid ? (int32_t)wt.row_deg : 0;
        } else if (WB == 2 || WB == 6) {
          LoadWbSeg(g, node, a.t1, &ws);
          lo = ws.lo; deg = (int32_t)ws.deg;
        } else {
          const int64_t row = LeanFindRow(g, node);
          if (row >= 0) {
            if (WB) {
              wr = g.wrec[row];
              lo = wr.lo; deg = (int32_t)wr.deg; total = wr.total;
            } else {
              const uint4 rec = *reinterpret_cast<const uint4*>(g.row_meta + row * 16);
              lo = rec.x; deg = (int32_t)rec.z; total = __uint_as_float(rec.w);
            }
          }
        }
      }
      const bool live = in && deg > 0 && !EG_FL_ABLATE(a, 8);
      uint64_t id[2]; float w[2]; uint32_t m[2];
      int32_t tt[2] = {a.t1, a.t1};
      bool sentinel = false;
      Philox4 pb;
      if (WB == 4 || WB == 5) WbSampleTypedPair4<WB == 5>(g, w4, a.type_mode, a.et1, a.k, a.seed, tile_call, node, jp, live,
                                      2u * jp + 1u < c1, id, w, m, tt, &sentinel);
      else if (WB == 3) WbSampleTypedPair(g, wt, a.type_mode, a.et1, a.k, a.seed, tile_call, node, jp, live,
                                          2u * jp + 1u < c1, id, w, m, tt, &sentinel);
      else pb = RngBlock(a.seed, tile_call, kDomainNeighbor, node, jp);
      if (WB >= 3 && WB <= 5) {}
      else if (WB == 6) UniformSamplePairG2(g, ws, ws, a.t1, a.t1, live, live && 2u * jp + 1u < c1,
                                            UnitFromWords(pb.w[0], pb.w[1]), UnitFromWords(pb.w[2], pb.w[3]), id, w, m);
      else if (WB == 2) WbSamplePairG(g, ws, a.t1, live, UnitFromWords(pb.w[0], pb.w[1]),
                                 UnitFromWords(pb.w[2], pb.w[3]), id, w, m);
      else if (WB) WbSamplePair(g, wr, live, UnitFromWords(pb.w[0], pb.w[1]),
                                UnitFromWords(pb.w[2], pb.w[3]), id, w, m);
      else if (UNIFORM) LeanSamplePairUniform(g, lo, deg, total, live, UnitFromWords(pb.w[0], pb.w[1]),
                                         UnitFromWords(pb.w[2], pb.w[3]), id, w, m);
      else LeanSamplePair(g, lo, deg, total, live, UnitFromWords(pb.w[0], pb.w[1]),
                          UnitFromWords(pb.w[2], pb.w[3]), id, w, m);
      by_edge = by_edge && __ballot(in && (deg > 64 || sentinel)) == 0ull;
      if (in) {
        const uint32_t j0 = 2u * jp;
        const uint32_t e0 = q * c1 + j0;
        s_c1[e0] = live ? id[0] : 0;        // a row without samples hands node id 0 on
        s_w1[e0] = live ? w[0] : 0.f;
        if (WB >= 3 && WB <= 5) s_t1[e0] = (int8_t)(live ? tt[0] : -1);
        unsigned long long bits = live ? 1ull << ((m[0] - lo) & 63u) : 1ull;
        if (j0 + 1u < c1) {
          s_c1[e0 + 1] = live ? id[1] : 0;
          s_w1[e0 + 1] = live ? w[1] : 0.f;
          if (WB >= 3 && WB <= 5) s_t1[e0 + 1] = (int8_t)(live ? tt[1] : -1);
          if (live) bits |= 1ull << ((m[1] - lo) & 63u);
          // the slot pass below needs the edge of every sample: park it in s_slot
          s_slot[e0 + 1] = (uint16_t)(live ? (m[1] - lo) & 63u : 0u);
        }
        s_slot[e0] = (uint16_t)(live ? (m[0] - lo) & 63u : 0u);
        atomicOr(&s_mask[q], bits);
        if (jp == 0) s_rvalid[q] = live ? 1 : 0;
      }
    }
    WaveSync();
    if (EG_FL_DBG(a)) t_s[1] = __builtin_readcyclecounter();
    // ---- P2: slots of the distinct children -----------------------------------------
    uint32_t n_slots = 0;
    if (by_edge) {
      for (uint32_t b = 0; b < p1; b += 64) {
        const uint32_t tk = b + lane;
        if (tk < p1) {
          const uint32_t q = a.div_c1(tk);
          uint32_t sbase = 0;
          for (uint32_t x = 0; x < q; ++x) sbase += (uint32_t)__popcll(s_mask[x]);
          const uint32_t off = s_slot[tk];
          const uint32_t slot = sbase + (uint32_t)__popcll(s_mask[q] & ((1ull << off) - 1ull));
          s_slot[tk] = (uint16_t)slot;        // own entry only: nobody else reads it
          s_slotid[slot] = s_c1[tk];          // every sample of the slot writes the same id
        }
      }
      for (uint32_t x = 0; x < nr; ++x) n_slots += (uint32_t)__popcll(s_mask[x]);
    } else {
      // some root has more than 64 edges: first occurrence by id among the root's samples
      for (uint32_t b = 0; b < p1; b += 64) {
        const uint32_t tk = b + lane;
        const bool in = tk < p1;
        const uint32_t q = a.div_c1(tk);
        const uint32_t j = tk - q * c1;
        const uint64_t mine = in ? s_c1[tk] : 0;
        uint32_t first = j;
        const uint64_t* row = s_c1 + q * c1;
        for (uint32_t i = 0; i + 1 < c1; ++i) {          // wave-uniform trip count
          const uint64_t v = in ? row[i] : 0;
          if (in && i < j && first == j && v == mine) first = i;
        }
        const bool rep = in && first == j;
        const uint64_t bal = __ballot(rep);
        if (rep) {
          const uint32_t slot = n_slots + (uint32_t)__popcll(bal & lt_mask);
          s_slot[tk] = (uint16_t)slot;
          s_slotid[slot] = mine;
        } else if (in) {
          s_slot[tk] = (uint16_t)(0x8000u | first);
        }
        n_slots += (uint32_t)__popcll(bal);
      }
      WaveSync();
      for (uint32_t b = 0; b < p1; b += 64) {
        const uint32_t tk = b + lane;
        if (tk < p1) {
          const uint32_t v = s_slot[tk];
          if (v & 0x8000u) {
            const uint32_t q = a.div_c1(tk);
            s_slot[tk] = s_slot[q * c1 + (v & 0x7FFFu)];
          }
        }
      }
    }
    WaveSync();
    if (a.row_index != nullptr) {
      for (uint32_t b = 0; b < p1; b += 64) {
        const uint32_t tk = b + lane;
        if (tk < p1) a.row_index[out1 + tk] = (uint32_t)(out1 + s_slot[tk]);
      }
    }
    if (EG_FL_DBG(a)) t_s[2] = __builtin_readcyclecounter();
    // ---- P3 / P4 per chunk of `cap` slots -------------------------------------------
    for (uint32_t s0 = 0; s0 < n_slots; s0 += cap) {
      const uint32_t ns = n_slots - s0 < cap ? n_slots - s0 : cap;
      const uint32_t t2n = ns * hp2;
      for (uint32_t b = 0; b < t2n; b += 64) {
        const uint32_t tk = b + lane;
        const bool in = tk < t2n;
        const uint32_t sl = a.div_h2(tk);
        const uint32_t xp = tk - sl * hp2;
        uint32_t lo = 0;
        int32_t deg = 0;
        float total = 0.f;
        uint64_t node = 0;
        WbRec wr{0u, 0u, 0u, 0.f};
        WbSeg ws;
        if (WB == 2 || WB == 6) { ws.wb_lo = 0; ws.row_deg = 0; ws.lo = 0; ws.deg = 0; ws.row_total = 0.f; ws.lim_b = 0.f; ws.lim_e = 0.f; ws.row_lo = 0; }
        WbRowT wt;
        WbRowT4 w4;
        if (WB == 3) { wt.hd = nullptr; wt.te = nullptr; wt.lim = nullptr; wt.tsum = nullptr; wt.row = -1; wt.row_lo = 0; wt.row_deg = 0; wt.valid = false; }
        if (WB == 4 || WB == 5) { w4.wb_lo = 0; w4.row_lo = 0; w4.row = -1; w4.row_deg = 0; w4.row_total = 0.f; w4.valid = false;
                       for (int i = 0; i < 4; ++i) { w4.te[i] = 0; w4.lim[i] = 0.f; w4.ts[i] = 0.f; } }
        if (in) {
          node = s_slotid[s0 + sl];
          if (WB == 4 || WB == 5) {
            LoadWbRowT4(g, node, a.type_mode, a.et2, a.k, &w4);
            lo = w4.row_lo; deg = w4.valid ? (int32_t)w4.row_deg : 0;
          } else if (WB == 3) {
            LoadWbRowT(g, node, a.type_mode, a.et2, a.k, &wt);
            lo = wt.row_lo; deg = wt.valid ? (int32_t)wt.row_deg : 0;
          } else if (WB == 2 || WB == 6) {
            LoadWbSeg(g, node, a.t2, &ws);
            lo = ws.lo; deg = (int32_t)ws.deg;
          } else {
            const int64_t row = LeanFindRow(g, node);
            if (row >= 0) {
              if (WB) {
                wr = g.wrec[row];
                lo = wr.lo; deg = (int32_t)wr.deg; total = wr.total;
              } else {
                const uint4 rec = *reinterpret_cast<const uint4*>(g.row_meta + row * 16);
                lo = rec.x; deg = (int32_t)rec.z; total = __uint_as_float(rec.w);
              }
            }
          }
        }
        const bool live = in && deg > 0 && !EG_FL_ABLATE(a, 2);
        uint64_t id[2]; float w[2]; uint32_t m[2];
        int32_t tt[2] = {a.t2, a.t2};
        bool sentinel = false;
        Philox4 pb;
        if (WB == 4 || WB == 5) WbSampleTypedPair4<WB == 5>(g, w4, a.type_mode, a.et2, a.k, a.seed, tile_call + 1u, node, xp, live,
                                        true, id, w, m, tt, &sentinel);
        else if (WB == 3) WbSampleTypedPair(g, wt, a.type_mode, a.et2, a.k, a.seed, tile_call + 1u, node, xp, live,
                                            true, id, w, m, tt, &sentinel);
        else pb = RngBlock(a.seed, tile_call + 1u, kDomainNeighbor, node, xp);
        if (WB >= 3 && WB <= 5) {}
        else if (WB == 6) UniformSamplePairG2(g, ws, ws, a.t2, a.t2, live, live, UnitFromWords(pb.w[0], pb.w[1]),
                                              UnitFromWords(pb.w[2], pb.w[3]), id, w, m);
        else if (WB == 2) WbSamplePairG(g, ws, a.t2, live, UnitFromWords(pb.w[0], pb.w[1]),
                                   UnitFromWords(pb.w[2], pb.w[3]), id, w, m);
        else if (WB) WbSamplePair(g, wr, live, UnitFromWords(pb.w[0], pb.w[1]),
                                  UnitFromWords(pb.w[2], pb.w[3]), id, w, m);
        else if (UNIFORM) LeanSamplePairUniform(g, lo, deg, total, live, UnitFromWords(pb.w[0], pb.w[1]),
                                           UnitFromWords(pb.w[2], pb.w[3]), id, w, m);
        else LeanSamplePair(g, lo, deg, total, live, UnitFromWords(pb.w[0], pb.w[1]),
                            UnitFromWords(pb.w[2], pb.w[3]), id, w, m, EG_FL_ABLATE_BITS(a));
        if (in) {
          fl_u64x2 iv;
          iv.x = live ? id[0] : (uint64_t)a.default_node;
          iv.y = live ? id[1] : (uint64_t)a.default_node;
          *reinterpret_cast<fl_u64x2*>(s_sid + sl * c2 + 2u * xp) = iv;
          *reinterpret_cast<float2*>(s_sw + sl * c2 + 2u * xp) =
              make_float2(live ? w[0] : 0.f, live ? w[1] : 0.f);
          if (xp == 0) s_st[sl] = live ? a.t2 : -1;
          if (WB >= 3 && WB <= 5) {
            s_t2[sl * c2 + 2u * xp] = (int8_t)(live ? tt[0] : -1);
            s_t2[sl * c2 + 2u * xp + 1u] = (int8_t)(live ? tt[1] : -1);
          }
        }
      }
      WaveSync();
      if (EG_FL_DBG(a) && s0 == 0) t_s[3] = __builtin_readcyclecounter();
      // -- P4, (unique rows, index) form: the chunk's rows as they are, once ------------
      if (a.row_index != nullptr) {
        const int64_t row0 = (out1 + (int64_t)s0) * (int64_t)c2;     // first sample of row r0 * c1 + s0
        for (uint32_t b = 0; b < ns * c2; b += 128) {
          const uint32_t e = b + 2 * lane;
          if (e < ns * c2) {
            *reinterpret_cast<fl_u64x2*>(a.id2 + row0 + e) = *reinterpret_cast<const fl_u64x2*>(s_sid + e);
            *reinterpret_cast<float2*>(a.w2 + row0 + e) = *reinterpret_cast<const float2*>(s_sw + e);
            const int32_t tv = s_st[a.div_c2(e)];
            *reinterpret_cast<int2*>(a.ty2 + row0 + e) =
                (WB >= 3 && WB <= 5) ? make_int2((int32_t)s_t2[e], (int32_t)s_t2[e + 1]) : make_int2(tv, tv);
          }
        }
      } else
      // -- P4: copy the finished rows to the positions that asked for them ------------
      if (!EG_FL_ABLATE(a, 1)) {
      for (uint32_t b = 0; b < p2; b += 128) {
        const uint32_t p = b + 2 * lane;
        if (p < p2) {
          const uint32_t gj = a.div_c2(p);
          const uint32_t x = p - gj * c2;
          const uint32_t sl = (uint32_t)s_slot[gj] - s0;
          if (sl < ns) {
            const fl_u64x2 iv = *reinterpret_cast<const fl_u64x2*>(s_sid + sl * c2 + x);
            if (EG_FL_ABLATE(a, 256)) __builtin_nontemporal_store(iv, reinterpret_cast<fl_u64x2*>(a.id2 + out2 + p));
            else *reinterpret_cast<fl_u64x2*>(a.id2 + out2 + p) = iv;
            if (!WIDE) {
              *reinterpret_cast<float2*>(a.w2 + out2 + p) =
                  *reinterpret_cast<const float2*>(s_sw + sl * c2 + x);
              const int32_t tv = s_st[sl];
              *reinterpret_cast<int2*>(a.ty2 + out2 + p) =
                  (WB >= 3 && WB <= 5) ? make_int2((int32_t)s_t2[sl * c2 + x], (int32_t)s_t2[sl * c2 + x + 1]) : make_int2(tv, tv);
            }
          }
        }
      }
      if (WIDE) {
        for (uint32_t b = 0; b < p2; b += 256) {
          const uint32_t p = b + 4 * lane;
          if (p < p2) {
            const uint32_t gja = a.div_c2(p);
            const uint32_t xa = p - gja * c2;
            const uint32_t sla = (uint32_t)s_slot[gja] - s0;
            const bool ina = sla < ns;
            const bool hasb = p + 2 < p2;
            uint32_t gjb = gja, xb = xa + 2;
            if (xb >= c2) { xb -= c2; ++gjb; }
            const uint32_t slb = hasb ? (uint32_t)s_slot[gjb] - s0 : 0xFFFFFFFFu;
            const bool inb = hasb && slb < ns;
            float2 wa = make_float2(0.f, 0.f), wb = wa;
            int32_t ta = -1, tb = -1, ta2 = -1, tb2 = -1;
            if (ina) { wa = *reinterpret_cast<const float2*>(s_sw + sla * c2 + xa); ta = s_st[sla]; ta2 = ta; }
            if (inb) { wb = *reinterpret_cast<const float2*>(s_sw + slb * c2 + xb); tb = s_st[slb]; tb2 = tb; }
            if (WB >= 3 && WB <= 5) {
              if (ina) { ta = (int32_t)s_t2[sla * c2 + xa]; ta2 = (int32_t)s_t2[sla * c2 + xa + 1]; }
              if (inb) { tb = (int32_t)s_t2[slb * c2 + xb]; tb2 = (int32_t)s_t2[slb * c2 + xb + 1]; }
            }
            float* wp = a.w2 + out2 + p;
            int32_t* tp = a.ty2 + out2 + p;
            if (ina && inb && EG_FL_ABLATE(a, 256)) {
              typedef float fl_f4 __attribute__((ext_vector_type(4)));
              typedef int fl_i4 __attribute__((ext_vector_type(4)));
              const fl_f4 wv4 = {wa.x, wa.y, wb.x, wb.y};
              const fl_i4 tv4 = {ta, ta2, tb, tb2};
              __builtin_nontemporal_store(wv4, reinterpret_cast<fl_f4*>(wp));
              __builtin_nontemporal_store(tv4, reinterpret_cast<fl_i4*>(tp));
            } else if (ina && inb) {
              *reinterpret_cast<float4*>(wp) = make_float4(wa.x, wa.y, wb.x, wb.y);
              *reinterpret_cast<int4*>(tp) = make_int4(ta, ta2, tb, tb2);
            } else if (ina) {
              *reinterpret_cast<float2*>(wp) = wa;
              *reinterpret_cast<int2*>(tp) = make_int2(ta, ta2);
            } else if (inb) {
              *reinterpret_cast<float2*>(wp + 2) = wb;
              *reinterpret_cast<int2*>(tp + 2) = make_int2(tb, tb2);
            }
          }
        }
      }
      }
      WaveSync();              // the next chunk rewrites the slot rows
    }
    if (EG_FL_DBG(a)) t_s[4] = __builtin_readcyclecounter();
    // ---- hop-1 outputs (contiguous over the tile) -----------------------------------
    for (uint32_t b = 0; b < p1 && !EG_FL_ABLATE(a, 4); b += 64) {
      const uint32_t tk = b + lane;
      if (tk < p1) {
        const uint32_t q = a.div_c1(tk);
        const bool ok = s_rvalid[q] != 0;
        a.id1[out1 + tk] = ok ? s_c1[tk] : (uint64_t)a.default_node;
        a.w1[out1 + tk] = s_w1[tk];
        a.ty1[out1 + tk] = (WB >= 3 && WB <= 5) ? (int32_t)s_t1[tk] : ok ? a.t1 : -1;
      }
    }
    WaveSync();
#ifdef EULER_GPU_MEASURE
    if (a.dbg != nullptr) {
      __builtin_amdgcn_s_waitcnt(0);          // the stores have left the wave's queue
      t_s[5] = __builtin_readcyclecounter();
      if (lane == 0) {
        unsigned long long* d = a.dbg + tile * 8;
        for (int x = 0; x < 6; ++x) d[x] = t_s[x];
        d[6] = n_slots;
        d[7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
      }
    }
#endif
  }
}

}  // namespace euler_gpu

#endif  // EULER_AMD_CSRC_FANOUT_LOCAL_H_
