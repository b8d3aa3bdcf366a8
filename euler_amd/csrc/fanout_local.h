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
  uint64_t* id1; float* w1; int32_t* ty1; uint8_t* mask0;
  uint64_t* id2; float* w2; int32_t* ty2; uint8_t* mask1;
};

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
// in its record, weighted, identity id map, no neighbour id 0, no row_inline lines - and the
// kernel is compiled with those as constants (fewer live registers: 8 waves per SIMD).
// WPS: waves per SIMD the register allocation targets (8: 64 VGPRs with a few cold spills
// at the phase boundaries; 5: 96, none).
template <bool WIDE, bool PLAIN, int WPS>
__global__ __launch_bounds__(256, WPS) void SampleFanoutLocalKernel(
    const FanoutLocalArgs a_in) {
  FanoutLocalArgs a = a_in;
  if (PLAIN) {
    a.g.T = 1; a.g.meta_stride = 16; a.g.total_in_meta = 1; a.g.uniform_w = 0;
    a.g.inline_k = 0; a.g.map_mode = 0; a.g.has_zero_nbr = 0; a.g.monotone = 1;
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
      const uint32_t call = a.call_id + (uint32_t)phase;
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

}  // namespace euler_gpu

#endif  // EULER_AMD_CSRC_FANOUT_LOCAL_H_
