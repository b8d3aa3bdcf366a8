// K1 "row" kernel (gfx950): ONE LANE PER ROOT.
//
// Round 1's kernels give every SAMPLE a lane: the `count` lanes of a root repeat
// the root's dependent chain (root id -> row record -> pivot window -> block
// line -> id) and a wave of 64 samples carries ~9 vector-memory instructions.
// On the metric's first hop (131 072 uniformly drawn roots x 25) that chain,
// not bytes or lines, is the cost: 67 us against ~25 us of traffic.
//
// Here a lane owns a root.  Most rows are short (94 % of uniformly drawn roots
// of the metric graph have <= 10 edges) and lie inside ONE or TWO consecutive
// 128-byte EdgeBlocks: the lane fetches the row record, then the <= 20 running
// sums of those blocks (six independent 16-byte loads), and draws all `count`
// samples from registers - no further memory instruction in the search.  Every
// draw is the index Node::SampleNeighbor's RandomSelect returns
// (core/graph/node.cc:150-158, common/compact_weighted_collection.h:30-52): the
// first m of the segment with nw[m] > r (non-decreasing rows; Q3 draws replay
// the reference loop).  The samples of the 64 roots of a wave are staged in LDS
// (slot index + weight) and written by the whole wave in OUTPUT order: ids are
// fetched from the block lines (L1 / L2 hits: the lines were just read) and
// leave as 16-byte stores, weights and types as 8-byte stores.
//
// Rows that span more than two blocks ("slow" roots: 6 % of the first hop, nearly
// all roots of a later hop) are not sampled here: the wave appends them to one of
// kSlowShards device-side queues and SampleNeighborSlowKernel, launched right
// behind, gives each of their samples a lane and the block-pivot search of
// k1_search.h - balanced over the chip, where doing them inside their tile made
// the tile with the most hubs the kernel's critical path.  The queues are sharded
// and their counters come in two sets used alternately (a call clears the set the
// NEXT call will use): a single counter made 2 048 waves queue behind one another
// on a same-address atomic (~9 ns each, returned value awaited: +18 us), and a
// "last workgroup clears" counter in the slow kernel did it a second time.
#ifndef EULER_AMD_CSRC_K1_ROW_H_
#define EULER_AMD_CSRC_K1_ROW_H_

#include <hip/hip_runtime.h>

#include "k1_args.h"
#include "k1_search.h"

namespace euler_gpu {

constexpr int kRowTile = 64;                        // roots per wave (= workgroup)
constexpr int kRowMaxCount = 64;
constexpr int kRowSlots = 2 * kEdgesPerBlock;       // sums a lane keeps in registers
constexpr int kSlowShards = 64;                     // queues of the slow roots (tile % 64)
// scratch: counters [2 sets][kSlowShards] u32, then kSlowShards queues of
// `slow_cap` root indices each (slow_cap >= 64 * tiles per shard)
constexpr int kSlowCounterWords = 2 * kSlowShards;
constexpr int64_t kRowMinSamples = 1 << 20;         // below: one lane per sample

// dynamic LDS of one wave:
//   sums [kRowSlots + 1][64] f32   (slot-major: lane l reads bank l % 32 whatever
//                                   its slot - conflict-free dynamic indexing)
//   ids  [64][kRowSlots]     u64   the neighbour ids of every root's (<= 2) blocks
//   m    [64 * count]        u8    slot of every sample, 0xFF = written by the Q3 replay
//   rs   [64]                u8    slot of the row's first edge (0xFF: in an earlier block)
//   flag [64]                u8    0 = sampled here, 1 = no samples (default row),
//                                  2 = slow root (SampleNeighborSlowKernel writes it)
// The write phase needs NO global load: a first version fetched the ids from the
// block lines again, 25 dependent load -> store iterations per wave (vector-memory
// operations retire in order on gfx950, so every iteration also waited for the
// previous iteration's stores): 95 us for the metric's first hop.
__host__ __device__ inline size_t RowKernelLdsBytes(int32_t count) {
  const size_t b = (size_t)(kRowSlots + 1) * 64 * 4 + (size_t)64 * kRowSlots * 8 +
                   (size_t)64 * count + 64 + 64;
  return (b + 15) & ~(size_t)15;
}

template <bool TF_LAYOUT>
__global__ __launch_bounds__(64) void SampleNeighborRowKernel(const SampleNbArgs a) {
  extern __shared__ __align__(16) uint8_t row_smem[];
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  int64_t n_roots;
  if (!DedupGate(a, &n_roots)) return;
  const int lane = threadIdx.x;
  const int32_t count = a.count;
  float* s_sum = reinterpret_cast<float*>(row_smem);
  uint64_t* s_id = reinterpret_cast<uint64_t*>(s_sum + (kRowSlots + 1) * 64);
  uint8_t* s_m = reinterpret_cast<uint8_t*>(s_id + 64 * kRowSlots);
  uint8_t* s_rs = s_m + 64 * count;
  uint8_t* s_flag = s_rs + 64;
  const int32_t t = a.et[0];
  const bool mark = a.mark_owner != nullptr;
  const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const double kInf = __builtin_huge_val();
  const int64_t tiles = (n_roots + kRowTile - 1) / kRowTile;
  // the counters of the stream's NEXT call
  if (blockIdx.x == 0 && lane < kSlowShards) a.slow_count_next[lane] = 0;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    // ---- P1: root id -> row record -> the sums and ids of the row's (<= 2) blocks
    const int64_t r = tile * kRowTile + lane;
    const bool live = r < n_roots;
    uint64_t node = 0;
    Segment sg;
    bool valid = false;
    if (live) {
      node = a.roots[r];
      if (a.root_mask != nullptr && a.root_mask[r / a.root_group]) node = 0;
      valid = LoadSegment<true>(a.g, FindRow(a.g, node), t, &sg);
    }
    // a row of <= kInlineEdges edges comes whole with its record (row_inline): slots
    // 0 .. e of a virtual block that starts at the row
    const bool inl = valid && sg.inl != nullptr;
    int64_t blk_lo = 0;
    int32_t i_lo = 0, i_hi = 0;
    if (inl) {
      i_hi = sg.e;
    } else if (valid) {
      blk_lo = sg.lo / kEdgesPerBlock;
      i_lo = (int32_t)(sg.lo - blk_lo * kEdgesPerBlock);
      i_hi = (int32_t)(sg.hi - blk_lo * kEdgesPerBlock);
    }
    const bool fast = valid && i_hi < kRowSlots;
    const bool slow = valid && !fast;
    double vd[kRowSlots - 1];          // compare keys of slots 0 .. 18, padded
    if (fast) {
      const EdgeBlock* bk = a.g.blk + blk_lo;
      const bool two = !inl && i_hi >= kEdgesPerBlock;
      float4 a0, a1, a2;
      u64x2 ia[kEdgesPerBlock / 2], ib[kEdgesPerBlock / 2];
      if (inl) {
        const float* ipw = reinterpret_cast<const float*>(sg.inl + 16);
        const uint64_t* inb = reinterpret_cast<const uint64_t*>(sg.inl + 56);
        a0 = *reinterpret_cast<const float4*>(ipw);
        a1 = *reinterpret_cast<const float4*>(ipw + 4);
        a2 = make_float4(ipw[8], 0.f, 0.f, 0.f);          // slot 8; slot 9 unused; sum before slot 0 = 0
#pragma unroll
        for (int q = 0; q < kEdgesPerBlock / 2 - 1; ++q) {
          ia[q].x = inb[2 * q];
          ia[q].y = inb[2 * q + 1];
        }
        ia[kEdgesPerBlock / 2 - 1].x = inb[kInlineEdges - 1];
        ia[kEdgesPerBlock / 2 - 1].y = 0;
      } else {
        a0 = *reinterpret_cast<const float4*>(bk->pw);
        a1 = *reinterpret_cast<const float4*>(bk->pw + 4);
        a2 = *reinterpret_cast<const float4*>(bk->pw + 8);   // pw[8], pw[9], prev_last
#pragma unroll
        for (int q = 0; q < kEdgesPerBlock / 2; ++q)
          ia[q] = *reinterpret_cast<const u64x2*>(bk->nbr + 2 * q);
      }
      float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0, b2 = b0;
#pragma unroll
      for (int q = 0; q < kEdgesPerBlock / 2; ++q) ib[q] = u64x2{0, 0};
      if (two) {
        b0 = *reinterpret_cast<const float4*>(bk[1].pw);
        b1 = *reinterpret_cast<const float4*>(bk[1].pw + 4);
        b2 = *reinterpret_cast<const float4*>(bk[1].pw + 8);
#pragma unroll
        for (int q = 0; q < kEdgesPerBlock / 2; ++q)
          ib[q] = *reinterpret_cast<const u64x2*>(bk[1].nbr + 2 * q);
      }
      const float v[kRowSlots] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y,
                                  b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y};
      s_sum[lane] = a2.z;                                  // the sum before slot 0
#pragma unroll
      for (int q = 0; q < kRowSlots; ++q) s_sum[(q + 1) * 64 + lane] = v[q];
      u64x2* my_ids = reinterpret_cast<u64x2*>(s_id + lane * kRowSlots);
#pragma unroll
      for (int q = 0; q < kEdgesPerBlock / 2; ++q) {
        my_ids[q] = ia[q];
        my_ids[kEdgesPerBlock / 2 + q] = ib[q];
      }
      // slots before the segment always count (-inf is never > r), slots from the
      // segment's last one on never do: pos = first slot of [i_lo, i_hi] whose sum > r
#pragma unroll
      for (int q = 0; q < kRowSlots - 1; ++q)
        vd[q] = q < i_lo ? -kInf : (q >= i_hi ? kInf : (double)v[q]);
      // `mid ? nw[mid-1] : 0` is row-relative: the slot whose predecessor sum is 0
      const int64_t rs = inl ? 0 : sg.row_ptr - blk_lo * kEdgesPerBlock;
      s_rs[lane] = rs >= 0 ? (uint8_t)rs : (uint8_t)0xFF;
    }
    s_flag[lane] = !valid ? 1 : (slow ? 2 : 0);
    // ---- P2: all `count` draws of a fast root from registers
    if (fast && (a.ablate & 1)) {
      for (int32_t j = 0; j < count; ++j) s_m[lane * count + j] = (uint8_t)i_lo;
    } else if (fast) {
      const float lb = sg.limit_begin, le = sg.limit_end;
      for (int32_t j = 0; j < count; j += 2) {
        Philox4 pb;
        if (a.ablate & 4) { pb.w[0] = pb.w[2] = 0x80000000u + (uint32_t)j; pb.w[1] = pb.w[3] = 0; }
        else pb = RngBlock(a.seed, a.call_id, kDomainNeighbor, node, ((uint32_t)j) >> 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (j + h >= count) break;
          const double u = h ? UnitFromWords(pb.w[2], pb.w[3]) : UnitFromWords(pb.w[0], pb.w[1]);
          const double rr = ScaleDraw(u, lb, le);
          uint8_t mm;
          if (!((double)le > rr)) {
            // Q3: r rounded up to the end of the segment - replay the reference
            const float* nw = a.g.prefix_w + sg.row_ptr;
            const int32_t m = (int32_t)RandomSelect(nw, (uint64_t)sg.b, (uint64_t)sg.e, u);
            const uint64_t id = a.g.nbr[sg.row_ptr + m];
            const int64_t s = r * (int64_t)count + j + h;
            a.out_id[s] = id;
            a.out_w[s] = __fsub_rn(nw[m], m == 0 ? 0.f : nw[m - 1]);
            if (mark) MarkNextHop(a.g, a.mark_owner, id, true, s);
            mm = 0xFF;
          } else {
            int32_t pos = 0;
#pragma unroll
            for (int q = 0; q < kRowSlots - 1; ++q) pos += (vd[q] > rr) ? 0 : 1;
            mm = (uint8_t)pos;
          }
          s_m[lane * count + j + h] = mm;
        }
      }
    }
    // ---- slow roots (more than two blocks): queued for SampleNeighborSlowKernel
    const uint64_t slow_mask = __ballot(slow);
    if (slow_mask != 0) {
      const int32_t shard = (int32_t)(tile % kSlowShards);
      uint32_t base_q = 0;
      if (lane == 0) base_q = atomicAdd(a.slow_count + shard, (uint32_t)__popcll(slow_mask));
      base_q = __shfl(base_q, 0);
      if (slow)
        a.slow_list[(int64_t)shard * a.slow_cap + base_q + (uint32_t)__popcll(slow_mask & lt_mask)] =
            (uint32_t)r;
    }
    if (live && a.out_row_mask != nullptr) a.out_row_mask[r] = valid ? 0 : 1;
    WaveSync();
    // ---- P3: the tile's samples in output order, two per lane; everything comes
    // from LDS, the loop is stores only
    const int64_t left = n_roots - tile * kRowTile;
    const int32_t nt = (int32_t)(left < kRowTile ? left : kRowTile) * count;
    const int64_t base = tile * kRowTile * (int64_t)count;
    for (int32_t e = lane * 2; e < nt && !(a.ablate & 2); e += 128) {
      uint64_t id[2] = {0, 0};
      float wv[2] = {0.f, 0.f};
      int32_t ot[2] = {t, t};
      // have: id and weight are written here; skip: the whole sample belongs to the
      // slow kernel
      bool have[2] = {false, false}, skip[2] = {true, true}, rv[2] = {true, true};
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        if (e + x >= nt) continue;
        const int32_t rl = (e + x) / count;
        const int32_t fl = s_flag[rl];
        if (fl == 2) continue;
        skip[x] = false;
        if (fl == 1) {
          id[x] = TF_LAYOUT ? (uint64_t)a.default_node : 0;
          ot[x] = TF_LAYOUT ? -1 : 0;
          have[x] = true;
          rv[x] = false;
          continue;
        }
        const int32_t mm = s_m[e + x];
        if (mm == 0xFF) continue;                     // id and weight written by the Q3 replay
        id[x] = s_id[rl * kRowSlots + mm];
        const float nw_m = s_sum[(mm + 1) * 64 + rl];
        const float prev = mm == (int32_t)s_rs[rl] ? 0.f : s_sum[mm * 64 + rl];
        wv[x] = __fsub_rn(nw_m, prev);
        have[x] = true;
      }
      const int64_t d = base + e;
      if (have[0] && have[1]) {
        const u64x2 i2 = {id[0], id[1]};
        *reinterpret_cast<u64x2*>(a.out_id + d) = i2;
        *reinterpret_cast<float2*>(a.out_w + d) = make_float2(wv[0], wv[1]);
      } else {
        if (have[0]) { a.out_id[d] = id[0]; a.out_w[d] = wv[0]; }
        if (have[1]) { a.out_id[d + 1] = id[1]; a.out_w[d + 1] = wv[1]; }
      }
      if (!skip[0] && !skip[1]) {
        *reinterpret_cast<int2*>(a.out_t + d) = make_int2(ot[0], ot[1]);
      } else {
        if (!skip[0]) a.out_t[d] = ot[0];
        if (!skip[1]) a.out_t[d + 1] = ot[1];
      }
      if (mark) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
          if (have[x]) MarkNextHop(a.g, a.mark_owner, id[x], rv[x], d + x);
      }
    }
    WaveSync();        // the next tile reuses the staging area
  }
}

// The queued slow roots, one lane per sample (see the head of this file).
template <bool TF_LAYOUT>
__global__ __launch_bounds__(256, kWavesPerSimd) void SampleNeighborSlowKernel(
    const SampleNbArgs a) {
  __shared__ uint32_t s_off[kSlowShards + 1];
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int x = 0; x < kSlowShards; ++x) { s_off[x] = run; run += a.slow_count[x]; }
    s_off[kSlowShards] = run;
  }
  __syncthreads();
  const uint32_t n_slow = s_off[kSlowShards];
  const int32_t count = a.count;
  const int32_t t = a.et[0];
  const int64_t total = (int64_t)n_slow * count;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += stride) {
    const uint32_t q = (uint32_t)(s / count);
    const int32_t j = (int32_t)(s - (int64_t)q * count);
    // shard of queue entry q: the last offset <= q
    int32_t lo = 0, hi = kSlowShards - 1;
    while (lo < hi) {
      const int32_t mid = (lo + hi + 1) >> 1;
      if (s_off[mid] <= q) lo = mid; else hi = mid - 1;
    }
    const int64_t r = (int64_t)a.slow_list[(int64_t)lo * a.slow_cap + (q - s_off[lo])];
    uint64_t node = a.roots[r];
    if (a.root_mask != nullptr && a.root_mask[r / a.root_group]) node = 0;
    Segment sg;
    (void)LoadSegment<true>(a.g, FindRow(a.g, node), t, &sg);     // valid: it was queued
    const Philox4 pb = RngBlock(a.seed, a.call_id, kDomainNeighbor, node, ((uint32_t)j) >> 1);
    const double u = (j & 1) ? UnitFromWords(pb.w[2], pb.w[3]) : UnitFromWords(pb.w[0], pb.w[1]);
    uint64_t id;
    float wv;
    BlockPivotSample(a.g, sg, u, &id, &wv);
    const int64_t d = r * (int64_t)count + j;
    a.out_id[d] = id;
    a.out_w[d] = wv;
    a.out_t[d] = t;
    if (a.mark_owner != nullptr) MarkNextHop(a.g, a.mark_owner, id, true, d);
  }
}

}  // namespace euler_gpu

#endif  // EULER_AMD_CSRC_K1_ROW_H_
