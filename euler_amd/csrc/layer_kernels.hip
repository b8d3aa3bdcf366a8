// Layerwise sampling (GQL sampleLNB without a weight function) and
// SparseGetAdj for gfx950, with their C-ABI entry points.  The per-item logic
// lives in layer_fns.h; the kernels here map items to lanes / waves.
//
//   API_GET_EDGE_SUM_WEIGHT  core/kernels/get_edge_sum_weight_op.cc:33-66
//   API_SAMPLE_ROOT          core/kernels/sample_root_op.cc:33-88
//   API_SAMPLE_L             core/kernels/sample_layer_op.cc:32-72
//   API_SPARSE_GET_ADJ       core/kernels/sparse_get_adj_op.cc:35-92
//   TF SparseGetAdj          tf_euler/kernels/sparse_get_adj_op.cc:43-134
//   TF SampleNeighborLayerwiseWithAdj
//                            tf_euler/kernels/sample_neighbor_layerwise_with_adj_op.cc:56-150
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "layer_fns.h"
#include "local_layer_host.h"
#include "wave_sums.h"

namespace euler_gpu {

// euler_gpu_set_tuning key 15.  API_SAMPLE_ROOT's table build (Vose's alias
// method with LIFO stacks, common/alias_method.cc:23-63) is one dependency
// chain per batch row: with many rows they run one per lane, but a call with
// FEW LONG rows - the layerwise dataflow passes the whole frontier as ONE row
// (tf_euler/python/dataflow/layerwise_dataflow.py:44-47) - leaves a single
// lane walking n elements through HBM latency (measured: 0.6 - 0.85 us per
// element whatever the batch; 60 ms for one row of 100 000).  Such calls build
// their tables with the host's cores (the same AliasBuildRow source, compiled
// for the host: ~8 ns per element + ~0.35 ms for two copies and a sync; 0.8 ms
// for that row) and keep the draws on the device.
// 1 = choose by that cost model [default], 0 = always the device, 2 = always the host.
thread_local int g_root_host_batch = 1;

static bool RootTablesOnHost(int64_t batch, int32_t n) {
  if (g_root_host_batch != 1) return g_root_host_batch == 2;
  const double dev_us = 0.85 * n + 20.0;
  const double host_us = 350.0 + 0.008 * (double)batch * n;
  return host_us < dev_us;
}
// key 16: SparseGetAdj mask by the direct scan (1) instead of the LDS hash (0).
thread_local int g_adj_scan = 0;
// key 17: SparseGetAdj: sources with more listed edges than this are split over
// workgroups by AdjLongRowsKernel (tests lower it to reach that path on small graphs).
thread_local int g_adj_long_row = 16384;
// key 18: long-row weight sums: 0 = 64 lanes load, lane-shifting DPP chain [default];
// 1 = scalar loads + a wave-uniform add chain (measured 2x slower: 11.9 vs 5.6 ms on
// the 4096 heaviest rows - the s_loads are not overlapped with the chain).
thread_local int g_sum_scalar = 0;

int ExclusiveScanI64(hipStream_t stream, const int64_t* in, int64_t* out,
                     int64_t n);   // mp_kernels.hip

namespace {

struct TypeList {
  int32_t k;
  int32_t et[kMaxListedTypes];
};

// ---------------------------------------------------------------- sum weight
// The sum must add the edge weights in storage order (f32 addition does not
// associate), so a row is one dependency chain.  Short rows: one lane per node
// (EdgeSumWeight).  Rows with more than kLongRow listed edges - a power-law
// graph's hubs, which made one lane of the 1M-node call run 100 000+ dependent
// iterations - are then added by the whole wave: 64 lanes fetch 64 consecutive
// weights (coalesced, four chunks in flight) and the wave adds them in order
// with a lane-shifting chain (ChunkChain): a few clocks per edge instead of a
// memory round trip.
constexpr int kLongRow = 128;

// carry + d[0] + d[1] + ... + d[cnt-1] in exactly that order, d[k] held by lane
// k.  Lane k's running sum is lane k-1's plus d[k]; instead of broadcasting one
// lane per step (v_readlane -> SGPR -> v_add: the SGPR hazard made it ~29
// clocks per edge) every lane repeats  s = shift_right_by_one_lane(s) + d  as
// ONE v_add_f32 with a DPP operand: after step t lane t is final, and
// re-evaluating a final lane reproduces the same bits, so nothing needs masking.
// The shift is row_shr:1 (inside the 16-lane DPP rows, a full-rate operand; the
// whole-wave wave_shr:1 measured ~24 clocks per step): the four rows take turns,
// 16 steps each, and between turns the next row's first lane takes the previous
// row's total through d (v_readlane).  A row's first lane shifts in 0, and its d
// carries the incoming sum: carry + d[0] is the reference's own addition, and
// since the sums start at +0 and can never be -0 the later `0 + x` is exact.
__device__ __forceinline__ float ChunkChain(float carry, float d, int lane, int cnt) {
  float dp = lane == 0 ? __fadd_rn(carry, d) : d;
  float s = dp;
#pragma unroll
  for (int row = 0; row < 4; ++row) {
    if (row > 0) {
      const float prev = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 16 * row - 1));
      if (lane == 16 * row) dp = __fadd_rn(prev, d);
      s = dp;           // first lane of the row: its final value; the others are redone below
    }
#pragma unroll
    for (int t = 1; t < 16; ++t)
      s = __fadd_rn(__int_as_float(__builtin_amdgcn_update_dpp(
                        0, __float_as_int(s), 0x111 /* row_shr:1 */, 0xf, 0xf, true)),
                    dp);
  }
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), cnt - 1));
}

// One row, added by a whole wave.  The weights of the NEXT group of
// kSumGroup x 64 edges are fetched while the chain of the current group runs
// (a group's chain takes about as long as one HBM round trip; without the
// overlap the wave sat idle for a round trip per group).
constexpr int kSumGroup = 8;

__device__ __forceinline__ void LoadWeightGroup(const float* nw, int32_t p0, int32_t e,
                                                int lane, int (&di)[kSumGroup]) {
#pragma unroll
  for (int u = 0; u < kSumGroup; ++u) {
    const int32_t p = p0 + 64 * u + lane;
    float d = 0.f;
    if (p < e) d = __fsub_rn(nw[p], p == 0 ? 0.f : nw[p - 1]);
    di[u] = __float_as_int(d);
  }
}

__device__ __forceinline__ float ChainGroup(float sum, const int (&di)[kSumGroup], int32_t p0,
                                            int32_t e, int lane) {
#pragma unroll
  for (int u = 0; u < kSumGroup; ++u) {
    const int32_t cnt = e - (p0 + 64 * u);
    if (cnt > 0) {
      // the integer form of the same sums first (wave_sums.h); the chain where it does not apply
      float total;
      if (BinadeChunkTotal(sum, __int_as_float(di[u]), lane, &total)) sum = total;
      else sum = ChunkChain(sum, __int_as_float(di[u]), lane, cnt < 64 ? cnt : 64);
    }
  }
  return sum;
}

// Variant of the long-row sum with no cross-lane traffic at all (tuning key 18 = 1;
// measured slower than the DPP chain, kept for A/B):
// the row index is wave-uniform, so the weights can come through SCALAR loads
// (16 floats per s_load) and every lane runs the same v_sub / v_add chain over
// SGPR operands.
__device__ __forceinline__ float UniformRowSum(const GraphView& g, const TypeList& tl,
                                               int64_t row) {
  const RowMeta m = LoadRowMeta(g, row);
  const float* nw = g.prefix_w + m.row_ptr;
  float sum = 0.f;
  for (int32_t x = 0; x < tl.k; ++x) {
    const int32_t t = tl.et[x];
    if (t < 0 || t >= g.T) continue;
    int32_t p = t == 0 ? 0 : m.type_end[t - 1];
    const int32_t e = m.type_end[t];
    if (p >= e) continue;
    float prev = p == 0 ? 0.f : nw[p - 1];
    for (; p + 16 <= e; p += 16) {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = nw[p + i];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        sum = __fadd_rn(sum, __fsub_rn(v[i], prev));
        prev = v[i];
      }
    }
    for (; p < e; ++p) {
      const float c = nw[p];
      sum = __fadd_rn(sum, __fsub_rn(c, prev));
      prev = c;
    }
  }
  return sum;
}

__device__ __forceinline__ float WaveRowSum(const GraphView& g, const TypeList& tl,
                                            int64_t row, int lane) {
  const RowMeta m = LoadRowMeta(g, row);
  const float* nw = g.prefix_w + m.row_ptr;
  float sum = 0.f;
  for (int32_t x = 0; x < tl.k; ++x) {
    const int32_t t = tl.et[x];
    if (t < 0 || t >= g.T) continue;
    const int32_t b = t == 0 ? 0 : m.type_end[t - 1];
    const int32_t e = m.type_end[t];
    if (b >= e) continue;
    // two register groups alternate (no copies between them, so the loads of one
    // stay in flight while the chain of the other runs)
    int ga[kSumGroup], gb[kSumGroup];
    LoadWeightGroup(nw, b, e, lane, ga);
    for (int32_t p0 = b; p0 < e; p0 += 2 * 64 * kSumGroup) {
      LoadWeightGroup(nw, p0 + 64 * kSumGroup, e, lane, gb);     // all zero past the end
      sum = ChainGroup(sum, ga, p0, e, lane);
      LoadWeightGroup(nw, p0 + 2 * 64 * kSumGroup, e, lane, ga);
      sum = ChainGroup(sum, gb, p0 + 64 * kSumGroup, e, lane);
    }
  }
  return sum;
}

__global__ __launch_bounds__(256) void EdgeSumWeightKernel(
    const GraphView g, const TypeList tl, const uint64_t* __restrict__ ids,
    int64_t n, float* __restrict__ out, int64_t* __restrict__ long_rows,
    int32_t* __restrict__ long_pos, unsigned long long* __restrict__ n_long) {
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // whole waves iterate together: the bound is rounded up to a multiple of 64
  const int64_t n_up = (n + 63) & ~(int64_t)63;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_up; i += stride) {
    int64_t row = -1;
    int64_t listed = 0;
    if (i < n) {
      row = FindRow(g, ids[i]);
      if (row >= 0) {
        const RowMeta m = LoadRowMeta(g, row);
        for (int32_t x = 0; x < tl.k; ++x) {
          const int32_t t = tl.et[x];
          if (t >= 0 && t < g.T) listed += m.type_end[t] - (t == 0 ? 0 : m.type_end[t - 1]);
        }
      }
    }
    const bool is_long = listed > kLongRow;
    if (i < n && !is_long) out[i] = EdgeSumWeight(g, ids[i], tl.et, tl.k);
    // long rows go to a queue (one atomic per wave) for EdgeSumLongRowsKernel
    const uint64_t pending = __ballot(is_long);
    if (pending) {
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(n_long, (unsigned long long)__popcll(pending));
      base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
      if (is_long) {
        const unsigned long long q = base + __popcll(pending & ((1ull << lane) - 1ull));
        long_rows[q] = row;
        long_pos[q] = (int32_t)i;
      }
    }
  }
}

// One wave per queued row, whatever wave met it: the hubs of a batch spread
// over the chip instead of queueing up inside the waves that found them.
__global__ __launch_bounds__(256) void EdgeSumLongRowsKernel(
    const GraphView g, const TypeList tl, const int64_t* __restrict__ long_rows,
    const int32_t* __restrict__ long_pos, const unsigned long long* __restrict__ n_long,
    float* __restrict__ out, int scalar_loads) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t total = (int64_t)*n_long;
  for (int64_t q = wave; q < total; q += n_waves) {
    // wave-uniform row index (scalar loop bounds in WaveRowSum)
    const int64_t rv = long_rows[q];
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rv);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)rv >> 32));
    const int64_t row = (int64_t)(((uint64_t)hi << 32) | lo);
    const float sum = scalar_loads ? UniformRowSum(g, tl, row) : WaveRowSum(g, tl, row, lane);
    if (lane == 0) out[long_pos[q]] = sum;
  }
}

// ---------------------------------------------------------------- root draw
struct RootScratch {
  float* wn;        // [n][batch] normalised weights (updated by the build)
  float* prob;      // [n][batch]
  int32_t* alias;   // [n][batch]
  int32_t* stack;   // [n][batch]
  float* sum;       // [batch]
};

__global__ __launch_bounds__(256) void SampleRootBuildKernel(
    const float* __restrict__ weights, int64_t batch, int32_t n, RootScratch s) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < batch; b += stride)
    s.sum[b] = AliasBuildRow(weights + b * n, n, batch, s.wn + b, s.prob + b,
                             s.alias + b, s.stack + b);
}

__global__ __launch_bounds__(256) void SampleRootDrawKernel(
    const uint64_t* __restrict__ roots, int64_t batch, int32_t n, int32_t m,
    uint64_t seed, uint32_t call_id, int64_t default_node, RootScratch s,
    uint64_t* __restrict__ out) {
  const int64_t total = batch * m;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t b = i / m;
    const int32_t j = (int32_t)(i - b * m);
    if (s.sum[b] == 0.f) {                     // sample_root_op.cc:74-78
      out[i] = (uint64_t)default_node;
    } else {
      const int32_t slot =
          SampleRootSlot(seed, call_id, b, j, n, batch, s.prob + b, s.alias + b);
      out[i] = roots[b * n + slot];
    }
  }
}

// ---------------------------------------------------------------- layer draw
__global__ __launch_bounds__(256) void SampleLayerKernel(
    const GraphView g, const TypeList tl, const uint64_t* __restrict__ roots,
    const int64_t* __restrict__ pos, int64_t n, uint64_t seed, uint32_t call_id,
    int64_t default_node, uint64_t* __restrict__ out_id, float* __restrict__ out_w,
    int32_t* __restrict__ out_t) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint64_t id; float w; int32_t t;
    // the RNG stream is the position in the op's root list: i, or the position the
    // requester of a multi-GPU hop gave this root
    SampleLayerAt(g, seed, call_id, pos ? pos[i] : i, roots[i], tl.et, tl.k, default_node,
                  &id, &w, &t);
    out_id[i] = id;
    if (out_w) out_w[i] = w;
    if (out_t) out_t[i] = t;
  }
}

// ---------------------------------------------------------------- adjacency
// SparseGetAdj asks, for every source node r of batch row b, which of the m
// candidates of that row it has a listed-type edge to.  The answers go into a
// bit mask (one bit per (source, candidate), words of 64) from which the
// count / scan / fill passes build either result layout without touching the
// graph again.
//
// AdjHashMaskKernel (default): O(deg + m) per source instead of O(deg * m).
// A workgroup takes a chunk of sources of one batch row and puts the row's
// candidates (<= kAdjChunk at a time) into an LDS hash table - slots hold the
// index of the first candidate with a key, duplicates resolve to it - then
// each wave streams the adjacency row of one source (lanes over edges,
// coalesced), probes the table with every neighbour id and sets the hit bit of
// the matching candidate in its LDS bitmap; finally lane j looks its own
// candidate up and the ballot of the hit bits is one mask word.  Degree skew
// no longer multiplies with m: the 4096 x 4096 WholeDataFlow query on the
// metric graph went from 77 ms (one wave per source comparing every candidate
// with the whole row, AdjScanMaskKernel, kept selectable: tuning key 16) to
// the time of streaming the rows once.
constexpr int kAdjChunk = 2048;          // candidates per table build (11 index bits)
constexpr uint32_t kAdjEmpty = 0xFFFFFFFFu;
// A table slot = (21-bit tag of the key's hash << 11) | candidate index: a probe
// compares tags with ONE LDS read and touches the 8-byte key only on a tag match
// (v1 read slot -> key for every probe, two dependent LDS round trips).  The
// table has 4 slots per candidate: the lanes of a wave probe in lockstep, so a
// step costs the LONGEST probe chain among 64 lanes - at half load that was ~10
// slots per step and a hub row went at 64 edges per 2 us.
constexpr int kAdjIdxBits = 11;
__device__ __forceinline__ uint32_t AdjTag(uint64_t h) {
  const uint32_t tag = (uint32_t)(h >> 43);            // 21 bits
  return tag == 0x1FFFFFu ? 0u : tag;                   // never the empty pattern
}

struct AdjArgs {
  GraphView g;
  TypeList tl;
  const uint64_t* roots;    // [batch * n]
  const uint64_t* l_nb;     // [batch * m]
  uint64_t* mask;           // [batch * n * words]
  int64_t batch;
  int32_t n, m;
  int32_t words;            // ceil(m / 64)
  int32_t roots_per_wg;     // sources of one workgroup (its 4 waves take every 4th)
  int32_t wgs_per_row;      // workgroups per batch row
  int32_t cap;              // hash slots (power of two >= 4 * min(m, kAdjChunk))
  int32_t long_row;         // rows with more listed edges go to AdjLongRowsKernel
  int64_t* long_src;        // [batch * n] queue of such sources (index b * n + slot)
  unsigned long long* n_long;
};

constexpr int kAdjGroup = 8;

__device__ __forceinline__ void LoadNbrGroup(const uint64_t* nbr, int32_t p0, int32_t ee,
                                             int lane, uint64_t (&d)[kAdjGroup]) {
#pragma unroll
  for (int u = 0; u < kAdjGroup; ++u) {
    const int32_t p = p0 + 64 * u + lane;
    d[u] = p < ee ? nbr[p] : 0;
  }
}

// index of the candidate with this key, or -1
__device__ __forceinline__ int AdjFind(uint64_t key, const uint32_t* table,
                                       const uint64_t* cand, uint32_t cmask, int cap) {
  const uint64_t h = Mix64(key);
  const uint32_t tag = AdjTag(h);
  uint32_t slot = (uint32_t)h & cmask;
  for (int probes = 0; probes < cap; ++probes) {
    const uint32_t e = table[slot];
    if (e == kAdjEmpty) return -1;
    if ((e >> kAdjIdxBits) == tag && cand[e & (kAdjChunk - 1)] == key)
      return (int)(e & (kAdjChunk - 1));
    slot = (slot + 1u) & cmask;
  }
  return -1;
}

__device__ __forceinline__ void AdjProbeGroup(const uint64_t (&d)[kAdjGroup], int32_t p0,
                                              int32_t ee, int lane, const uint32_t* table,
                                              const uint64_t* cand, uint32_t cmask, int cap,
                                              uint32_t* my_bits) {
#pragma unroll
  for (int u = 0; u < kAdjGroup; ++u) {
    if (p0 + 64 * u + lane >= ee) continue;
    const int jv = AdjFind(d[u], table, cand, cmask, cap);
    if (jv >= 0) {
      // a hub among the candidates is hit by most lanes of most steps: only the
      // first hit pays for the (same-address, serialised) LDS atomic
      const uint32_t bit = 1u << (jv & 31);
      if (!(my_bits[jv >> 5] & bit)) atomicOr(&my_bits[jv >> 5], bit);
    }
  }
}

__device__ __forceinline__ void AdjWaveSync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// All 256 threads: the candidates [c0, c0 + mc) of batch row b into the LDS table.
__device__ __forceinline__ void AdjBuildTable(const AdjArgs& a, int64_t b, int32_t c0, int mc,
                                              int tid, uint64_t* cand, uint32_t* table,
                                              uint32_t cmask) {
  __syncthreads();                       // everybody is done with the previous table
  for (int i = tid; i < a.cap; i += 256) table[i] = kAdjEmpty;
  for (int j = tid; j < mc; j += 256) cand[j] = a.l_nb[b * a.m + c0 + j];
  __syncthreads();
  for (int j = tid; j < mc; j += 256) {
    const uint64_t key = cand[j];
    const uint64_t h = Mix64(key);
    const uint32_t tag = AdjTag(h);
    const uint32_t entry = (tag << kAdjIdxBits) | (uint32_t)j;
    uint32_t slot = (uint32_t)h & cmask;
    for (int probes = 0; probes < a.cap; ++probes) {
      const uint32_t prev = atomicCAS(&table[slot], kAdjEmpty, entry);
      if (prev == kAdjEmpty) break;                                    // inserted
      if ((prev >> kAdjIdxBits) == tag && cand[prev & (kAdjChunk - 1)] == key) break;  // duplicate
      slot = (slot + 1u) & cmask;
    }
  }
  __syncthreads();
}

// Listed edges of a row (what a source streams): sum over the valid listed types.
__device__ __forceinline__ int64_t AdjListedDegree(const AdjArgs& a, const RowMeta& rm) {
  int64_t d = 0;
  for (int32_t x = 0; x < a.tl.k; ++x) {
    const int32_t t = a.tl.et[x];
    if (t >= 0 && t < a.g.T) d += rm.type_end[t] - (t == 0 ? 0 : rm.type_end[t - 1]);
  }
  return d;
}

__global__ __launch_bounds__(256) void AdjHashMaskKernel(const AdjArgs a) {
  extern __shared__ uint64_t adj_lds[];
  const int mc_max = a.m < kAdjChunk ? a.m : kAdjChunk;
  uint64_t* cand = adj_lds;                                        // [mc_max]
  uint32_t* table = reinterpret_cast<uint32_t*>(cand + mc_max);    // [cap]
  uint32_t* bits = table + a.cap;                                  // [4][bw]
  const int bw = (mc_max + 31) >> 5;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t* my_bits = bits + wave * bw;
  const uint32_t cmask = (uint32_t)a.cap - 1u;
  const int64_t total_wgs = a.batch * a.wgs_per_row;
  for (int64_t wg = blockIdx.x; wg < total_wgs; wg += gridDim.x) {
    const int64_t b = wg / a.wgs_per_row;
    const int32_t r_begin = (int32_t)(wg - b * a.wgs_per_row) * a.roots_per_wg;
    const int32_t r_end = min(a.n, r_begin + a.roots_per_wg);
    for (int32_t c0 = 0; c0 < a.m; c0 += kAdjChunk) {
      const int mc = min(kAdjChunk, a.m - c0);
      AdjBuildTable(a, b, c0, mc, tid, cand, table, cmask);
      // The waves walk their sources independently (a hub row delays only its own
      // wave): the bitmap is private to the wave, whose LDS operations execute in
      // program order, so a wave-scope fence (no reordering by the compiler) is
      // all the clear / mark / read-back phases need between them.
      for (int32_t slot_r = r_begin + wave; slot_r < r_end; slot_r += 4) {
        for (int i = lane; i < bw; i += 64) my_bits[i] = 0u;
        AdjWaveSync();
        {
          const int64_t row = FindRow(a.g, a.roots[b * a.n + slot_r]);
          bool here = row >= 0;
          if (here) {
            const RowMeta rm0 = LoadRowMeta(a.g, row);
            if (AdjListedDegree(a, rm0) > a.long_row) {
              // a hub: queued once (first candidate pass) for AdjLongRowsKernel, which
              // splits the row over many workgroups; its mask words stay zero here
              here = false;
              if (c0 == 0 && lane == 0) a.long_src[atomicAdd(a.n_long, 1ull)] = b * a.n + slot_r;
            }
          }
          if (here) {
            const RowMeta rm = LoadRowMeta(a.g, row);
            const uint64_t* nbr = a.g.nbr + rm.row_ptr;
            for (int32_t x = 0; x < a.tl.k; ++x) {
              const int32_t t = a.tl.et[x];
              if (t < 0 || t >= a.g.T) continue;
              const int32_t eb = t == 0 ? 0 : rm.type_end[t - 1];
              const int32_t ee = rm.type_end[t];
              // two register groups of 8 x 64 neighbour ids alternate: one loads
              // while the other probes the table
              if (eb >= ee) continue;
              uint64_t ga[kAdjGroup], gb[kAdjGroup];
              LoadNbrGroup(nbr, eb, ee, lane, ga);
              for (int32_t p0 = eb; p0 < ee; p0 += 2 * 64 * kAdjGroup) {
                LoadNbrGroup(nbr, p0 + 64 * kAdjGroup, ee, lane, gb);
                AdjProbeGroup(ga, p0, ee, lane, table, cand, cmask, a.cap, my_bits);
                LoadNbrGroup(nbr, p0 + 2 * 64 * kAdjGroup, ee, lane, ga);
                AdjProbeGroup(gb, p0 + 64 * kAdjGroup, ee, lane, table, cand, cmask, a.cap,
                              my_bits);
              }
            }
          }
        }
        AdjWaveSync();
        {
          uint64_t* out = a.mask + (b * a.n + slot_r) * (int64_t)a.words + (c0 >> 6);
          for (int j0 = 0; j0 < mc; j0 += 64) {
            const int j = j0 + lane;
            bool hit = false;
            if (j < mc) {
              const int jv = AdjFind(cand[j], table, cand, cmask, a.cap);   // >= 0: j was inserted
              if (jv >= 0) hit = (my_bits[jv >> 5] >> (jv & 31)) & 1u;
            }
            const uint64_t word = __ballot(hit);
            if (lane == 0) out[j0 >> 6] = word;
          }
        }
        AdjWaveSync();
      }
    }
  }
}

// Hub rows.  One wave streams 64 edges per ~0.8 us through the table, so the
// 545 K-edge row of the metric graph alone took 7 ms per candidate pass.  The
// sources the main kernel queued are cut into segments of kAdjSegment listed
// edges; a workgroup takes one (source, segment) unit at a time, its four waves
// share the segment and ONE LDS bitmap, and the hits leave through 64-bit
// atomicOr on the (zeroed) mask words - few, because adjacency is sparse.
constexpr int kAdjSegment = 8192;

__global__ __launch_bounds__(256) void AdjLongRowsKernel(const AdjArgs a) {
  extern __shared__ uint64_t adj_lds[];
  const int mc_max = a.m < kAdjChunk ? a.m : kAdjChunk;
  uint64_t* cand = adj_lds;
  uint32_t* table = reinterpret_cast<uint32_t*>(cand + mc_max);
  uint32_t* bits = table + a.cap;                                  // [bw] shared by the 4 waves
  const int bw = (mc_max + 31) >> 5;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t cmask = (uint32_t)a.cap - 1u;
  const int64_t n_long = (int64_t)*a.n_long;
  // units are numbered source by source; every thread walks the queue the same way
  int64_t unit0 = 0;            // first unit of queue entry q
  for (int64_t q = 0; q < n_long; ++q) {
    const int64_t r = a.long_src[q];
    const int64_t b = r / a.n;
    const int64_t row = FindRow(a.g, a.roots[r]);
    const RowMeta rm = LoadRowMeta(a.g, row);
    const int64_t D = AdjListedDegree(a, rm);
    const int64_t nseg = (D + kAdjSegment - 1) / kAdjSegment;
    // this workgroup's units inside [unit0, unit0 + nseg)
    int64_t u = unit0 + ((int64_t)blockIdx.x - unit0 % gridDim.x + gridDim.x) % gridDim.x;
    for (; u < unit0 + nseg; u += gridDim.x) {
      const int64_t s_begin = (u - unit0) * kAdjSegment;
      const int64_t s_end = min(D, s_begin + (int64_t)kAdjSegment);
      for (int32_t c0 = 0; c0 < a.m; c0 += kAdjChunk) {
        const int mc = min(kAdjChunk, a.m - c0);
        AdjBuildTable(a, b, c0, mc, tid, cand, table, cmask);
        for (int i = tid; i < bw; i += 256) bits[i] = 0u;
        __syncthreads();
        // the segment's positions in the concatenation of the listed ranges
        const uint64_t* nbr = a.g.nbr + rm.row_ptr;
        int64_t base = 0;
        for (int32_t x = 0; x < a.tl.k; ++x) {
          const int32_t t = a.tl.et[x];
          if (t < 0 || t >= a.g.T) continue;
          const int32_t eb = t == 0 ? 0 : rm.type_end[t - 1];
          const int32_t ee = rm.type_end[t];
          const int64_t lo = max(s_begin, base), hi = min(s_end, base + (ee - eb));
          base += ee - eb;
          if (lo >= hi) continue;
          const int32_t pb = eb + (int32_t)(lo - (base - (ee - eb)));
          const int32_t pe = eb + (int32_t)(hi - (base - (ee - eb)));
          // wave w takes the groups w, w + 4, ... of [pb, pe)
          for (int32_t p0 = pb + wave * 64 * kAdjGroup; p0 < pe; p0 += 4 * 64 * kAdjGroup) {
            uint64_t gr[kAdjGroup];
            LoadNbrGroup(nbr, p0, pe, lane, gr);
            AdjProbeGroup(gr, p0, pe, lane, table, cand, cmask, a.cap, bits);
          }
        }
        __syncthreads();
        uint64_t* out = a.mask + r * (int64_t)a.words + (c0 >> 6);
        for (int j = tid; j < mc; j += 256) {
          const int jv = AdjFind(cand[j], table, cand, cmask, a.cap);
          if (jv >= 0 && ((bits[jv >> 5] >> (jv & 31)) & 1u))
            atomicOr(reinterpret_cast<unsigned long long*>(out + (j >> 6)), 1ull << (j & 63));
        }
      }
    }
    unit0 += nseg;
  }
}

// The direct form (tuning key 16 = 1): one wave per source, lanes over the
// candidates, every lane compares its candidate with the whole row
// (EdgeExistAny).  Same mask.
__global__ __launch_bounds__(256) void AdjScanMaskKernel(const AdjArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t total = a.batch * a.n;
  for (int64_t r = wave; r < total; r += n_waves) {
    const int64_t b = r / a.n;
    const int64_t row = FindRow(a.g, a.roots[r]);
    for (int32_t base = 0; base < a.m; base += 64) {
      const int32_t j = base + lane;
      const bool hit = j < a.m &&
                       EdgeExistAny(a.g, row, a.l_nb[b * a.m + j], a.tl.et, a.tl.k);
      const uint64_t word = __ballot(hit);
      if (lane == 0) a.mask[r * (int64_t)a.words + (base >> 6)] = word;
    }
  }
}

// The TF kernels add an explicit zero at (b, n-1, m-1) when that pair is no
// edge (tf_euler/kernels/sparse_get_adj_op.cc:112-118): the word an entry list
// is built from is the mask word plus that corner bit.
__device__ __forceinline__ uint64_t AdjEmitWord(const uint64_t* mask, int64_t r,
                                                int32_t w, int32_t n, int32_t m,
                                                int32_t words, int32_t tf) {
  uint64_t word = mask[r * (int64_t)words + w];
  if (tf && w == words - 1 && (int32_t)(r % n) == n - 1) word |= 1ull << ((m - 1) & 63);
  return word;
}

__global__ __launch_bounds__(256) void AdjCountKernel(const uint64_t* __restrict__ mask,
                                                      int64_t R, int32_t n, int32_t m,
                                                      int32_t words, int32_t tf,
                                                      int64_t* __restrict__ counts) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += stride) {
    int64_t c = 0;
    for (int32_t w = 0; w < words; ++w) c += __popcll(AdjEmitWord(mask, r, w, n, m, words, tf));
    counts[r] = c;
  }
}

// One lane per (source, mask word): entries of a source keep candidate order
// (the push_back order of core/kernels/sparse_get_adj_op.cc:60-72).
__global__ __launch_bounds__(256) void AdjFillKernel(
    const uint64_t* __restrict__ mask, const uint64_t* __restrict__ l_nb, int64_t R,
    int32_t n, int32_t m, int32_t words, int32_t tf, const int64_t* __restrict__ off,
    uint64_t* __restrict__ out_id, int64_t* __restrict__ indices,
    int64_t* __restrict__ values) {
  const int64_t total = R * words;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += stride) {
    const int64_t r = x / words;
    const int32_t w = (int32_t)(x - r * words);
    uint64_t word = AdjEmitWord(mask, r, w, n, m, words, tf);
    if (word == 0) continue;
    int64_t p = off[r];
    for (int32_t v = 0; v < w; ++v) p += __popcll(AdjEmitWord(mask, r, v, n, m, words, tf));
    const uint64_t hits = mask[x];
    const int64_t b = r / n;
    while (word) {
      const int bit = __ffsll((unsigned long long)word) - 1;
      word &= word - 1;
      const int32_t j = w * 64 + bit;
      if (tf) {
        indices[3 * p] = b;
        indices[3 * p + 1] = r - b * n;
        indices[3 * p + 2] = j;
        values[p] = (hits >> bit) & 1ull;
      } else {
        out_id[p] = l_nb[b * m + j];
      }
      ++p;
    }
  }
}

__global__ void AdjOffsetsToIdxKernel(const int64_t* __restrict__ off, int64_t n,
                                      int32_t* __restrict__ idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    idx[2 * i] = (int32_t)off[i];
    idx[2 * i + 1] = (int32_t)off[i + 1];
  }
}

// ---------------------------------------------------------------- local layer
// API_LOCAL_SAMPLE_L draws (core/kernels/local_sample_layer_op.cc:128-145): one
// lane per (batch row, sample): CompactWeightedCollection::Sample =
// RandomSelect over the row's running sums, or the op's memset fill.
struct LocalLayerArgs {
  const int64_t* seg;       // [batch + 1] offsets of the rows' distinct entries
  const uint64_t* u_id;
  const float* u_w;         // accumulated (and sqrt'ed) weight of every entry
  const int32_t* u_t;
  const float* sum_w;       // running f32 sums inside every row
  int64_t batch;
  int32_t m;
  uint64_t seed;
  uint32_t call_id;
  uint64_t fill_id;         // default_node's low byte repeated (the op memsets)
  uint64_t* out_id;
  float* out_w;
  int32_t* out_t;
};

__global__ __launch_bounds__(256) void LocalSampleLayerKernel(const LocalLayerArgs a) {
  const int64_t total = a.batch * a.m;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += stride) {
    const int64_t b = x / a.m;
    const int32_t j = (int32_t)(x - b * a.m);
    const int64_t s0 = a.seg[b];
    const int64_t mid = LocalLayerPick(a.seed, a.call_id, b, j, a.sum_w + s0, a.seg[b + 1] - s0);
    if (mid < 0) {                                            // :129-134
      a.out_id[x] = a.fill_id; a.out_w[x] = 0.f; a.out_t[x] = 0;
      continue;
    }
    a.out_id[x] = a.u_id[s0 + mid];
    a.out_w[x] = a.u_w[s0 + mid];
    a.out_t[x] = a.u_t[s0 + mid];
  }
}

// ---------------------------------------------------------------- node types
__global__ __launch_bounds__(256) void NodeTypeKernel(
    const GraphView g, const int32_t* __restrict__ node_type,
    const uint64_t* __restrict__ ids, int64_t n, int32_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = NodeTypeOf(g, node_type, ids[i]);
}

// one lane per (row i, sample j) of API_SAMPLE_N_WITH_TYPES
__global__ __launch_bounds__(256) void SampleNWithTypesKernel(
    const NodeSamplerView s, const int32_t* __restrict__ types, int64_t n,
    int32_t count, uint64_t seed, uint32_t call_id, uint64_t* __restrict__ out,
    int32_t* __restrict__ bad) {
  const int64_t total = n * count;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += stride) {
    const int64_t i = x / count;
    const int32_t j = (int32_t)(x - i * count);
    bool ok;
    out[x] = SampleNodeOfType(s, seed, call_id, (uint64_t)i, types[i], j, &ok);
    if (!ok && j == 0) *bad = 1;
  }
}

int FillTypes(const int32_t* edge_types_host, int32_t k, TypeList* tl, const char* who) {
  if (k < 0 || k > kMaxListedTypes || (k > 0 && !edge_types_host))
    return Fail(EULER_GPU_EINVAL, std::string(who) + ": bad edge type list (<= 32)");
  tl->k = k;
  for (int32_t i = 0; i < k; ++i) tl->et[i] = edge_types_host[i];
  for (int32_t i = k; i < kMaxListedTypes; ++i) tl->et[i] = 0;
  return EULER_GPU_OK;
}

// Workspace of a SparseGetAdj query: offsets [R + 1] int64, then the mask.
struct AdjWorkspace {
  int64_t* off;
  uint64_t* mask;
  int32_t words;
};

AdjWorkspace SplitAdjWorkspace(void* ws, int64_t R, int32_t m) {
  AdjWorkspace w;
  w.off = static_cast<int64_t*>(ws);
  w.mask = reinterpret_cast<uint64_t*>(w.off + R + 1);
  w.words = (m + 63) / 64;
  return w;
}

int AdjMaskOffsets(hipStream_t st, const uint64_t* mask, int64_t R, int32_t n, int32_t m,
                   int32_t words, int32_t tf, int64_t* off);

// the hit mask of every source: [R][words] uint64, bit c of source r = candidate c of its
// batch row is in the source's row
int BuildAdjMaskOnly(const euler_gpu_graph* g, hipStream_t st, AdjArgs a, uint64_t* mask,
                     int32_t words) {
  const int64_t R = a.batch * a.n;
  const int block = 256;
  a.g = g->view;
  a.mask = mask;
  a.words = words;
  if (a.m > 0) {
    if (g_adj_scan) {
      hipLaunchKernelGGL(AdjScanMaskKernel, dim3(GridFor(R * 64, block)), dim3(block), 0,
                         st, a);
    } else {
      const int32_t mc = a.m < kAdjChunk ? a.m : kAdjChunk;
      int32_t cap = 64;
      while (cap < 4 * mc) cap <<= 1;
      a.cap = cap;
      // ~4096 workgroups over the whole query, >= 4 sources (one per wave) each
      const int64_t want = std::max<int64_t>(1, 4096 / std::max<int64_t>(a.batch, 1));
      int32_t rpw = (int32_t)((a.n + want - 1) / want);
      rpw = std::max(4, (rpw + 3) & ~3);
      a.roots_per_wg = rpw;
      a.wgs_per_row = (a.n + rpw - 1) / rpw;
      const int64_t wgs = a.batch * (int64_t)a.wgs_per_row;
      const size_t lds = (size_t)mc * 8 + (size_t)cap * 4 + (size_t)4 * ((mc + 31) / 32) * 4;
      // queue of the hub sources: [R] indices + a counter
      uint8_t* q = nullptr;
      EG_HIP(hipMallocAsync((void**)&q, (size_t)R * 8 + 8, st));
      a.long_src = reinterpret_cast<int64_t*>(q);
      a.n_long = reinterpret_cast<unsigned long long*>(q + (size_t)R * 8);
      a.long_row = g_adj_long_row;
      hipError_t e0 = hipMemsetAsync(a.n_long, 0, 8, st);
      hipLaunchKernelGGL(AdjHashMaskKernel, dim3((unsigned)std::min<int64_t>(wgs, 1 << 16)),
                         dim3(block), lds, st, a);
      hipLaunchKernelGGL(AdjLongRowsKernel, dim3(2048), dim3(block), lds, st, a);
      hipError_t e1 = hipGetLastError();
      hipError_t e2 = hipFreeAsync(q, st);
      EG_HIP(e0); EG_HIP(e1); EG_HIP(e2);
    }
    EG_HIP(hipGetLastError());
  }
  return EULER_GPU_OK;
}

// mask -> counts -> offsets [R + 1] (off[R] = total)
int AdjMaskOffsets(hipStream_t st, const uint64_t* mask, int64_t R, int32_t n, int32_t m,
                   int32_t words, int32_t tf, int64_t* off) {
  const int block = 256;
  int64_t* counts = nullptr;
  EG_HIP(hipMallocAsync((void**)&counts, (size_t)(R + 1) * sizeof(int64_t), st));
  EG_HIP(hipMemsetAsync(counts + R, 0, sizeof(int64_t), st));
  hipLaunchKernelGGL(AdjCountKernel, dim3(GridFor(R, block)), dim3(block), 0, st, mask, R, n, m,
                     words, tf, counts);
  int rc = ExclusiveScanI64(st, counts, off, R + 1);
  hipError_t f = hipFreeAsync(counts, st);
  if (rc != EULER_GPU_OK) return rc;
  EG_HIP(f);
  return EULER_GPU_OK;
}

// mask -> counts -> offsets, all on the stream
int BuildAdjMask(const euler_gpu_graph* g, hipStream_t st, AdjArgs a, int32_t tf,
                 const AdjWorkspace& w) {
  const int rc = BuildAdjMaskOnly(g, st, a, w.mask, w.words);
  if (rc != EULER_GPU_OK) return rc;
  return AdjMaskOffsets(st, w.mask, a.batch * a.n, a.n, a.m, w.words, tf, w.off);
}

}  // namespace
}  // namespace euler_gpu

using namespace euler_gpu;

extern "C" {

int euler_gpu_get_edge_sum_weight(const euler_gpu_graph* g, void* stream,
                                  const uint64_t* ids_dev, int64_t n,
                                  const int32_t* edge_types_host, int32_t k,
                                  float* out_w_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "get_edge_sum_weight: null graph");
  if (n < 0) return Fail(EULER_GPU_EINVAL, "get_edge_sum_weight: n < 0");
  TypeList tl;
  int rc = FillTypes(edge_types_host, k, &tl, "get_edge_sum_weight");
  if (rc != EULER_GPU_OK) return rc;
  if (n == 0) return EULER_GPU_OK;
  if (!ids_dev || !out_w_dev)
    return Fail(EULER_GPU_EINVAL, "get_edge_sum_weight: null buffer");
  if (n > 0x7fffffffLL) return Fail(EULER_GPU_EINVAL, "get_edge_sum_weight: n exceeds int32");
  const int block = 256;
  hipStream_t st = (hipStream_t)stream;
  // queue of the long rows: [n] rows, [n] positions, one counter
  uint8_t* q = nullptr;
  EG_HIP(hipMallocAsync((void**)&q, (size_t)n * 12 + 16, st));
  int64_t* long_rows = reinterpret_cast<int64_t*>(q);
  int32_t* long_pos = reinterpret_cast<int32_t*>(long_rows + n);
  unsigned long long* n_long =
      reinterpret_cast<unsigned long long*>(q + (((size_t)n * 12 + 7) & ~(size_t)7));
  hipError_t e = hipMemsetAsync(n_long, 0, 8, st);
  hipLaunchKernelGGL(EdgeSumWeightKernel, dim3(GridFor(n, block)), dim3(block), 0, st,
                     g->view, tl, ids_dev, n, out_w_dev, long_rows, long_pos, n_long);
  hipLaunchKernelGGL(EdgeSumLongRowsKernel, dim3(GridFor(n * 64, block)), dim3(block), 0, st,
                     g->view, tl, long_rows, long_pos, n_long, out_w_dev, g_sum_scalar);
  hipError_t l = hipGetLastError();
  hipError_t f = hipFreeAsync(q, st);
  EG_HIP(e); EG_HIP(l); EG_HIP(f);
  return EULER_GPU_OK;
}

int euler_gpu_sample_root(void* stream, uint64_t seed, uint32_t call_id,
                          const uint64_t* roots_dev, const float* weights_dev,
                          int64_t batch, int32_t n, int32_t m, int64_t default_node,
                          uint64_t* out_dev) {
  if (batch < 0 || n <= 0 || m < 0)
    return Fail(EULER_GPU_EINVAL, "sample_root: need batch >= 0, n > 0, m >= 0");
  if (batch == 0 || m == 0) return EULER_GPU_OK;
  if (!roots_dev || !weights_dev || !out_dev)
    return Fail(EULER_GPU_EINVAL, "sample_root: null buffer");
  hipStream_t st = (hipStream_t)stream;
  const int64_t cells = batch * n;
  uint8_t* buf = nullptr;
  EG_HIP(hipMallocAsync((void**)&buf, (size_t)(cells * 16 + batch * 4), st));
  RootScratch s;
  s.wn = reinterpret_cast<float*>(buf);
  s.prob = s.wn + cells;
  s.alias = reinterpret_cast<int32_t*>(s.prob + cells);
  s.stack = s.alias + cells;
  s.sum = reinterpret_cast<float*>(s.stack + cells);
  const int block = 256;
  const bool on_host = RootTablesOnHost(batch, n);
  std::vector<float> h_w, h_wn, h_prob, h_sum;
  std::vector<int32_t> h_alias, h_stack;
  if (on_host) {
    h_w.resize(cells); h_wn.resize(cells); h_prob.assign(cells, 0.f); h_sum.resize(batch);
    h_alias.assign(cells, 0); h_stack.resize(cells);
    hipError_t c = hipMemcpyAsync(h_w.data(), weights_dev, (size_t)cells * 4,
                                  hipMemcpyDeviceToHost, st);
    if (c == hipSuccess) c = hipStreamSynchronize(st);
    if (c != hipSuccess) { (void)hipFreeAsync(buf, st); EG_HIP(c); }
    auto build = [&](int64_t b0, int64_t b1) {
      for (int64_t b = b0; b < b1; ++b)
        h_sum[b] = AliasBuildRow(h_w.data() + b * n, n, batch, h_wn.data() + b,
                                 h_prob.data() + b, h_alias.data() + b,
                                 h_stack.data() + b);
    };
    const int64_t hw = (int64_t)std::thread::hardware_concurrency();
    const int64_t n_thr = std::min<int64_t>(std::min<int64_t>(batch, 8),
                                            std::max<int64_t>(1, std::min<int64_t>(hw, cells >> 14)));
    if (n_thr <= 1) {
      build(0, batch);
    } else {
      std::vector<std::thread> pool;
      for (int64_t t = 0; t < n_thr; ++t)
        pool.emplace_back(build, batch * t / n_thr, batch * (t + 1) / n_thr);
      for (auto& th : pool) th.join();
    }
    c = hipMemcpyAsync(s.prob, h_prob.data(), (size_t)cells * 4, hipMemcpyHostToDevice, st);
    if (c == hipSuccess)
      c = hipMemcpyAsync(s.alias, h_alias.data(), (size_t)cells * 4, hipMemcpyHostToDevice, st);
    if (c == hipSuccess)
      c = hipMemcpyAsync(s.sum, h_sum.data(), (size_t)batch * 4, hipMemcpyHostToDevice, st);
    if (c != hipSuccess) { (void)hipFreeAsync(buf, st); EG_HIP(c); }
  } else {
    hipLaunchKernelGGL(SampleRootBuildKernel, dim3(GridFor(batch, block)), dim3(block), 0,
                       st, weights_dev, batch, n, s);
  }
  hipLaunchKernelGGL(SampleRootDrawKernel, dim3(GridFor(batch * m, block)), dim3(block),
                     0, st, roots_dev, batch, n, m, seed, call_id, default_node, s,
                     out_dev);
  hipError_t e = hipGetLastError();
  // the host vectors feed asynchronous copies: they must outlive them
  hipError_t y = on_host ? hipStreamSynchronize(st) : hipSuccess;
  hipError_t f = hipFreeAsync(buf, st);
  EG_HIP(e);
  EG_HIP(y);
  EG_HIP(f);
  return EULER_GPU_OK;
}

int euler_gpu_sample_layer(const euler_gpu_graph* g, void* stream, uint64_t seed,
                           uint32_t call_id, const uint64_t* roots_dev, int64_t n,
                           const int32_t* edge_types_host, int32_t k,
                           int64_t default_node, uint64_t* out_id_dev,
                           float* out_w_dev, int32_t* out_t_dev) {
  return euler_gpu_sample_layer_at(g, stream, seed, call_id, roots_dev, nullptr, n,
                                   edge_types_host, k, default_node, out_id_dev, out_w_dev,
                                   out_t_dev);
}

int euler_gpu_sample_layer_at(const euler_gpu_graph* g, void* stream, uint64_t seed,
                              uint32_t call_id, const uint64_t* roots_dev,
                              const int64_t* pos_dev, int64_t n,
                              const int32_t* edge_types_host, int32_t k,
                              int64_t default_node, uint64_t* out_id_dev,
                              float* out_w_dev, int32_t* out_t_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "sample_layer: null graph");
  if (n < 0) return Fail(EULER_GPU_EINVAL, "sample_layer: n < 0");
  TypeList tl;
  int rc = FillTypes(edge_types_host, k, &tl, "sample_layer");
  if (rc != EULER_GPU_OK) return rc;
  if (n == 0) return EULER_GPU_OK;
  if (!roots_dev || !out_id_dev)
    return Fail(EULER_GPU_EINVAL, "sample_layer: null buffer");
  const int block = 256;
  hipLaunchKernelGGL(SampleLayerKernel, dim3(GridFor(n, block)), dim3(block), 0,
                     (hipStream_t)stream, g->view, tl, roots_dev, pos_dev, n, seed, call_id,
                     default_node, out_id_dev, out_w_dev, out_t_dev);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

int euler_gpu_sample_neighbor_layerwise(const euler_gpu_graph* g, void* stream,
                                        uint64_t seed, uint32_t call_id,
                                        const uint64_t* nodes_dev, int64_t batch,
                                        int32_t n, const int32_t* edge_types_host,
                                        int32_t k, int32_t count,
                                        int64_t default_node, uint64_t* out_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "sample_neighbor_layerwise: null graph");
  if (batch < 0 || n <= 0 || count < 0)
    return Fail(EULER_GPU_EINVAL,
                "sample_neighbor_layerwise: need batch >= 0, n > 0, count >= 0");
  if (batch == 0 || count == 0) return EULER_GPU_OK;
  if (!nodes_dev || !out_dev)
    return Fail(EULER_GPU_EINVAL, "sample_neighbor_layerwise: null buffer");
  hipStream_t st = (hipStream_t)stream;
  uint8_t* buf = nullptr;
  const int64_t cells = batch * n, draws = batch * (int64_t)count;
  EG_HIP(hipMallocAsync((void**)&buf, (size_t)(draws * 8 + cells * 4), st));
  uint64_t* l_root = reinterpret_cast<uint64_t*>(buf);
  float* weights = reinterpret_cast<float*>(l_root + draws);
  int rc = euler_gpu_get_edge_sum_weight(g, stream, nodes_dev, cells, edge_types_host,
                                         k, weights);
  if (rc == EULER_GPU_OK)
    rc = euler_gpu_sample_root(stream, seed, call_id, nodes_dev, weights, batch, n,
                               count, default_node, l_root);
  if (rc == EULER_GPU_OK)
    rc = euler_gpu_sample_layer(g, stream, seed, call_id, l_root, draws,
                                edge_types_host, k, default_node, out_dev, nullptr,
                                nullptr);
  hipError_t f = hipFreeAsync(buf, st);
  if (rc != EULER_GPU_OK) return rc;
  EG_HIP(f);
  return EULER_GPU_OK;
}

int euler_gpu_local_sample_layer(void* stream, uint64_t seed, uint32_t call_id,
                                 const int32_t* idx_dev, const uint64_t* ids_dev,
                                 const float* w_dev, const int32_t* t_dev, int64_t total,
                                 int64_t batch, int32_t n, int32_t m,
                                 const char* weight_func, int64_t default_node,
                                 uint64_t* out_id_dev, float* out_w_dev,
                                 int32_t* out_t_dev) {
  if (batch < 0 || n <= 0 || m < 0 || total < 0)
    return Fail(EULER_GPU_EINVAL, "local_sample_layer: bad sizes");
  if (batch == 0 || m == 0) return EULER_GPU_OK;
  if (!idx_dev || !out_id_dev || !out_w_dev || !out_t_dev ||
      (total > 0 && (!ids_dev || !w_dev || !t_dev)))
    return Fail(EULER_GPU_EINVAL, "local_sample_layer: null buffer");
  hipStream_t st = (hipStream_t)stream;
  // ---- the candidate tables are built where the reference builds them: on the
  // host, in a std::unordered_map<std::string, ...> per batch row - the ORDER of
  // the candidates is that container's iteration order (local_sample_layer_op.cc:
  // 73-121), a property of libstdc++ this library shares with the reference.
  const int64_t R = batch * n;
  std::vector<int32_t> idx((size_t)R * 2), t((size_t)total);
  std::vector<uint64_t> ids((size_t)total);
  std::vector<float> w((size_t)total);
  EG_HIP(hipMemcpyAsync(idx.data(), idx_dev, (size_t)R * 8, hipMemcpyDeviceToHost, st));
  if (total > 0) {
    EG_HIP(hipMemcpyAsync(ids.data(), ids_dev, (size_t)total * 8, hipMemcpyDeviceToHost, st));
    EG_HIP(hipMemcpyAsync(w.data(), w_dev, (size_t)total * 4, hipMemcpyDeviceToHost, st));
    EG_HIP(hipMemcpyAsync(t.data(), t_dev, (size_t)total * 4, hipMemcpyDeviceToHost, st));
  }
  EG_HIP(hipStreamSynchronize(st));
  LocalLayerTables tb;
  if (!BuildLocalLayerTables(idx.data(), ids.data(), w.data(), t.data(), total, batch, n,
                             weight_func && std::string(weight_func) == "sqrt", &tb))
    return Fail(EULER_GPU_EINVAL, "local_sample_layer: idx does not index the values");
  std::vector<int64_t>& seg = tb.seg;
  std::vector<uint64_t>& u_id = tb.u_id;
  std::vector<float>&u_w = tb.u_w, &sum_w = tb.sum_w;
  std::vector<int32_t>& u_t = tb.u_t;
  const size_t U = u_id.size();
  uint8_t* buf = nullptr;
  const size_t bytes = (size_t)(batch + 1) * 8 + U * 8 + U * 12 + 64;
  EG_HIP(hipMallocAsync((void**)&buf, bytes, st));
  LocalLayerArgs a{};
  int64_t* d_seg = reinterpret_cast<int64_t*>(buf);
  uint64_t* d_id = reinterpret_cast<uint64_t*>(d_seg + batch + 1);
  float* d_w = reinterpret_cast<float*>(d_id + U);
  float* d_sum = d_w + U;
  int32_t* d_t = reinterpret_cast<int32_t*>(d_sum + U);
  hipError_t c = hipMemcpyAsync(d_seg, seg.data(), (size_t)(batch + 1) * 8, hipMemcpyHostToDevice, st);
  if (U > 0) {
    if (c == hipSuccess) c = hipMemcpyAsync(d_id, u_id.data(), U * 8, hipMemcpyHostToDevice, st);
    if (c == hipSuccess) c = hipMemcpyAsync(d_w, u_w.data(), U * 4, hipMemcpyHostToDevice, st);
    if (c == hipSuccess) c = hipMemcpyAsync(d_sum, sum_w.data(), U * 4, hipMemcpyHostToDevice, st);
    if (c == hipSuccess) c = hipMemcpyAsync(d_t, u_t.data(), U * 4, hipMemcpyHostToDevice, st);
  }
  a.seg = d_seg; a.u_id = d_id; a.u_w = d_w; a.u_t = d_t; a.sum_w = d_sum;
  a.batch = batch; a.m = m; a.seed = seed; a.call_id = call_id;
  // memset(out, default_node, ...) fills BYTES with the value's low byte (:130)
  a.fill_id = 0x0101010101010101ULL * (uint64_t)(uint8_t)default_node;
  a.out_id = out_id_dev; a.out_w = out_w_dev; a.out_t = out_t_dev;
  const int block = 256;
  if (c == hipSuccess) {
    hipLaunchKernelGGL(LocalSampleLayerKernel, dim3(GridFor(batch * m, block)), dim3(block), 0,
                       st, a);
    c = hipGetLastError();
  }
  hipError_t y = hipStreamSynchronize(st);      // the host vectors feed async copies
  hipError_t f = hipFreeAsync(buf, st);
  EG_HIP(c); EG_HIP(y); EG_HIP(f);
  return EULER_GPU_OK;
}

int euler_gpu_get_node_type(const euler_gpu_graph* g, void* stream,
                            const uint64_t* ids_dev, int64_t n, int32_t* out_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "get_node_type: null graph");
  if (n < 0) return Fail(EULER_GPU_EINVAL, "get_node_type: n < 0");
  if (n == 0) return EULER_GPU_OK;
  if (!ids_dev || !out_dev) return Fail(EULER_GPU_EINVAL, "get_node_type: null buffer");
  const int block = 256;
  hipLaunchKernelGGL(NodeTypeKernel, dim3(GridFor(n, block)), dim3(block), 0,
                     (hipStream_t)stream, g->view, g->node_type_dev, ids_dev, n, out_dev);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

int euler_gpu_sample_n_with_types(const euler_gpu_graph* g, void* stream, uint64_t seed,
                                  uint32_t call_id, const int32_t* types_dev, int64_t n,
                                  int32_t count, uint64_t* out_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "sample_n_with_types: null graph");
  if (!g->has_sampler)
    return Fail(EULER_GPU_ENOGRAPH, "sample_n_with_types: graph has no global sampler");
  if (n < 0 || count < 0) return Fail(EULER_GPU_EINVAL, "sample_n_with_types: bad sizes");
  if (n == 0 || count == 0) return EULER_GPU_OK;
  if (!types_dev || !out_dev)
    return Fail(EULER_GPU_EINVAL, "sample_n_with_types: null buffer");
  hipStream_t st = (hipStream_t)stream;
  int32_t* bad = nullptr;
  EG_HIP(hipMallocAsync((void**)&bad, sizeof(int32_t), st));
  EG_HIP(hipMemsetAsync(bad, 0, sizeof(int32_t), st));
  const int block = 256;
  hipLaunchKernelGGL(SampleNWithTypesKernel, dim3(GridFor(n * count, block)), dim3(block),
                     0, st, g->sampler, types_dev, n, count, seed, call_id, out_dev, bad);
  hipError_t e = hipGetLastError();
  int32_t bad_host = 0;
  hipError_t c = hipMemcpyAsync(&bad_host, bad, sizeof(int32_t), hipMemcpyDeviceToHost, st);
  hipError_t y = hipStreamSynchronize(st);
  hipError_t f = hipFreeAsync(bad, st);
  EG_HIP(e); EG_HIP(c); EG_HIP(y); EG_HIP(f);
  if (bad_host)
    return Fail(EULER_GPU_EEMPTY,
                "sample_n_with_types: a listed type is unknown or has zero weight");
  return EULER_GPU_OK;
}

size_t euler_gpu_sparse_get_adj_workspace(int64_t batch, int32_t n, int32_t m) {
  if (batch < 0 || n < 0 || m < 0) return 0;
  const int64_t R = batch * n;
  return (size_t)(R + 1) * 8 + (size_t)R * (size_t)((m + 63) / 64) * 8;
}

int euler_gpu_sparse_get_adj(const euler_gpu_graph* g, void* stream,
                             const uint64_t* roots_dev, const uint64_t* l_nb_dev,
                             int64_t batch, int32_t n, int32_t m,
                             const int32_t* edge_types_host, int32_t k,
                             void* workspace_dev, int32_t* idx_dev,
                             int64_t* total_host, uint64_t* out_id_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "sparse_get_adj: null graph");
  if (batch < 0 || n < 0 || m < 0)
    return Fail(EULER_GPU_EINVAL, "sparse_get_adj: negative size");
  AdjArgs a{};
  int rc = FillTypes(edge_types_host, k, &a.tl, "sparse_get_adj");
  if (rc != EULER_GPU_OK) return rc;
  const int64_t R = batch * n;
  if (R == 0) { if (total_host) *total_host = 0; return EULER_GPU_OK; }
  if (!roots_dev || !idx_dev || !workspace_dev || (m > 0 && !l_nb_dev))
    return Fail(EULER_GPU_EINVAL, "sparse_get_adj: null buffer");
  if ((int64_t)m * n * batch > 0x7fffffffLL)
    return Fail(EULER_GPU_EINVAL, "sparse_get_adj: result offsets exceed int32");
  hipStream_t st = (hipStream_t)stream;
  a.roots = roots_dev; a.l_nb = l_nb_dev; a.batch = batch; a.n = n; a.m = m;
  const AdjWorkspace w = SplitAdjWorkspace(workspace_dev, R, m);
  const int block = 256;
  if (out_id_dev == nullptr) {
    rc = BuildAdjMask(g, st, a, 0, w);
    if (rc != EULER_GPU_OK) return rc;
    hipLaunchKernelGGL(AdjOffsetsToIdxKernel, dim3((unsigned)((R + block - 1) / block)),
                       dim3(block), 0, st, w.off, R, idx_dev);
    int64_t total = 0;
    EG_HIP(hipMemcpyAsync(&total, w.off + R, 8, hipMemcpyDeviceToHost, st));
    EG_HIP(hipStreamSynchronize(st));
    if (total_host) *total_host = total;
    return EULER_GPU_OK;
  }
  if (m == 0) return EULER_GPU_OK;
  hipLaunchKernelGGL(AdjFillKernel, dim3(GridFor(R * w.words, block)), dim3(block), 0, st,
                     w.mask, l_nb_dev, R, n, m, w.words, 0, w.off, out_id_dev,
                     (int64_t*)nullptr, (int64_t*)nullptr);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

int euler_gpu_sparse_get_adj_tf(const euler_gpu_graph* g, void* stream,
                                const uint64_t* nodes_dev, const uint64_t* nb_nodes_dev,
                                int64_t batch, int32_t n, int32_t m,
                                const int32_t* edge_types_host, int32_t k,
                                void* workspace_dev, int64_t* nnz_host,
                                int64_t* indices_dev, int64_t* values_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "sparse_get_adj_tf: null graph");
  if (batch < 0 || n < 0 || m < 0)
    return Fail(EULER_GPU_EINVAL, "sparse_get_adj_tf: negative size");
  AdjArgs a{};
  int rc = FillTypes(edge_types_host, k, &a.tl, "sparse_get_adj_tf");
  if (rc != EULER_GPU_OK) return rc;
  const int64_t R = batch * n;
  if (R == 0 || m == 0) { if (nnz_host) *nnz_host = 0; return EULER_GPU_OK; }
  if (!nodes_dev || !nb_nodes_dev || !workspace_dev)
    return Fail(EULER_GPU_EINVAL, "sparse_get_adj_tf: null buffer");
  hipStream_t st = (hipStream_t)stream;
  a.roots = nodes_dev; a.l_nb = nb_nodes_dev; a.batch = batch; a.n = n; a.m = m;
  const AdjWorkspace w = SplitAdjWorkspace(workspace_dev, R, m);
  const int block = 256;
  if (indices_dev == nullptr) {
    rc = BuildAdjMask(g, st, a, 1, w);
    if (rc != EULER_GPU_OK) return rc;
    int64_t total = 0;
    EG_HIP(hipMemcpyAsync(&total, w.off + R, 8, hipMemcpyDeviceToHost, st));
    EG_HIP(hipStreamSynchronize(st));
    if (nnz_host) *nnz_host = total;
    return EULER_GPU_OK;
  }
  if (!values_dev) return Fail(EULER_GPU_EINVAL, "sparse_get_adj_tf: null values");
  hipLaunchKernelGGL(AdjFillKernel, dim3(GridFor(R * w.words, block)), dim3(block), 0, st,
                     w.mask, nb_nodes_dev, R, n, m, w.words, 1, w.off, (uint64_t*)nullptr,
                     indices_dev, values_dev);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

// The owner's half of SparseGetAdj on a sharded graph: the hit mask of the sources this
// shard OWNS (a source without a row here contributes zeros), [batch * n][(m + 63) / 64]
// uint64.  The masks of all shards OR together to the mask of the whole graph - an id has
// one owner - so a requester needs 8 bytes per 64 candidates and source from every shard
// instead of the sources' whole rows (core/kernels/sparse_get_adj_op.cc:35-92 is the
// per-shard op the reference runs remotely).
int euler_gpu_sparse_adj_mask(const euler_gpu_graph* g, void* stream, const uint64_t* nodes_dev,
                              const uint64_t* nb_nodes_dev, int64_t batch, int32_t n, int32_t m,
                              const int32_t* edge_types_host, int32_t k, uint64_t* mask_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "sparse_adj_mask: null graph");
  if (batch < 0 || n < 0 || m < 0) return Fail(EULER_GPU_EINVAL, "sparse_adj_mask: negative size");
  AdjArgs a{};
  int rc = FillTypes(edge_types_host, k, &a.tl, "sparse_adj_mask");
  if (rc != EULER_GPU_OK) return rc;
  const int64_t R = batch * n;
  if (R == 0 || m == 0) return EULER_GPU_OK;
  if (!nodes_dev || !nb_nodes_dev || !mask_dev)
    return Fail(EULER_GPU_EINVAL, "sparse_adj_mask: null buffer");
  a.roots = nodes_dev; a.l_nb = nb_nodes_dev; a.batch = batch; a.n = n; a.m = m;
  return BuildAdjMaskOnly(g, (hipStream_t)stream, a, mask_dev, (m + 63) / 64);
}

// ... and the requester's: the TF SparseGetAdj triple (tf_euler/kernels/sparse_get_adj_op.cc:
// 92-124, explicit zero at (b, n-1, m-1) included) from a hit mask.  Two calls as
// euler_gpu_sparse_get_adj_tf: indices_dev == NULL sizes (nnz_host, stream sync; the offsets
// stay in workspace_dev), the second fills.  workspace: 8 * (batch * n + 1) bytes.
int euler_gpu_sparse_adj_from_mask_tf(void* stream, const uint64_t* mask_dev, int64_t batch,
                                      int32_t n, int32_t m, void* workspace_dev,
                                      int64_t* nnz_host, int64_t* indices_dev,
                                      int64_t* values_dev) {
  if (batch < 0 || n < 0 || m < 0)
    return Fail(EULER_GPU_EINVAL, "sparse_adj_from_mask_tf: negative size");
  const int64_t R = batch * n;
  if (R == 0 || m == 0) { if (nnz_host) *nnz_host = 0; return EULER_GPU_OK; }
  if (!mask_dev || !workspace_dev)
    return Fail(EULER_GPU_EINVAL, "sparse_adj_from_mask_tf: null buffer");
  hipStream_t st = (hipStream_t)stream;
  int64_t* off = static_cast<int64_t*>(workspace_dev);
  const int32_t words = (m + 63) / 64;
  if (indices_dev == nullptr) {
    const int rc = AdjMaskOffsets(st, mask_dev, R, n, m, words, 1, off);
    if (rc != EULER_GPU_OK) return rc;
    int64_t total = 0;
    EG_HIP(hipMemcpyAsync(&total, off + R, 8, hipMemcpyDeviceToHost, st));
    EG_HIP(hipStreamSynchronize(st));
    if (nnz_host) *nnz_host = total;
    return EULER_GPU_OK;
  }
  if (!values_dev) return Fail(EULER_GPU_EINVAL, "sparse_adj_from_mask_tf: null values");
  hipLaunchKernelGGL(AdjFillKernel, dim3(GridFor(R * words, 256)), dim3(256), 0, st, mask_dev,
                     (const uint64_t*)nullptr, R, n, m, words, 1, off, (uint64_t*)nullptr,
                     indices_dev, values_dev);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

}  // extern "C"
