// Message-passing (scatter / gather), ID_UNIQUE / *_GATHER and shard
// split / merge kernels for gfx950, plus their C-ABI entry points.
//
// rocPRIM (via hipCUB) is used only for the two plain library primitives this
// file needs - a device-wide exclusive scan and a stable radix sort; every
// domain kernel is written here.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <atomic>
#include <map>
#include <new>
#include <mutex>
#include <utility>

#include "device_fns.h"

struct euler_gpu_front {
  int64_t* stage = nullptr;      // pinned + mapped: [kMaxShards + 1] bucket starts, then the sequence word
  int64_t* stage_dev = nullptr;  // the same buffer as the kernels address it
  hipEvent_t done = nullptr;
  int64_t seq = 0;               // sequence number of the call in flight (the kernel echoes it)
  int32_t shards = 0;
  int32_t pending = 0;           // a begin without its end
};

namespace euler_gpu {

int ExclusiveScanI64(hipStream_t stream, const int64_t* in, int64_t* out,
                     int64_t n) {
  size_t tmp_bytes = 0;
  EG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, in, out, (int)n,
                                          stream));
  void* tmp = nullptr;
  EG_HIP(hipMallocAsync(&tmp, tmp_bytes + 16, stream));
  EG_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, in, out, (int)n,
                                          stream));
  EG_HIP(hipFreeAsync(tmp, stream));
  return EULER_GPU_OK;
}

// ------------------------------------------------------------------------
// MPGather (tf_euler/kernels/gather_op.cc:46-51): out[i,:] = params[idx[i],:]
// Rows are D contiguous floats: lanes sweep a row with 16-byte accesses when
// D % 4 == 0 (coalesced 1 KiB per wave-instruction), several rows per wave
// when D is small.
// ------------------------------------------------------------------------
template <typename V>
__global__ __launch_bounds__(256) void GatherRowsKernel(
    const V* __restrict__ params, const int32_t* __restrict__ idx, int64_t e,
    int64_t dv, V* __restrict__ out) {
  const int64_t total = e * dv;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total;
       x += stride) {
    const int64_t i = x / dv;
    const int64_t c = x - i * dv;
    out[x] = params[(int64_t)idx[i] * dv + c];
  }
}

// ------------------------------------------------------------------------
// MPScatterAdd / MPScatterMax (tf_euler/kernels/scatter_op.cc:32-92).
// The reference loops over the E updates in input order; fp32 addition is not
// associative, so the result depends on that order.  We keep it: updates are
// grouped by destination with a STABLE sort (identity when the indices are
// already non-decreasing, the common case for sampled blocks), and each output
// element then accumulates its segment sequentially in input order - a
// segment-reduce with one lane per output column, bit-identical to the
// reference, no atomics, zero-/-1e9-fill fused into the same pass.
// ------------------------------------------------------------------------
__global__ void IsSortedKernel(const int32_t* idx, int64_t e, int32_t* flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i + 1 < e && idx[i] > idx[i + 1]) *flag = 0;
}

__global__ void IotaKernel(uint32_t* v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (uint32_t)i;
}

__device__ __forceinline__ int64_t LowerBound(const int32_t* a, int64_t n,
                                              int32_t key) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Segment [b, en) of destination r in the grouped key array.  Sampled blocks
// scatter `count` updates to every destination in order (keys = 0,0,..,1,1,..): the
// proportional guess r * e / size is then exact and costs two loads; anything else
// falls back to the bisection.
__device__ __forceinline__ int64_t SegStart(const int32_t* keys, int64_t e, int32_t size,
                                            int64_t r) {
  // r == size: the end of the LAST destination's segment - the first key >= size, not e:
  // out-of-range scatter indices (undefined behaviour in the reference,
  // tf_euler/kernels/scatter_op.cc:27-105) are left out instead of being folded into
  // row size - 1
  if (r >= size) return (e > 0 && keys[e - 1] >= size) ? LowerBound(keys, e, size) : e;
  const int64_t g = r * e / size;
  if ((g == 0 || keys[g - 1] < (int32_t)r) && (g == e || keys[g] >= (int32_t)r)) return g;
  return LowerBound(keys, e, (int32_t)r);
}

// Where destination r's updates are: a grouped key array (scatter: bisected / guessed),
// explicit offsets (segment reduce), or `count` updates per destination.
struct SegSpec {
  const int32_t* keys;
  const int64_t* ptr;      // [size + 1] when keys == nullptr (nullptr: uniform `count`)
  int64_t count;
  int64_t e;
  int32_t size;
};

__device__ __forceinline__ void SegBounds(const SegSpec& s, int64_t r, int64_t* b, int64_t* en) {
  if (s.keys != nullptr) {
    *b = SegStart(s.keys, s.e, s.size, r);
    *en = SegStart(s.keys, s.e, s.size, r + 1);
  } else if (s.ptr != nullptr) {
    *b = s.ptr[r];
    *en = s.ptr[r + 1];
  } else {
    *b = r * s.count;
    *en = *b + s.count;
  }
}

// One wave-slot per output row: blockDim = (64, 4): 4 rows per block,
// 64 lanes over the columns.  keys[] = destination of the p-th update in
// grouped order; perm[p] = original update index (nullptr = identity).
// MODE 0 = add, 1 = max, 2 = mean: the reference's scatter_mean is
// scatter_add(x) / (scatter_add(ones) + 1e-7) (euler_ops/mp_ops.py:65-69); the
// count of a destination is its segment length (an exact f32 below 2^24), so the
// same correctly rounded f32 add and divide give the same bits in one pass.
// gsrc != nullptr: update p is row gsrc[p] of `upd` (the gather of the message passing
// step folded into the reduce: the E x d block of gathered rows is never written).
// gstride = 2: gsrc points at int64 ids (what a sampler returns) and every index is the low
// word of its id - the int32 a cast would have produced, without the cast's pass.
template <int MODE>
__global__ __launch_bounds__(256) void SegmentReduceKernel(
    const float* __restrict__ upd, const SegSpec seg,
    const uint32_t* __restrict__ perm, const int32_t* __restrict__ gsrc, int64_t d,
    float* __restrict__ out, const int32_t gstride, const uint32_t row_max) {
  const int lane = threadIdx.x;
  const int32_t size = seg.size;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.y + threadIdx.y; r < size;
       r += (int64_t)gridDim.x * blockDim.y) {
    int64_t b, en;
    SegBounds(seg, r, &b, &en);
    constexpr bool IS_MAX = MODE == 1;
    const float denom = __fadd_rn((float)(en - b), 1e-7f);
    for (int64_t c = lane; c < d; c += 64) {
      float acc = IS_MAX ? (float)-1e9 : 0.f;   // scatter_op.cc:47,78
      int64_t p = b;
      // the additions stay in input order (fp32 is not associative); only the
      // loads are issued eight at a time so that their latencies overlap
      for (; p + 8 <= en; p += 8) {
        float v[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) {
          int64_t src = perm ? (int64_t)perm[p + x] : p + x;
          if (gsrc) src = (int32_t)min((uint32_t)gsrc[src * gstride], row_max);
          v[x] = upd[src * d + c];
        }
#pragma unroll
        for (int x = 0; x < 8; ++x) {
          if (IS_MAX) { if (v[x] > acc) acc = v[x]; }
          else acc = __fadd_rn(acc, v[x]);
        }
      }
      for (; p < en; ++p) {
        int64_t src = perm ? (int64_t)perm[p] : p;
        if (gsrc) src = (int32_t)min((uint32_t)gsrc[src * gstride], row_max);
        const float v = upd[src * d + c];
        if (IS_MAX) { if (v > acc) acc = v; }
        else acc = __fadd_rn(acc, v);
      }
      out[r * d + c] = MODE == 2 ? __fdiv_rn(acc, denom) : acc;
    }
  }
}

// The same reduce for d % 4 == 0 with d / 4 a divisor of 64 (d = 4 .. 256): a lane
// owns FOUR adjacent columns (16-byte loads and stores), d / 4 lanes a row, so a
// wave reduces 256 / d rows at once.  Every column still adds its updates in input
// order - identical bits, a quarter of the memory instructions.
template <int MODE>
__global__ __launch_bounds__(256) void SegmentReduceVec4Kernel(
    const float* __restrict__ upd, const SegSpec seg,
    const uint32_t* __restrict__ perm, const int32_t* __restrict__ gsrc, int32_t d4,
    float* __restrict__ out, const int32_t gstride, const uint32_t row_max) {
  constexpr bool IS_MAX = MODE == 1;
  const int32_t size = seg.size;
  const int32_t rows_per_wave = 64 / d4;
  const int32_t sub = threadIdx.x / d4, cl = threadIdx.x - sub * d4;
  const int64_t rows_per_block = (int64_t)blockDim.y * rows_per_wave;
  for (int64_t r = (int64_t)blockIdx.x * rows_per_block + threadIdx.y * rows_per_wave + sub;
       r < size; r += (int64_t)gridDim.x * rows_per_block) {
    int64_t b, en;
    SegBounds(seg, r, &b, &en);
    const float denom = __fadd_rn((float)(en - b), 1e-7f);
    const float init = IS_MAX ? (float)-1e9 : 0.f;
    float4 acc = make_float4(init, init, init, init);
    const float4* u4 = reinterpret_cast<const float4*>(upd);
    int64_t p = b;
    // the additions stay in input order; the loads (for gathered rows: the row numbers
    // first, then the rows) are issued eight at a time so that their latencies overlap
    for (; p + 8 <= en; p += 8) {
      int64_t src[8];
#pragma unroll
      for (int x = 0; x < 8; ++x) src[x] = perm ? (int64_t)perm[p + x] : p + x;
      if (gsrc) {
#pragma unroll
        for (int x = 0; x < 8; ++x) src[x] = (int32_t)min((uint32_t)gsrc[src[x] * gstride], row_max);
      }
      float4 v[8];
#pragma unroll
      for (int x = 0; x < 8; ++x) v[x] = u4[src[x] * d4 + cl];
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        if (IS_MAX) {
          acc.x = v[x].x > acc.x ? v[x].x : acc.x; acc.y = v[x].y > acc.y ? v[x].y : acc.y;
          acc.z = v[x].z > acc.z ? v[x].z : acc.z; acc.w = v[x].w > acc.w ? v[x].w : acc.w;
        } else {
          acc.x = __fadd_rn(acc.x, v[x].x); acc.y = __fadd_rn(acc.y, v[x].y);
          acc.z = __fadd_rn(acc.z, v[x].z); acc.w = __fadd_rn(acc.w, v[x].w);
        }
      }
    }
    for (; p + 4 <= en; p += 4) {
      float4 v[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        int64_t src = perm ? (int64_t)perm[p + x] : p + x;
        if (gsrc) src = (int32_t)min((uint32_t)gsrc[src * gstride], row_max);
        v[x] = u4[src * d4 + cl];
      }
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        if (IS_MAX) {
          acc.x = v[x].x > acc.x ? v[x].x : acc.x; acc.y = v[x].y > acc.y ? v[x].y : acc.y;
          acc.z = v[x].z > acc.z ? v[x].z : acc.z; acc.w = v[x].w > acc.w ? v[x].w : acc.w;
        } else {
          acc.x = __fadd_rn(acc.x, v[x].x); acc.y = __fadd_rn(acc.y, v[x].y);
          acc.z = __fadd_rn(acc.z, v[x].z); acc.w = __fadd_rn(acc.w, v[x].w);
        }
      }
    }
    for (; p < en; ++p) {
      int64_t src = perm ? (int64_t)perm[p] : p;
      if (gsrc) src = (int32_t)min((uint32_t)gsrc[src * gstride], row_max);
      const float4 v = u4[src * d4 + cl];
      if (IS_MAX) {
        acc.x = v.x > acc.x ? v.x : acc.x; acc.y = v.y > acc.y ? v.y : acc.y;
        acc.z = v.z > acc.z ? v.z : acc.z; acc.w = v.w > acc.w ? v.w : acc.w;
      } else {
        acc.x = __fadd_rn(acc.x, v.x); acc.y = __fadd_rn(acc.y, v.y);
        acc.z = __fadd_rn(acc.z, v.z); acc.w = __fadd_rn(acc.w, v.w);
      }
    }
    if (MODE == 2) {
      acc.x = __fdiv_rn(acc.x, denom); acc.y = __fdiv_rn(acc.y, denom);
      acc.z = __fdiv_rn(acc.z, denom); acc.w = __fdiv_rn(acc.w, denom);
    }
    reinterpret_cast<float4*>(out)[r * d4 + cl] = acc;
  }
}

template <int MODE>
static int ScatterImpl(hipStream_t st, const float* upd, const int32_t* idx,
                       int64_t e, int64_t d, int32_t size, float* out,
                       const int32_t* gsrc = nullptr) {
  if (e < 0 || d < 0 || size < 0) return Fail(EULER_GPU_EINVAL, "scatter: bad shape");
  if (size == 0 || d == 0) return EULER_GPU_OK;
  if (!out || (e > 0 && (!upd || !idx)))
    return Fail(EULER_GPU_EINVAL, "scatter: null buffer");
  if (e >= (1LL << 31)) return Fail(EULER_GPU_EINVAL, "scatter: e >= 2^31");
  const int32_t* keys = idx;
  const uint32_t* perm = nullptr;
  // stream-ordered scratch of the unsorted path, released on every exit
  struct Scratch {
    hipStream_t st;
    void* p = nullptr;
    explicit Scratch(hipStream_t s) : st(s) {}
    ~Scratch() { if (p) (void)hipFreeAsync(p, st); }
  } sc(st);
  void*& scratch = sc.p;
  if (e > 1) {
    int32_t* flag = nullptr;
    EG_HIP(hipMallocAsync((void**)&flag, 16, st));
    int32_t one = 1, sorted = 1;
    EG_HIP(hipMemcpyAsync(flag, &one, 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(IsSortedKernel, dim3((e + 255) / 256), dim3(256), 0, st, idx,
                       e, flag);
    EG_HIP(hipMemcpyAsync(&sorted, flag, 4, hipMemcpyDeviceToHost, st));
    EG_HIP(hipStreamSynchronize(st));
    EG_HIP(hipFreeAsync(flag, st));
    if (!sorted) {
      // stable sort of (destination, original position)
      const size_t n = (size_t)e;
      const size_t bytes = n * (4 + 4 + 4 + 4) + 64;
      EG_HIP(hipMallocAsync(&scratch, bytes, st));
      int32_t* keys_out = (int32_t*)scratch;
      uint32_t* vals_in = (uint32_t*)(keys_out + n);
      uint32_t* vals_out = vals_in + n;
      hipLaunchKernelGGL(IotaKernel, dim3((e + 255) / 256), dim3(256), 0, st,
                         vals_in, e);
      size_t tmp_bytes = 0;
      EG_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, idx, keys_out,
                                                vals_in, vals_out, (int)e, 0, 32,
                                                st));
      void* tmp = nullptr;
      EG_HIP(hipMallocAsync(&tmp, tmp_bytes + 16, st));
      EG_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, idx, keys_out,
                                                vals_in, vals_out, (int)e, 0, 32,
                                                st));
      EG_HIP(hipFreeAsync(tmp, st));
      keys = keys_out;
      perm = vals_out;
    }
  }
  const dim3 block(64, 4);
  const int64_t d4 = d / 4;
  if (d % 4 == 0 && d4 <= 64 && 64 % d4 == 0 && ((uintptr_t)upd % 16 == 0) &&
      ((uintptr_t)out % 16 == 0)) {
    const int64_t rows_per_block = 4 * (64 / d4);
    int64_t blocks = ((int64_t)size + rows_per_block - 1) / rows_per_block;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(SegmentReduceVec4Kernel<MODE>, dim3((unsigned)blocks), block, 0, st, upd,
                       SegSpec{keys, nullptr, 0, e, size}, perm, gsrc, (int32_t)d4, out, 1, 0xFFFFFFFFu);
  } else {
    int64_t blocks = ((int64_t)size + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(SegmentReduceKernel<MODE>, dim3((unsigned)blocks), block, 0,
                       st, upd, SegSpec{keys, nullptr, 0, e, size}, perm, gsrc, d, out, 1, 0xFFFFFFFFu);
  }
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

template <int MODE>
static int SegmentReduceImpl(hipStream_t st, const float* params, const int32_t* gsrc,
                             const int64_t* seg_ptr, int64_t count, int64_t d, int32_t size,
                             float* out, int32_t gstride = 1, uint32_t row_max = 0xFFFFFFFFu) {
  const dim3 block(64, 4);
  const int64_t d4 = d / 4;
  const SegSpec seg{nullptr, seg_ptr, count, 0, size};
  if (d % 4 == 0 && d4 <= 64 && 64 % d4 == 0 && ((uintptr_t)params % 16 == 0) &&
      ((uintptr_t)out % 16 == 0)) {
    const int64_t rows_per_block = 4 * (64 / d4);
    int64_t blocks = ((int64_t)size + rows_per_block - 1) / rows_per_block;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(SegmentReduceVec4Kernel<MODE>, dim3((unsigned)blocks), block, 0, st, params,
                       seg, nullptr, gsrc, (int32_t)d4, out, gstride, row_max);
  } else {
    int64_t blocks = ((int64_t)size + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(SegmentReduceKernel<MODE>, dim3((unsigned)blocks), block, 0, st, params, seg,
                       nullptr, gsrc, d, out, gstride, row_max);
  }
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

// ------------------------------------------------------------------------
// Post-process of API_GET_NB_NODE (core/kernels/get_neighbor_op.cc:117-168):
// `order_by id|weight [desc]` sorts every row of a FillNeighbor-layout result,
// `limit k` keeps its first k entries.  Rows are segments of one flat array,
// so the sort is one segmented radix sort of (key, position) pairs (rocPRIM,
// stable: equal keys keep storage order - the reference sorts with a
// non-strict comparator, so its order of equal keys is undefined), then one
// wave per row moves the surviving entries to their packed place.
// ------------------------------------------------------------------------
struct IdxEdge {
  const int32_t* idx;
  int32_t which;
  __host__ __device__ __forceinline__ int32_t operator()(const int32_t& i) const {
    return idx[2 * i + which];
  }
};

// Rows of up to kWaveSortMax entries (most rows of a batch) are ranked inside one
// wave: every entry counts the entries that sort before it - smaller key, or
// equal key and earlier position (stable, as the radix sort used for longer
// rows).  perm[b + rank] = b + j.  Longer rows are left to the segmented radix
// sort, which sees the short rows as empty segments.
constexpr int32_t kWaveSortMax = 64;     // rows up to this length are ranked by one wave (longer
                                         // ones cost len^2 / 64 steps there: measured slower at 1024)

struct IdxEdgeLong {
  const int32_t* idx;
  int32_t which;
  __host__ __device__ __forceinline__ int32_t operator()(const int32_t& i) const {
    const int32_t b = idx[2 * i], e = idx[2 * i + 1];
    if (e - b <= kWaveSortMax) return b;   // empty segment: handled by WaveRankKernel
    return which ? e : b;
  }
};

// One wave per row: lane j holds entries j, j + 64, ... and counts, for each,
// the entries that sort before it (the keys of a row are read by all lanes at
// the same address: one request each, then the L1).
template <typename K, bool DESC>
__global__ __launch_bounds__(256) void SmallRowRankKernel(
    const int32_t* __restrict__ idx, int64_t n, const K* __restrict__ keys,
    int32_t* __restrict__ perm) {
  const int lane = threadIdx.x & 63;
  const int64_t waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < n;
       i += waves) {
    const int32_t b = idx[2 * i], len = idx[2 * i + 1] - b;
    if (len <= 0 || len > kWaveSortMax) continue;
    if (len <= 64) {
      const bool live = lane < len;
      const K mine = live ? keys[b + lane] : K();
      int32_t rank = 0;
      for (int32_t o = 0; o < len; ++o) {
        const K other = __shfl(mine, o);
        const bool before = DESC ? (other > mine) : (other < mine);
        rank += (before || (other == mine && o < lane)) ? 1 : 0;
      }
      if (live) perm[b + rank] = b + lane;
    } else {
      for (int32_t j = lane; j < len; j += 64) {
        const K mine = keys[b + j];
        int32_t rank = 0;
        for (int32_t o = 0; o < len; ++o) {
          const K other = keys[b + o];
          const bool before = DESC ? (other > mine) : (other < mine);
          rank += (before || (other == mine && o < j)) ? 1 : 0;
        }
        perm[b + rank] = b + j;
      }
    }
  }
}

__global__ void MaxRowLenKernel(const int32_t* idx, int64_t n, int32_t* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int32_t l = i < n ? idx[2 * i + 1] - idx[2 * i] : 0;
  for (int off = 32; off > 0; off >>= 1) l = max(l, __shfl_xor(l, off));
  if ((threadIdx.x & 63) == 0 && l > kWaveSortMax) atomicMax(out, l);   // only long rows report
}

__global__ void IotaKernel(int32_t* p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (int32_t)i;
}

__global__ void LimitLenKernel(const int32_t* idx, int64_t n, int64_t limit,
                               int64_t* len) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    int64_t l = idx[2 * i + 1] - idx[2 * i];
    if (limit >= 0 && l > limit) l = limit;
    len[i] = l;
  }
}

// one wave per row: dst[new_off[i] + p] = src[perm ? perm[b + p] : b + p]
__global__ __launch_bounds__(256) void RepackRowsKernel(
    const int32_t* __restrict__ old_idx, const int64_t* __restrict__ new_len,
    const int64_t* __restrict__ new_off, const int32_t* __restrict__ perm, int64_t n,
    const uint64_t* __restrict__ s_id, const float* __restrict__ s_w,
    const int32_t* __restrict__ s_t, uint64_t* __restrict__ d_id,
    float* __restrict__ d_w, int32_t* __restrict__ d_t) {
  const int lane = threadIdx.x & 63;
  const int64_t waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < n;
       i += waves) {
    const int64_t b = old_idx[2 * i], len = new_len[i], o = new_off[i];
    for (int64_t p = lane; p < len; p += 64) {
      const int64_t src = perm ? (int64_t)perm[b + p] : b + p;
      d_id[o + p] = s_id[src];
      d_w[o + p] = s_w[src];
      d_t[o + p] = s_t[src];
    }
  }
}

__global__ void LenOffToIdxKernel(const int64_t* len, const int64_t* off, int64_t n,
                                  int32_t* idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    idx[2 * i] = (int32_t)off[i];
    idx[2 * i + 1] = (int32_t)(off[i] + len[i]);
  }
}

// TF GetTopKNeighbor dense fill (tf_euler/kernels/get_top_k_neighbor_op.cc:
// 70-75 prefill default_node / 0.0 / -1, :101-109 copy the row's entries).
__global__ __launch_bounds__(256) void NeighborToDenseKernel(
    const int32_t* __restrict__ idx, const uint64_t* __restrict__ ids,
    const float* __restrict__ w, const int32_t* __restrict__ t, int64_t n, int32_t k,
    int64_t default_node, int64_t* __restrict__ out_id, float* __restrict__ out_w,
    int32_t* __restrict__ out_t) {
  const int64_t total = n * (int64_t)k;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total;
       s += stride) {
    const int64_t i = s / k;
    const int32_t p = (int32_t)(s - i * k);
    const int32_t b = idx[2 * i], e = idx[2 * i + 1];
    const bool have = b + p < e;
    out_id[s] = have ? (int64_t)ids[b + p] : default_node;
    out_w[s] = have ? w[b + p] : 0.f;
    out_t[s] = have ? t[b + p] : -1;
  }
}

// ------------------------------------------------------------------------
// GetDenseFeature (tf_euler/kernels/get_dense_feature_op.cc:63-125 over
// Node::GetFloat32Feature, core/graph/node.cc:330-394).  One lane per output
// element: the dim lanes of a row read consecutive floats of the node's value
// block (one or two 128-byte lines for typical dims) and write consecutive
// floats of the output - a row gather, HBM-streaming bound; the row lookup and
// the two slot ends are the same address for the whole row (one request).
// Bytes: 4*dim in + 4*dim out per node (+ 8 id + 16 row metadata).
// ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void DenseFeatureKernel(
    const GraphView g, const uint64_t* __restrict__ nodes, int64_t n, int32_t fid,
    int32_t dim, float* __restrict__ out) {
  const int64_t total = n * (int64_t)dim;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total;
       s += stride) {
    const int64_t j = s / dim;
    const int32_t c = (int32_t)(s - j * dim);
    float v = 0.f;
    const int64_t row = FindRow(g, nodes[j]);
    if (row >= 0 && fid >= 0 && fid < g.n_float) {
      const int32_t* idx = g.feat_idx + (g.feat_uniform ? 0 : row * g.n_float);
      const int32_t pre = fid == 0 ? 0 : idx[fid - 1];
      const int32_t now = idx[fid];
      if (c < now - pre) {
        const int64_t base = g.feat_uniform ? row * g.feat_stride : g.feat_ptr[row];
        v = g.feat_val[base + pre + c];
      }
    }
    out[s] = v;
  }
}

// ------------------------------------------------------------------------
// ID_UNIQUE (core/kernels/id_unique_op.cc:35-64): first-occurrence order.
// Open-addressing table in HBM: every id claims a slot (atomicCAS on the key)
// and atomicMin's its position into the slot; the ids whose position equals
// the slot minimum are the first occurrences, and an exclusive scan of that
// flag over positions IS the reference's first-occurrence rank.
// The all-ones key is the empty marker; an id equal to it uses a side slot.
// ------------------------------------------------------------------------
constexpr uint64_t kEmptyKey = ~0ULL;

struct UniqueTable {
  unsigned long long* keys;   // [cap + 1]; slot `cap` is the side slot
  uint32_t* minpos;           // [cap + 1]
  int32_t* rank;              // [cap + 1]
  uint64_t mask;              // cap - 1
};

__global__ void UniqueInsertKernel(const uint64_t* ids, int64_t n, UniqueTable t,
                                   uint32_t* slot_of) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t id = ids[i];
  uint64_t h;
  if (id == kEmptyKey) {
    h = t.mask + 1;
  } else {
    h = Mix64(id) & t.mask;
    for (;;) {
      // look before the atomic (agent scope: never from this CU's stale L1):
      // device-scope atomics execute at the memory side on this multi-die part
      // (~0.5 ns each, serialised per address), a repeated id only needs a load
      unsigned long long old =
          __hip_atomic_load(&t.keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == kEmptyKey)
        old = atomicCAS(&t.keys[h], (unsigned long long)kEmptyKey,
                        (unsigned long long)id);
      if (old == kEmptyKey || old == id) break;
      h = (h + 1) & t.mask;
    }
  }
  // positions are handed out in launch order, so the slot usually already holds
  // a smaller one: only then skip the atomicMin
  if (__hip_atomic_load(&t.minpos[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >
      (uint32_t)i)
    atomicMin(&t.minpos[h], (uint32_t)i);
  slot_of[i] = (uint32_t)h;
}

__global__ void UniqueFlagKernel(int64_t n, UniqueTable t, const uint32_t* slot_of,
                                 int64_t* is_first) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) is_first[i] = t.minpos[slot_of[i]] == (uint32_t)i ? 1 : 0;
}

__global__ void UniqueEmitKernel(const uint64_t* ids, int64_t n, UniqueTable t,
                                 const uint32_t* slot_of, const int64_t* is_first,
                                 const int64_t* rank, uint64_t* unique_ids) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && is_first[i]) {
    unique_ids[rank[i]] = ids[i];
    t.rank[slot_of[i]] = (int32_t)rank[i];
  }
}

__global__ void UniqueGatherIdxKernel(int64_t n, UniqueTable t,
                                      const uint32_t* slot_of,
                                      int32_t* gather_idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) gather_idx[i] = t.rank[slot_of[i]];
}

// IDX_GATHER (core/kernels/idx_gather_op.cc:33-55)
__global__ void SegLenKernel(const int32_t* idx, const int32_t* gather_idx,
                             int64_t n, int64_t* len) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int32_t a = gather_idx[i] * 2;
    len[i] = idx[a + 1] - idx[a];
  }
}

__global__ void EmitIdxKernel(const int64_t* len, const int64_t* off, int64_t n,
                              int32_t* out_idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    out_idx[2 * i] = (int32_t)off[i];
    out_idx[2 * i + 1] = (int32_t)(off[i] + len[i]);
  }
}

// DATA_GATHER (core/kernels/data_gather_op.cc:33-46): 16 lanes per output row.
template <typename V>
__global__ __launch_bounds__(256) void DataGatherKernel(
    const V* data, const int32_t* idx, const int32_t* gather_idx,
    const int32_t* out_idx, int64_t n, V* out) {
  const int sub = threadIdx.x & 15;
  const int64_t g0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int64_t ng = ((int64_t)gridDim.x * blockDim.x) >> 4;
  for (int64_t i = g0; i < n; i += ng) {
    const int32_t a = gather_idx[i] * 2;
    const int32_t b = idx[a], e = idx[a + 1];
    const int32_t o = out_idx[2 * i];
    for (int32_t p = sub; p < e - b; p += 16) out[o + p] = data[b + p];
  }
}

// ------------------------------------------------------------------------
// ID_SPLIT (core/kernels/id_split_op.cc:46-99): stable bucket by owner.
// Stability (the reference pushes ids in input order) comes from ranking each
// id inside its bucket with a per-wave ballot prefix + a per-block exclusive
// offset obtained from a scan of per-block histograms.
// ------------------------------------------------------------------------
constexpr int kSplitBlock = 256;
constexpr int kMaxShards = 64;

__device__ __forceinline__ int32_t OwnerOf(uint64_t id, int32_t partitions,
                                           int32_t shards) {
  return (int32_t)((id % (uint64_t)partitions) % (uint64_t)shards);
}

__global__ __launch_bounds__(kSplitBlock) void SplitHistKernel(
    const uint64_t* ids, int64_t n, int32_t partitions, int32_t shards,
    int64_t* block_hist /* [shards, n_blocks] */) {
  __shared__ int32_t hist[kMaxShards];
  if (threadIdx.x < kMaxShards) hist[threadIdx.x] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * kSplitBlock + threadIdx.x;
  if (i < n) atomicAdd(&hist[OwnerOf(ids[i], partitions, shards)], 1);
  __syncthreads();
  if ((int)threadIdx.x < shards)
    block_hist[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = hist[threadIdx.x];
}

__global__ __launch_bounds__(kSplitBlock) void SplitScatterKernel(
    const uint64_t* ids, int64_t n, int32_t partitions, int32_t shards,
    const int64_t* block_off /* scanned [shards, n_blocks] */,
    uint64_t* shard_ids, int32_t* merge_idx) {
  __shared__ int32_t wave_cnt[kSplitBlock / 64][kMaxShards];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * kSplitBlock + threadIdx.x;
  const bool live = i < n;
  const uint64_t id = live ? ids[i] : 0;
  const int32_t own = live ? OwnerOf(id, partitions, shards) : -1;
  // rank of this lane among the lanes of its wave with the same owner
  int32_t rank_in_wave = 0, wave_total = 0;
  for (int32_t s = 0; s < shards; ++s) {
    const unsigned long long m = __ballot(own == s);
    if (own == s) {
      rank_in_wave = __popcll(m & ((1ULL << lane) - 1));
    }
    if (lane == 0) wave_cnt[wave][s] = __popcll(m);
  }
  (void)wave_total;
  __syncthreads();
  if (live) {
    int32_t before = 0;
    for (int w = 0; w < wave; ++w) before += wave_cnt[w][own];
    const int64_t pos =
        block_off[(int64_t)own * gridDim.x + blockIdx.x] + before + rank_in_wave;
    shard_ids[pos] = id;
    merge_idx[pos] = (int32_t)i;
  }
}

// IDX_MERGE / DATA_MERGE with fixed-size rows (idx_merge_op.cc:61-77,
// data_merge_op.cc:44-67): out[merge_idx[j]] = in[j], 4-byte granularity.
__global__ __launch_bounds__(256) void MergeRowsKernel(
    const uint32_t* in, const int32_t* merge_idx, int64_t n_rows, int64_t row_words,
    uint32_t* out) {
  const int64_t total = n_rows * row_words;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total;
       x += stride) {
    const int64_t j = x / row_words, c = x - j * row_words;
    out[(int64_t)merge_idx[j] * row_words + c] = in[x];
  }
}

}  // namespace euler_gpu

using namespace euler_gpu;

extern "C" {

int euler_gpu_gather(void* stream, const float* params_dev,
                     const int32_t* indices_dev, int64_t e, int64_t d,
                     int64_t n_params, float* out_dev) {
  (void)n_params;
  if (e < 0 || d < 0) return Fail(EULER_GPU_EINVAL, "gather: bad shape");
  if (e == 0 || d == 0) return EULER_GPU_OK;
  if (!params_dev || !indices_dev || !out_dev)
    return Fail(EULER_GPU_EINVAL, "gather: null buffer");
  hipStream_t st = (hipStream_t)stream;
  const int block = 256;
  const bool vec4 = (d % 4 == 0) && (((uintptr_t)params_dev | (uintptr_t)out_dev) % 16 == 0);
  if (vec4) {
    hipLaunchKernelGGL(GatherRowsKernel<float4>, dim3(GridFor(e * d / 4, block)),
                       dim3(block), 0, st, (const float4*)params_dev, indices_dev,
                       e, d / 4, (float4*)out_dev);
  } else {
    hipLaunchKernelGGL(GatherRowsKernel<float>, dim3(GridFor(e * d, block)),
                       dim3(block), 0, st, params_dev, indices_dev, e, d, out_dev);
  }
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

int euler_gpu_scatter_add(void* stream, const float* updates_dev,
                          const int32_t* indices_dev, int64_t e, int64_t d,
                          int32_t size, float* out_dev) {
  return ScatterImpl<0>((hipStream_t)stream, updates_dev, indices_dev, e, d,
                        size, out_dev);
}

int euler_gpu_scatter_mean(void* stream, const float* updates_dev,
                           const int32_t* indices_dev, int64_t e, int64_t d,
                           int32_t size, float* out_dev) {
  if (e >= (1LL << 24)) {
    // a destination could collect 2^24 or more updates: its f32 count would no longer
    // be its length - callers compose scatter_add as the reference does
    return Fail(EULER_GPU_EINVAL, "scatter_mean: e >= 2^24, compose scatter_add instead");
  }
  return ScatterImpl<2>((hipStream_t)stream, updates_dev, indices_dev, e, d, size, out_dev);
}

int euler_gpu_scatter_max(void* stream, const float* updates_dev,
                          const int32_t* indices_dev, int64_t e, int64_t d,
                          int32_t size, float* out_dev) {
  return ScatterImpl<1>((hipStream_t)stream, updates_dev, indices_dev, e, d,
                        size, out_dev);
}

int euler_gpu_gather_segment_reduce(void* stream, int32_t mode, const float* params_dev,
                                    const int32_t* gather_indices_dev,
                                    const int64_t* seg_ptr_dev, int64_t count, int64_t d,
                                    int32_t size, float* out_dev) {
  if (mode < 0 || mode > 2)
    return Fail(EULER_GPU_EINVAL, "gather_segment_reduce: mode is 0 add, 1 max, 2 mean");
  if (d < 0 || size < 0 || (!seg_ptr_dev && count < 0))
    return Fail(EULER_GPU_EINVAL, "gather_segment_reduce: bad shape");
  if (size == 0 || d == 0) return EULER_GPU_OK;
  if (!out_dev || !params_dev) return Fail(EULER_GPU_EINVAL, "gather_segment_reduce: null buffer");
  if (!seg_ptr_dev && mode == 2 && count >= (1LL << 24))
    return Fail(EULER_GPU_EINVAL, "gather_segment_reduce: mean needs segments shorter than 2^24");
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0)
    return SegmentReduceImpl<0>(st, params_dev, gather_indices_dev, seg_ptr_dev, count, d, size, out_dev);
  if (mode == 1)
    return SegmentReduceImpl<1>(st, params_dev, gather_indices_dev, seg_ptr_dev, count, d, size, out_dev);
  return SegmentReduceImpl<2>(st, params_dev, gather_indices_dev, seg_ptr_dev, count, d, size, out_dev);
}

int euler_gpu_gather_segment_reduce_ids(void* stream, int32_t mode, const float* params_dev,
                                        int64_t params_rows, const int64_t* gather_ids_dev,
                                        const int64_t* seg_ptr_dev, int64_t count, int64_t d,
                                        int32_t size, float* out_dev) {
  if (mode < 0 || mode > 2)
    return Fail(EULER_GPU_EINVAL, "gather_segment_reduce_ids: mode is 0 add, 1 max, 2 mean");
  if (d < 0 || size < 0 || (!seg_ptr_dev && count < 0))
    return Fail(EULER_GPU_EINVAL, "gather_segment_reduce_ids: bad shape");
  if (size == 0 || d == 0) return EULER_GPU_OK;
  if (!out_dev || !params_dev || !gather_ids_dev)
    return Fail(EULER_GPU_EINVAL, "gather_segment_reduce_ids: null buffer");
  if (!seg_ptr_dev && mode == 2 && count >= (1LL << 24))
    return Fail(EULER_GPU_EINVAL, "gather_segment_reduce_ids: mean needs segments shorter than 2^24");
  // the index is the low 32-bit word of the id: the table must be addressable by it, and an id
  // past the table (a neighbour that is not a node of this graph, a dangling edge) reads the
  // LAST row instead of memory outside the table
  if (params_rows < 0 || params_rows >= ((int64_t)1 << 31))
    return Fail(EULER_GPU_EINVAL, "gather_segment_reduce_ids: the table must have fewer than 2^31 rows");
  const uint32_t row_max = params_rows > 0 ? (uint32_t)(params_rows - 1) : 0xFFFFFFFFu;
  hipStream_t st = (hipStream_t)stream;
  const int32_t* lo = reinterpret_cast<const int32_t*>(gather_ids_dev);     // little endian: word 0 of every id
  if (mode == 0) return SegmentReduceImpl<0>(st, params_dev, lo, seg_ptr_dev, count, d, size, out_dev, 2, row_max);
  if (mode == 1) return SegmentReduceImpl<1>(st, params_dev, lo, seg_ptr_dev, count, d, size, out_dev, 2, row_max);
  return SegmentReduceImpl<2>(st, params_dev, lo, seg_ptr_dev, count, d, size, out_dev, 2, row_max);
}

int euler_gpu_gather_scatter(void* stream, int32_t mode, const float* params_dev,
                             const int32_t* gather_indices_dev,
                             const int32_t* scatter_indices_dev, int64_t e, int64_t d,
                             int32_t size, float* out_dev) {
  if (mode < 0 || mode > 2) return Fail(EULER_GPU_EINVAL, "gather_scatter: mode is 0 add, 1 max, 2 mean");
  if (e > 0 && !gather_indices_dev) return Fail(EULER_GPU_EINVAL, "gather_scatter: null buffer");
  if (mode == 2 && e >= (1LL << 24))
    return Fail(EULER_GPU_EINVAL, "gather_scatter: mean needs e < 2^24");
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0)
    return ScatterImpl<0>(st, params_dev, scatter_indices_dev, e, d, size, out_dev, gather_indices_dev);
  if (mode == 1)
    return ScatterImpl<1>(st, params_dev, scatter_indices_dev, e, d, size, out_dev, gather_indices_dev);
  return ScatterImpl<2>(st, params_dev, scatter_indices_dev, e, d, size, out_dev, gather_indices_dev);
}

int euler_gpu_neighbor_post_process(void* stream, int64_t n, int32_t* idx_dev,
                                    int64_t total, uint64_t* id_dev, float* w_dev,
                                    int32_t* t_dev, int32_t order_by, int32_t desc,
                                    int64_t limit, int64_t* total_host) {
  if (n < 0 || total < 0 || order_by < 0 || order_by > 2)
    return Fail(EULER_GPU_EINVAL, "neighbor_post_process: bad arguments");
  if (total_host) *total_host = total;
  if (n == 0 || total == 0 || (order_by == 0 && limit < 0)) return EULER_GPU_OK;
  if (!idx_dev || !id_dev || !w_dev || !t_dev)
    return Fail(EULER_GPU_EINVAL, "neighbor_post_process: null buffer");
  if (total >= (1LL << 31)) return Fail(EULER_GPU_EINVAL, "neighbor_post_process: total >= 2^31");
  hipStream_t st = (hipStream_t)stream;
  const int block = 256;
  // scratch: copies of the three value arrays, the permutation, lengths/offsets
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_id = 0, o_w = o_id + al((size_t)total * 8), o_t = o_w + al((size_t)total * 4);
  const size_t o_pin = o_t + al((size_t)total * 4), o_pout = o_pin + al((size_t)total * 4);
  const size_t o_key = o_pout + al((size_t)total * 4);
  const size_t o_len = o_key + al((size_t)total * 8), o_off = o_len + al((size_t)(n + 1) * 8);
  const size_t bytes = o_off + al((size_t)(n + 1) * 8);
  uint8_t* buf = nullptr;
  EG_HIP(hipMallocAsync((void**)&buf, bytes, st));
  uint64_t* c_id = (uint64_t*)(buf + o_id);
  float* c_w = (float*)(buf + o_w);
  int32_t* c_t = (int32_t*)(buf + o_t);
  int32_t* p_in = (int32_t*)(buf + o_pin);
  int32_t* p_out = (int32_t*)(buf + o_pout);
  int64_t* len = (int64_t*)(buf + o_len);
  int64_t* off = (int64_t*)(buf + o_off);
  EG_HIP(hipMemcpyAsync(c_id, id_dev, (size_t)total * 8, hipMemcpyDeviceToDevice, st));
  EG_HIP(hipMemcpyAsync(c_w, w_dev, (size_t)total * 4, hipMemcpyDeviceToDevice, st));
  EG_HIP(hipMemcpyAsync(c_t, t_dev, (size_t)total * 4, hipMemcpyDeviceToDevice, st));
  const int32_t* perm = nullptr;
  if (order_by != 0) {
    hipLaunchKernelGGL(IotaKernel, dim3((total + block - 1) / block), dim3(block), 0, st,
                       p_in, total);
    hipcub::CountingInputIterator<int32_t> row_it(0);
    hipcub::TransformInputIterator<int32_t, IdxEdgeLong,
                                   hipcub::CountingInputIterator<int32_t>>
        seg_b(row_it, IdxEdgeLong{idx_dev, 0}), seg_e(row_it, IdxEdgeLong{idx_dev, 1});
    size_t tmp_bytes = 0;
    void* tmp = nullptr;
    // the library pass is only needed when some row is longer than a wave ranks
    int32_t max_len = 0;
    {
      int32_t* d_max = (int32_t*)(buf + o_len);      // free until LimitLenKernel
      EG_HIP(hipMemsetAsync(d_max, 0, 4, st));
      hipLaunchKernelGGL(MaxRowLenKernel, dim3((n + block - 1) / block), dim3(block), 0,
                         st, idx_dev, n, d_max);
      EG_HIP(hipMemcpyAsync(&max_len, d_max, 4, hipMemcpyDeviceToHost, st));
      EG_HIP(hipStreamSynchronize(st));
    }
#define EG_SEGSORT(KEY_T, KEYS_IN, FN)                                               \
    { if (max_len > kWaveSortMax)                                                    \
    {                                                                                \
      KEY_T* keys_out = (KEY_T*)(buf + o_key);                                       \
      EG_HIP(hipcub::DeviceSegmentedRadixSort::FN(nullptr, tmp_bytes, KEYS_IN,       \
                                                  keys_out, p_in, p_out, (int)total, \
                                                  (int)n, seg_b, seg_e, 0,           \
                                                  (int)sizeof(KEY_T) * 8, st));      \
      EG_HIP(hipMallocAsync(&tmp, tmp_bytes + 16, st));                              \
      EG_HIP(hipcub::DeviceSegmentedRadixSort::FN(tmp, tmp_bytes, KEYS_IN, keys_out, \
                                                  p_in, p_out, (int)total, (int)n,   \
                                                  seg_b, seg_e, 0,                   \
                                                  (int)sizeof(KEY_T) * 8, st));      \
    } }
    if (order_by == 1) {
      if (desc) EG_SEGSORT(uint64_t, c_id, SortPairsDescending)
      else EG_SEGSORT(uint64_t, c_id, SortPairs)
    } else {
      if (desc) EG_SEGSORT(float, c_w, SortPairsDescending)
      else EG_SEGSORT(float, c_w, SortPairs)
    }
#undef EG_SEGSORT
    if (tmp) EG_HIP(hipFreeAsync(tmp, st));
    // short rows: ranked in a wave, written into the same permutation array
    {
      const int gridw = GridFor(n * 64, block);
      if (order_by == 1) {
        if (desc) hipLaunchKernelGGL((SmallRowRankKernel<uint64_t, true>), dim3(gridw),
                                     dim3(block), 0, st, idx_dev, n, c_id, p_out);
        else hipLaunchKernelGGL((SmallRowRankKernel<uint64_t, false>), dim3(gridw),
                                dim3(block), 0, st, idx_dev, n, c_id, p_out);
      } else {
        if (desc) hipLaunchKernelGGL((SmallRowRankKernel<float, true>), dim3(gridw),
                                     dim3(block), 0, st, idx_dev, n, c_w, p_out);
        else hipLaunchKernelGGL((SmallRowRankKernel<float, false>), dim3(gridw),
                                dim3(block), 0, st, idx_dev, n, c_w, p_out);
      }
    }
    perm = p_out;
  }
  hipLaunchKernelGGL(LimitLenKernel, dim3((n + block - 1) / block), dim3(block), 0, st,
                     idx_dev, n, limit, len);
  {
    const int rc = ExclusiveScanI64(st, len, off, n);
    if (rc != EULER_GPU_OK) return rc;
  }
  hipLaunchKernelGGL(RepackRowsKernel, dim3(GridFor(n * 64, block)), dim3(block), 0, st,
                     idx_dev, len, off, perm, n, c_id, c_w, c_t, id_dev, w_dev, t_dev);
  hipLaunchKernelGGL(LenOffToIdxKernel, dim3((n + block - 1) / block), dim3(block), 0,
                     st, len, off, n, idx_dev);
  EG_HIP(hipGetLastError());
  int32_t last[2];
  EG_HIP(hipMemcpyAsync(last, idx_dev + 2 * (n - 1), 8, hipMemcpyDeviceToHost, st));
  EG_HIP(hipStreamSynchronize(st));
  EG_HIP(hipFreeAsync(buf, st));
  if (total_host) *total_host = last[1];
  return EULER_GPU_OK;
}

int euler_gpu_neighbor_to_dense(void* stream, int64_t n, const int32_t* idx_dev,
                                const uint64_t* id_dev, const float* w_dev,
                                const int32_t* t_dev, int32_t k, int64_t default_node,
                                int64_t* out_id_dev, float* out_w_dev,
                                int32_t* out_t_dev) {
  if (n < 0 || k < 0) return Fail(EULER_GPU_EINVAL, "neighbor_to_dense: bad n/k");
  if (n == 0 || k == 0) return EULER_GPU_OK;
  if (!idx_dev || !out_id_dev || !out_w_dev || !out_t_dev)
    return Fail(EULER_GPU_EINVAL, "neighbor_to_dense: null buffer");
  const int block = 256;
  hipLaunchKernelGGL(NeighborToDenseKernel, dim3(GridFor(n * (int64_t)k, block)),
                     dim3(block), 0, (hipStream_t)stream, idx_dev, id_dev, w_dev, t_dev, n,
                     k, default_node, out_id_dev, out_w_dev, out_t_dev);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

thread_local int g_feature_vec4 = 1;   // euler_gpu_set_tuning key 8

// 16-byte lanes: dim % 4 == 0, fixed-stride table whose rows and slots start on
// 16-byte boundaries (feat_uniform, stride % 4 == 0, slot begin % 4 == 0).
__global__ __launch_bounds__(256) void DenseFeatureVec4Kernel(
    const GraphView g, const uint64_t* __restrict__ nodes, int64_t n, int32_t fid,
    int32_t dv /* dim / 4 */, float4* __restrict__ out) {
  const int32_t pre = fid == 0 ? 0 : g.feat_idx[fid - 1];
  const int32_t len = g.feat_idx[fid] - pre;
  const uint32_t total = (uint32_t)(n * dv);
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < total; s += stride) {
    const uint32_t j = s / (uint32_t)dv;
    const int32_t c = (int32_t)(s - j * (uint32_t)dv) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t row = FindRow(g, nodes[j]);
    if (row >= 0 && c < len) {
      const float* src = g.feat_val + row * g.feat_stride + pre + c;
      if (c + 3 < len) {
        v = *reinterpret_cast<const float4*>(src);
      } else {
        v.x = src[0];
        if (c + 1 < len) v.y = src[1];
        if (c + 2 < len) v.z = src[2];
      }
    }
    out[s] = v;
  }
}

int euler_gpu_get_dense_feature(const euler_gpu_graph* g, void* stream,
                                const uint64_t* nodes_dev, int64_t n, int32_t fid,
                                int32_t dim, float* out_dev) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "get_dense_feature: null graph");
  if (n < 0 || dim < 0) return Fail(EULER_GPU_EINVAL, "get_dense_feature: bad n/dim");
  if (n == 0 || dim == 0) return EULER_GPU_OK;
  if (!nodes_dev || !out_dev)
    return Fail(EULER_GPU_EINVAL, "get_dense_feature: null buffer");
  const int block = 256;
  const GraphView& v = g->view;
  const bool vec4 = g_feature_vec4 != 0 && v.feat_uniform && fid >= 0 && fid < v.n_float && dim % 4 == 0 &&
                    v.feat_stride % 4 == 0 && g->feat_slot_aligned &&
                    ((uintptr_t)out_dev % 16 == 0) && n * (int64_t)(dim / 4) < 0xffffffffLL;
  if (vec4) {
    hipLaunchKernelGGL(DenseFeatureVec4Kernel, dim3(GridFor(n * (int64_t)(dim / 4), block)),
                       dim3(block), 0, (hipStream_t)stream, v, nodes_dev, n, fid, dim / 4,
                       reinterpret_cast<float4*>(out_dev));
  } else {
    hipLaunchKernelGGL(DenseFeatureKernel, dim3(GridFor(n * (int64_t)dim, block)),
                       dim3(block), 0, (hipStream_t)stream, v, nodes_dev, n, fid, dim,
                       out_dev);
  }
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

int euler_gpu_id_unique(void* stream, const uint64_t* ids_dev, int64_t n,
                        uint64_t* unique_dev, int32_t* gather_idx_dev,
                        int64_t* n_unique_host) {
  if (n < 0) return Fail(EULER_GPU_EINVAL, "id_unique: bad n");
  if (n == 0) { if (n_unique_host) *n_unique_host = 0; return EULER_GPU_OK; }
  if (n >= (1LL << 31)) return Fail(EULER_GPU_EINVAL, "id_unique: n >= 2^31");
  if (!ids_dev || !unique_dev || !gather_idx_dev)
    return Fail(EULER_GPU_EINVAL, "id_unique: null buffer");
  hipStream_t st = (hipStream_t)stream;
  uint64_t cap = 64;
  while (cap < (uint64_t)n * 2) cap <<= 1;
  const size_t slots = cap + 1;
  const size_t bytes = slots * (8 + 4 + 4) + (size_t)n * (4 + 8 + 8) + 256;
  uint8_t* buf = nullptr;
  EG_HIP(hipMallocAsync((void**)&buf, bytes, st));
  UniqueTable t;
  t.keys = (unsigned long long*)buf;
  int64_t* is_first = (int64_t*)(t.keys + slots);
  int64_t* rank = is_first + n;
  t.minpos = (uint32_t*)(rank + n);
  t.rank = (int32_t*)(t.minpos + slots);
  uint32_t* slot_of = (uint32_t*)(t.rank + slots);
  t.mask = cap - 1;
  EG_HIP(hipMemsetAsync(t.keys, 0xFF, slots * 8, st));
  EG_HIP(hipMemsetAsync(t.minpos, 0xFF, slots * 4, st));
  const int block = 256;
  const dim3 grid((unsigned)((n + block - 1) / block));
  hipLaunchKernelGGL(UniqueInsertKernel, grid, dim3(block), 0, st, ids_dev, n, t,
                     slot_of);
  hipLaunchKernelGGL(UniqueFlagKernel, grid, dim3(block), 0, st, n, t, slot_of,
                     is_first);
  int rc = ExclusiveScanI64(st, is_first, rank, n);
  if (rc != EULER_GPU_OK) return rc;
  hipLaunchKernelGGL(UniqueEmitKernel, grid, dim3(block), 0, st, ids_dev, n, t,
                     slot_of, is_first, rank, unique_dev);
  hipLaunchKernelGGL(UniqueGatherIdxKernel, grid, dim3(block), 0, st, n, t, slot_of,
                     gather_idx_dev);
  EG_HIP(hipGetLastError());
  int64_t tail[2];
  EG_HIP(hipMemcpyAsync(&tail[0], rank + (n - 1), 8, hipMemcpyDeviceToHost, st));
  EG_HIP(hipMemcpyAsync(&tail[1], is_first + (n - 1), 8, hipMemcpyDeviceToHost, st));
  EG_HIP(hipStreamSynchronize(st));
  EG_HIP(hipFreeAsync(buf, st));
  if (n_unique_host) *n_unique_host = tail[0] + tail[1];
  return EULER_GPU_OK;
}

int euler_gpu_idx_gather(void* stream, const int32_t* idx_dev,
                         const int32_t* gather_idx_dev, int64_t n,
                         int32_t* out_idx_dev, int64_t* total_host) {
  if (n < 0) return Fail(EULER_GPU_EINVAL, "idx_gather: bad n");
  if (n == 0) { if (total_host) *total_host = 0; return EULER_GPU_OK; }
  hipStream_t st = (hipStream_t)stream;
  int64_t* len = nullptr;
  EG_HIP(hipMallocAsync((void**)&len, (2 * n + 2) * 8, st));
  int64_t* off = len + n + 1;
  const int block = 256;
  const dim3 grid((unsigned)((n + block - 1) / block));
  hipLaunchKernelGGL(SegLenKernel, grid, dim3(block), 0, st, idx_dev,
                     gather_idx_dev, n, len);
  int rc = ExclusiveScanI64(st, len, off, n);
  if (rc != EULER_GPU_OK) return rc;
  hipLaunchKernelGGL(EmitIdxKernel, grid, dim3(block), 0, st, len, off, n,
                     out_idx_dev);
  int32_t last[2];
  EG_HIP(hipMemcpyAsync(last, out_idx_dev + 2 * (n - 1), 8, hipMemcpyDeviceToHost, st));
  EG_HIP(hipStreamSynchronize(st));
  EG_HIP(hipFreeAsync(len, st));
  if (total_host) *total_host = last[1];
  return EULER_GPU_OK;
}

int euler_gpu_data_gather(void* stream, const void* data_dev, int32_t elem_size,
                          const int32_t* idx_dev, const int32_t* gather_idx_dev,
                          const int32_t* out_idx_dev, int64_t n, void* out_dev) {
  if (n < 0) return Fail(EULER_GPU_EINVAL, "data_gather: bad n");
  if (n == 0) return EULER_GPU_OK;
  hipStream_t st = (hipStream_t)stream;
  const int block = 256;
  const int grid = GridFor(n * 16, block);
  if (elem_size == 8) {
    hipLaunchKernelGGL(DataGatherKernel<uint64_t>, dim3(grid), dim3(block), 0, st,
                       (const uint64_t*)data_dev, idx_dev, gather_idx_dev,
                       out_idx_dev, n, (uint64_t*)out_dev);
  } else if (elem_size == 4) {
    hipLaunchKernelGGL(DataGatherKernel<uint32_t>, dim3(grid), dim3(block), 0, st,
                       (const uint32_t*)data_dev, idx_dev, gather_idx_dev,
                       out_idx_dev, n, (uint32_t*)out_dev);
  } else if (elem_size == 1) {
    hipLaunchKernelGGL(DataGatherKernel<uint8_t>, dim3(grid), dim3(block), 0, st,
                       (const uint8_t*)data_dev, idx_dev, gather_idx_dev,
                       out_idx_dev, n, (uint8_t*)out_dev);
  } else {
    return Fail(EULER_GPU_EINVAL, "data_gather: elem_size must be 1, 4 or 8");
  }
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

int euler_gpu_id_split(void* stream, const uint64_t* ids_dev, int64_t n,
                       int32_t partitions, int32_t shards,
                       int64_t* shard_off_host, uint64_t* shard_ids_dev,
                       int32_t* merge_idx_dev) {
  if (n < 0 || partitions <= 0 || shards <= 0 || shards > kMaxShards ||
      !shard_off_host)
    return Fail(EULER_GPU_EINVAL, "id_split: bad arguments (shards <= 64)");
  for (int s = 0; s <= shards; ++s) shard_off_host[s] = 0;
  if (n == 0) return EULER_GPU_OK;
  if (n >= (1LL << 31)) return Fail(EULER_GPU_EINVAL, "id_split: n >= 2^31");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n_blocks = (n + kSplitBlock - 1) / kSplitBlock;
  const int64_t cells = n_blocks * shards;
  int64_t* hist = nullptr;
  EG_HIP(hipMallocAsync((void**)&hist, (2 * cells + 2) * 8, st));
  int64_t* off = hist + cells + 1;
  hipLaunchKernelGGL(SplitHistKernel, dim3((unsigned)n_blocks), dim3(kSplitBlock), 0,
                     st, ids_dev, n, partitions, shards, hist);
  int rc = ExclusiveScanI64(st, hist, off, cells);
  if (rc != EULER_GPU_OK) return rc;
  hipLaunchKernelGGL(SplitScatterKernel, dim3((unsigned)n_blocks), dim3(kSplitBlock),
                     0, st, ids_dev, n, partitions, shards, off, shard_ids_dev,
                     merge_idx_dev);
  EG_HIP(hipGetLastError());
  // shard s starts at the scanned offset of its first block cell
  std::vector<int64_t> starts(shards);
  for (int s = 0; s < shards; ++s)
    EG_HIP(hipMemcpyAsync(&starts[s], off + (int64_t)s * n_blocks, 8,
                          hipMemcpyDeviceToHost, st));
  EG_HIP(hipStreamSynchronize(st));
  EG_HIP(hipFreeAsync(hist, st));
  for (int s = 0; s < shards; ++s) shard_off_host[s] = starts[s];
  shard_off_host[shards] = n;
  return EULER_GPU_OK;
}

// ------------------------------------------------------------------------
// InflateIdx (tf_euler/kernels/inflate_idx_op.cc:34-66; op tf_euler/ops/util_ops.cc): for an
// index vector whose values are exactly 0 .. U-1, out[i] = (entries with a smaller value) +
// (entries with the same value before i) - the place of entry i after a STABLE sort by value.
// The reference counts, prefix-sums and hands the places out in input order on one thread;
// here: a stable radix sort of (value, position), then position perm[j] gets place j.  The
// pass that writes the places also counts the value changes of the sorted keys, so the
// reference's "expect input idx in [0,unique_cnt)" is one comparison on the host afterwards.
// ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void InflatePlaceKernel(const int32_t* __restrict__ keys,
                                                          const uint32_t* __restrict__ perm,
                                                          int64_t n, int32_t* __restrict__ out,
                                                          int32_t* __restrict__ stat) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool edge = false;
  if (j < n) {
    const int32_t k = keys[j];
    out[perm[j]] = (int32_t)j;
    edge = j > 0 && keys[j - 1] != k;
    if (j == 0) stat[1] = k;            // smallest value
    if (j == n - 1) stat[2] = k;        // largest value
  }
  const uint64_t b = __ballot(edge);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(stat, __popcll(b));
}

int euler_gpu_inflate_idx(void* stream, const int32_t* idx_dev, int64_t n, int32_t* out_dev) {
  if (n < 0 || (n > 0 && (!idx_dev || !out_dev)))
    return Fail(EULER_GPU_EINVAL, "inflate_idx: bad arguments");
  if (n == 0) return EULER_GPU_OK;
  if (n >= (1LL << 31)) return Fail(EULER_GPU_EINVAL, "inflate_idx: n >= 2^31");
  hipStream_t st = (hipStream_t)stream;
  const size_t un = (size_t)n;
  size_t tmp_bytes = 0;
  EG_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, idx_dev, (int32_t*)nullptr,
                                            (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n,
                                            0, 32, st));
  const size_t head = (un * 12 + 16 + 255) & ~(size_t)255;     // sorted keys, positions in / out, stat
  char* buf = nullptr;
  EG_HIP(hipMallocAsync((void**)&buf, head + tmp_bytes + 16, st));
  int32_t* keys_out = (int32_t*)buf;
  uint32_t* vals_in = (uint32_t*)(keys_out + un);
  uint32_t* vals_out = vals_in + un;
  int32_t* stat = (int32_t*)(vals_out + un);
  const unsigned blocks = (unsigned)((n + 255) / 256);
  int rc = EULER_GPU_OK;
  int32_t h[3] = {0, 0, 0};
  do {
    if (hipMemsetAsync(stat, 0, 16, st) != hipSuccess) { rc = EULER_GPU_EHIP; break; }
    hipLaunchKernelGGL(IotaKernel, dim3(blocks), dim3(256), 0, st, vals_in, n);
    if (hipcub::DeviceRadixSort::SortPairs(buf + head, tmp_bytes, idx_dev, keys_out, vals_in,
                                           vals_out, (int)n, 0, 32, st) != hipSuccess) {
      rc = EULER_GPU_EHIP; break;
    }
    hipLaunchKernelGGL(InflatePlaceKernel, dim3(blocks), dim3(256), 0, st, keys_out, vals_out, n,
                       out_dev, stat);
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(h, stat, 12, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      rc = EULER_GPU_EHIP;
  } while (false);
  (void)hipFreeAsync(buf, st);
  if (rc != EULER_GPU_OK) return Fail(rc, "inflate_idx: HIP call failed");
  // h[0] + 1 distinct values, all of them >= h[1] and <= h[2]
  if (h[1] < 0 || (int64_t)h[2] > (int64_t)h[0])
    return Fail(EULER_GPU_EINVAL, "inflate_idx: expect input idx in [0,unique_cnt)");
  return EULER_GPU_OK;
}

// ------------------------------------------------------------------------
// Front end of a multi-GPU hop: the distinct ids of a batch, bucketed by owner
// (ID_UNIQUE then ID_SPLIT, parser/compiler.cc:76-90 + core/kernels/
// id_split_op.cc), plus for every input position the index of its id in the
// bucketed array - so the rows that come back from the shards (in the order the
// ids were sent) expand straight to positions, without a merge pass.
// Duplicates are found without atomics (see sample_kernels.hip, DedupArgs):
// owner[hash(id)] = position by plain stores, the survivor of a slot represents
// every position whose id equals its own; a position that lost its slot to a
// DIFFERENT id simply represents itself (its duplicates are then not merged:
// more rows on the wire, same result).  Two rounds (two tables of 4x the batch,
// two hashes) leave ~0.01 % of the distinct ids in that state.
// ------------------------------------------------------------------------
struct DedupIdsArgs {
  const uint64_t* ids;
  const uint8_t* root_mask;   // optional: a set byte makes its group sample as id 0
  int32_t root_group;
  int32_t pad;
  int64_t n;
  uint32_t mask;
  uint32_t* owner;     // [mask + 1]
  uint32_t* owner2;    // [mask + 1] second round (stale entries are checked by id)
  // a level in SLAB layout (the walk enqueued without host waits, csrc/sharded.cc): `n` positions =
  // slabs of in_stride words, word 0 of a slab its header, entries 1 .. in_lens[slab] the level's
  // nodes, the rest padding.  null: every position below n holds an id.
  const uint32_t* in_lens;
  uint32_t in_stride;
};

__device__ __forceinline__ uint64_t DedupIdAt(const DedupIdsArgs& a, int64_t i) {
  if (a.root_mask != nullptr && a.root_mask[i / a.root_group]) return 0;
  return a.ids[i];
}

// does position i hold an id of the batch?
__device__ __forceinline__ bool DedupLive(const DedupIdsArgs& a, int64_t i) {
  if (i >= a.n) return false;
  if (a.in_lens == nullptr) return true;
  const uint32_t p = (uint32_t)i / a.in_stride, j = (uint32_t)i - p * a.in_stride;
  return j != 0u && j <= a.in_lens[p];
}

__device__ __forceinline__ uint32_t DedupSlot2(uint64_t id, uint32_t mask) {
  return (uint32_t)(Mix64(id ^ 0x9E3779B97F4A7C15ULL) >> 20) & mask;
}

// Representative of position i: the survivor of its first-round slot if that
// is the same id, else the survivor of its second-round slot (a second table,
// a second hash, written only by first-round losers), else i itself.
__device__ __forceinline__ uint32_t DedupRep(const DedupIdsArgs& a, uint32_t i) {
  const uint64_t id = DedupIdAt(a, i);
  const uint32_t o = a.owner[(uint32_t)Mix64(id) & a.mask];
  if (DedupIdAt(a, o) == id) return o;
  const uint32_t o2 = a.owner2[DedupSlot2(id, a.mask)];
  // (a stale entry may name a position that holds no id in THIS call - padding of a slab level)
  return (DedupLive(a, o2) && DedupIdAt(a, o2) == id) ? o2 : i;
}

__global__ __launch_bounds__(256) void DedupIdsMarkKernel(const DedupIdsArgs a) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride)
    if (DedupLive(a, i)) a.owner[(uint32_t)Mix64(DedupIdAt(a, i)) & a.mask] = (uint32_t)i;
}

// second round: positions whose slot went to a different id try another slot
__global__ __launch_bounds__(256) void DedupIdsMark2Kernel(const DedupIdsArgs a) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
    if (!DedupLive(a, i)) continue;
    const uint64_t id = DedupIdAt(a, i);
    const uint32_t o = a.owner[(uint32_t)Mix64(id) & a.mask];
    if (DedupIdAt(a, o) != id) a.owner2[DedupSlot2(id, a.mask)] = (uint32_t)i;
  }
}

// ---- v3 front end (round 5): representatives -> places, THREE launches, no scan, the bucket
// sizes written straight into pinned host memory.
//   mark      owner[slot(id)] = position (plain stores; block 0 also clears the shard totals)
//   rep/hist  a workgroup takes a chunk of kFrontChunk positions: representative of every
//             position; a representative takes its rank within (chunk, owner shard) from an
//             LDS counter; the chunk adds its per-shard counts to the shard totals with ONE
//             global atomic per shard and keeps what it got back - its base in the bucket
//   place     bucket start of a shard = the totals before it (<= 64 numbers, summed by every
//             workgroup itself); place of a representative = start + chunk base + rank;
//             pos[i] = place of i's representative; block 0 writes the starts into the
//             caller's pinned buffer and then the call's sequence number - the host polls it.
// The order inside a bucket is the arrival order of the chunks' atomics: arbitrary, like the
// choice of representatives (plain-store races) - results never depend on either, the rows
// come back in the order asked and pos[] is their map.  (v2: per-(shard, block) histogram ->
// hipcub scan -> scatter -> compose -> 8-byte copy to the host: six launches and a copy; v1:
// 0.61 ms for the metric's hop 2 on 8 shards, most of it host time.)
constexpr int kFrontChunk = 2048;
struct FrontArgs {
  DedupIdsArgs d;            // ids / mask / hash tables (hash mode)
  uint32_t* dense_owner;     // dense mode: [dense_limit + 1], slot = id (or the limit)
  uint64_t dense_limit;
  uint32_t* rep;             // [n] representative of every position
  uint32_t* place;           // [n] at representatives: owner shard << 16 | rank within (chunk, shard)
  uint32_t* chunk_base;      // [shards, n_chunks] base of the chunk's representatives in the bucket
  uint32_t* total;           // [kMaxShards] representatives per shard
  int64_t n_chunks;
  int32_t partitions, shards;
  // buckets in SLAB layout (FrontSlabKernel): shard s's ids at out_stride * s + 1 .., their number
  // in word 0 of the slab and in out_lens[s] - fixed-size messages, nothing for the host to wait
  // for.  0: buckets packed one behind the other, their starts to the host (stage).
  uint32_t out_stride;
  uint32_t* out_lens;        // [shards]
  // (FrontSlabKernel: out_lens doubles as the shard totals - zero when the kernel starts)
};

__device__ __forceinline__ uint32_t DenseSlot(const FrontArgs& a, uint64_t id) {
  return id < a.dense_limit ? (uint32_t)id : (uint32_t)a.dense_limit;
}

// dense ids: owner[id] = position by plain stores (see sample_kernels.hip,
// DedupMarkKernel); every id >= the limit is "no such node" and shares a slot -
// whichever owner answers for its representative answers the default row
__global__ __launch_bounds__(256) void FrontMarkDenseKernel(const FrontArgs a) {
  if (blockIdx.x == 0 && threadIdx.x < kMaxShards) a.total[threadIdx.x] = 0u;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.d.n; i += stride)
    if (DedupLive(a.d, i)) a.dense_owner[DenseSlot(a, DedupIdAt(a.d, i))] = (uint32_t)i;
}
// (Round 5, measured and removed: electing one LEADER lane per distinct id of a wave - a loop of
// ballots and shuffles - so that only leaders touch the table.  The second hop's positions are
// ~8 distinct ids per wave, so the table traffic drops 8x; the kernels are latency-bound, not
// traffic-bound: the sharded step went 0.400 -> 0.413 ms with one minibatch in flight, the
// sharded walk - whose levels are mostly distinct - 2.55 -> 3.12 ms.)

__global__ void FrontClearTotalsKernel(uint32_t* total) { total[threadIdx.x] = 0u; }

__global__ __launch_bounds__(256) void FrontRepHistKernel(const FrontArgs a) {
  const bool DENSE = a.dense_owner != nullptr;
  __shared__ uint32_t hist[kMaxShards];
  if (threadIdx.x < kMaxShards) hist[threadIdx.x] = 0u;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kFrontChunk;
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t lt = lane == 0u ? 0ull : (~0ull >> (64u - lane));
  constexpr int kPer = kFrontChunk / 256;
  // dense ids: the chunk's ids, then its table entries - every load of a kind in flight together
  // (taken position by position the chunk is kPer dependent pairs of round trips long)
  uint64_t idv[kPer];
  uint32_t repv[kPer];
  bool live[kPer];
#pragma unroll
  for (int32_t k = 0; k < kPer; ++k) live[k] = DedupLive(a.d, base + (int64_t)k * 256 + threadIdx.x);
  if (DENSE) {
#pragma unroll
    for (int32_t k = 0; k < kPer; ++k) {
      const int64_t i = base + (int64_t)k * 256 + threadIdx.x;
      idv[k] = live[k] ? DedupIdAt(a.d, i) : 0ull;
    }
#pragma unroll
    for (int32_t k = 0; k < kPer; ++k)
      repv[k] = live[k] ? a.dense_owner[DenseSlot(a, idv[k])] : 0xFFFFFFFFu;
  }
#pragma unroll
  for (int32_t k = 0; k < kPer; ++k) {
    const int64_t i = base + (int64_t)k * 256 + threadIdx.x;
    bool is_rep = false;
    uint32_t own = 0xFFFFFFFFu;
    if (live[k]) {
      const uint64_t id = DENSE ? idv[k] : DedupIdAt(a.d, i);
      const uint32_t r = DENSE ? repv[k] : DedupRep(a.d, (uint32_t)i);
      a.rep[i] = r;
      is_rep = r == (uint32_t)i;
      if (is_rep) own = (uint32_t)OwnerOf(id, a.partitions, a.shards);
    }
    // ranks within (chunk, owner shard): ONE LDS atomic per shard present in the wave, the lanes'
    // ranks from the ballot.  (A lane-per-representative atomicAdd is 64 serialized operations on
    // one LDS address whenever a wave's representatives share an owner - always with one rank,
    // every eighth with eight: it made this kernel 21 us long whatever the level held.)
    uint32_t rank = 0;
    for (uint32_t s = 0; s < (uint32_t)a.shards; ++s) {          // (wave-uniform trip count)
      const bool mine = is_rep && own == s;
      const uint64_t same = __ballot(mine);
      if (same == 0ull) continue;
      const uint32_t first = (uint32_t)__ffsll((unsigned long long)same) - 1u;
      uint32_t got = 0;
      if (lane == first) got = atomicAdd(&hist[s], (uint32_t)__popcll(same));
      got = (uint32_t)__shfl((int)got, (int)first);
      if (mine) rank = got + (uint32_t)__popcll(same & lt);
    }
    if (is_rep) a.place[i] = own << 16 | rank;
  }
  __syncthreads();
  if ((int)threadIdx.x < a.shards)
    a.chunk_base[(int64_t)threadIdx.x * a.n_chunks + blockIdx.x] =
        hist[threadIdx.x] != 0u ? atomicAdd(&a.total[threadIdx.x], hist[threadIdx.x]) : 0u;
}

__global__ __launch_bounds__(256) void FrontPlaceKernel(const FrontArgs a, uint64_t* __restrict__ shard_ids,
                                                        int32_t* __restrict__ pos_out,
                                                        volatile int64_t* stage, const int64_t seq) {
  __shared__ uint32_t bstart[kMaxShards + 1];
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (int32_t s = 0; s < a.shards; ++s) { bstart[s] = acc; acc += a.total[s]; }
    bstart[a.shards] = acc;
    if (blockIdx.x == 0) {           // the host waits for exactly this: sizes first, then the echo
      for (int32_t s = 0; s <= a.shards; ++s) stage[s] = (int64_t)bstart[s];
      __threadfence_system();
      stage[kMaxShards + 1] = seq;
    }
  }
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.d.n; i += stride) {
    if (!DedupLive(a.d, i)) continue;
    const uint32_t r = a.rep[i];
    const uint32_t w = a.place[r];
    const uint32_t own = w >> 16;
    const uint32_t q = bstart[own] + a.chunk_base[(int64_t)own * a.n_chunks + r / kFrontChunk] + (w & 0xFFFFu);
    pos_out[i] = (int32_t)q;
    if (r == (uint32_t)i) shard_ids[q] = DedupIdAt(a.d, i);
  }
}

// ---- the front end as ONE kernel (slab layout; the walk enqueued without host waits) --------
// With the buckets in slabs a representative's place is known the moment its chunk has taken its
// base from the shard's total - bucket starts are fixed, no second pass over the totals:
//   rep     MODE 0 (hashed ids): DedupRep over the tables the two mark kernels wrote;
//           MODE 2 (id-indexed table): the entry FrontMarkDenseKernel left - the default;
//           MODE 3: every position represents itself (no mark pass, no table: the late levels of
//           a walk, which hold few new duplicates - the buckets are only split by owner);
//           (measured and removed, round 6: a single pass that claims the table slot by
//           compare-and-swap, entries tagged with an 8-bit epoch and checked by content - no mark
//           pass; 1M x 40 walk 1.96-2.30 ms against 1.72-1.83: a level repeats its hubs thousands
//           of times and their swaps queue up at one address, profiles/r6_walk_enqueued_ab.txt)
//   place   representatives rank within (chunk, owner) as in FrontRepHistKernel, the chunk adds
//           its counts to the totals (= out_lens) and places them at slab start + 1 + base + rank;
//           a position that is NOT its id's representative leaves ~representative in pos[] - the
//           walk's path kernel follows the one indirection (the representative's chunk may not
//           have run yet);
//   sizes   out_lens holds them when the kernel has ended; FrontSlabHeadersKernel copies them into
//           the slabs' headers when the slabs travel (a "last workgroup writes them" tail - a
//           fence + one counter every workgroup bumps - cost 30 us a step: 489 atomics on one word).
__global__ void FrontSlabHeadersKernel(uint64_t* slabs, uint32_t stride, const uint32_t* lens, int32_t shards) {
  if ((int)threadIdx.x < shards) slabs[(int64_t)threadIdx.x * stride] = (uint64_t)lens[threadIdx.x];
}

extern "C++" {
template <int MODE>
__global__ __launch_bounds__(256) void FrontSlabKernel(const FrontArgs a, uint64_t* __restrict__ shard_ids,
                                                       int32_t* __restrict__ pos_out) {
  __shared__ uint32_t hist[kMaxShards];
  if (threadIdx.x < kMaxShards) hist[threadIdx.x] = 0u;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kFrontChunk;
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t lt = lane == 0u ? 0ull : (~0ull >> (64u - lane));
  constexpr int kPer = kFrontChunk / 256;
  uint64_t idv[kPer];
  uint32_t repv[kPer], wv[kPer];
  bool live[kPer];
#pragma unroll
  for (int32_t k = 0; k < kPer; ++k) live[k] = DedupLive(a.d, base + (int64_t)k * 256 + threadIdx.x);
#pragma unroll
  for (int32_t k = 0; k < kPer; ++k)
    idv[k] = live[k] ? DedupIdAt(a.d, base + (int64_t)k * 256 + threadIdx.x) : 0ull;
  if (MODE == 2) {
#pragma unroll
    for (int32_t k = 0; k < kPer; ++k) repv[k] = live[k] ? a.dense_owner[DenseSlot(a, idv[k])] : 0u;
  } else if (MODE == 3) {
#pragma unroll
    for (int32_t k = 0; k < kPer; ++k) repv[k] = (uint32_t)(base + (int64_t)k * 256 + threadIdx.x);
  } else {
#pragma unroll
    for (int32_t k = 0; k < kPer; ++k)
      repv[k] = live[k] ? DedupRep(a.d, (uint32_t)(base + (int64_t)k * 256 + threadIdx.x)) : 0u;
  }
#pragma unroll
  for (int32_t k = 0; k < kPer; ++k) {
    const int64_t i = base + (int64_t)k * 256 + threadIdx.x;
    const bool is_rep = live[k] && repv[k] == (uint32_t)i;
    const uint32_t own = is_rep ? (uint32_t)OwnerOf(idv[k], a.partitions, a.shards) : 0xFFFFFFFFu;
    uint32_t rank = 0;
    for (uint32_t s = 0; s < (uint32_t)a.shards; ++s) {          // (wave-uniform trip count)
      const bool mine = is_rep && own == s;
      const uint64_t same = __ballot(mine);
      if (same == 0ull) continue;
      const uint32_t first = (uint32_t)__ffsll((unsigned long long)same) - 1u;
      uint32_t got = 0;
      if (lane == first) got = atomicAdd(&hist[s], (uint32_t)__popcll(same));
      got = (uint32_t)__shfl((int)got, (int)first);
      if (mine) rank = got + (uint32_t)__popcll(same & lt);
    }
    wv[k] = is_rep ? (own << 24 | rank) : 0xFFFFFFFFu;           // (rank < 2 048, own < 64)
  }
  __syncthreads();
  if ((int)threadIdx.x < a.shards) {
    const uint32_t h = hist[threadIdx.x];
    hist[threadIdx.x] = h != 0u ? atomicAdd(&a.out_lens[threadIdx.x], h) : 0u;     // the chunk's base
  }
  __syncthreads();
#pragma unroll
  for (int32_t k = 0; k < kPer; ++k) {
    if (!live[k]) continue;
    const int64_t i = base + (int64_t)k * 256 + threadIdx.x;
    if (wv[k] != 0xFFFFFFFFu) {
      const uint32_t own = wv[k] >> 24;
      const uint32_t q = own * a.out_stride + 1u + hist[own] + (wv[k] & 0xFFFFFFu);
      pos_out[i] = (int32_t)q;
      shard_ids[q] = idv[k];
    } else {
      pos_out[i] = (int32_t)~repv[k];
    }
  }
}
}  // extern "C++"

// Grow-only device scratch per (device, stream) for the front end: it runs
// twice per hop with very different sizes, and alternating stream-ordered
// allocations of 1 MB and 60 MB cost ~0.2 ms per call in the allocator.  The
// entry is locked for the whole call (which ends with a stream sync, so the
// scratch is idle again when the lock drops).
struct StreamScratch {
  std::mutex mu;
  void* ptr = nullptr;
  size_t bytes = 0;
};
static StreamScratch* ScratchEntry(hipStream_t stream) {
  static std::mutex map_mu;
  static std::map<std::pair<int, void*>, StreamScratch*> entries;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(map_mu);
  auto& e = entries[std::make_pair(dev, (void*)stream)];
  if (e == nullptr) e = new StreamScratch();
  return e;
}
// caller holds e->mu
static int ScratchReserve(StreamScratch* e, hipStream_t stream, size_t bytes, void** out) {
  if (e->bytes < bytes) {
    if (e->ptr != nullptr) {
      EG_HIP(hipStreamSynchronize(stream));
      EG_HIP(hipFree(e->ptr));
      e->ptr = nullptr; e->bytes = 0;
    }
    const size_t want = bytes + bytes / 4;
    hipError_t err = hipMalloc(&e->ptr, want);
    if (err != hipSuccess) {
      e->ptr = nullptr;
      return Fail(EULER_GPU_ENOMEM, std::string("dedup_split scratch: ") + hipGetErrorString(err));
    }
    e->bytes = want;
  }
  *out = e->ptr;
  return EULER_GPU_OK;
}

// One front-end call in flight: the bucket sizes travel through a small pinned
// buffer (a pageable device-to-host copy costs ~40 us; this one is written by the
// copy engine directly) and an event marks their arrival, so that a caller can
// enqueue the front ends of several minibatches before it waits for the first.
int euler_gpu_front_create(euler_gpu_front** out) {
  if (!out) return Fail(EULER_GPU_EINVAL, "front_create: null");
  euler_gpu_front* f = new (std::nothrow) euler_gpu_front();
  if (!f) return Fail(EULER_GPU_ENOMEM, "front_create: out of memory");
  if (hipHostMalloc((void**)&f->stage, (kMaxShards + 2) * sizeof(int64_t),
                    hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostGetDevicePointer((void**)&f->stage_dev, f->stage, 0) != hipSuccess ||
      hipEventCreateWithFlags(&f->done, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    if (f->stage) (void)hipHostFree(f->stage);
    delete f;
    return Fail(EULER_GPU_ENOMEM, "front_create: pinned buffer / event");
  }
  f->stage[kMaxShards + 1] = 0;
  *out = f;
  return EULER_GPU_OK;
}

void euler_gpu_front_destroy(euler_gpu_front* f) {
  if (!f) return;
  if (f->pending) (void)hipEventSynchronize(f->done);
  (void)hipEventDestroy(f->done);
  (void)hipHostFree(f->stage);
  delete f;
}

// The front end's three launches (dense ids; five with hashing) on `st`.  in_lens / in_stride: the
// batch is a level in slab layout (DedupIdsArgs); out_stride / out_lens: the buckets leave in slab
// layout and their sizes stay on the device (FrontArgs), else their starts go to `stage` + `seq`.
static int FrontEnqueue(hipStream_t st, const uint64_t* ids_dev, int64_t n, const uint8_t* root_mask_dev,
                        int32_t root_group, const uint32_t* in_lens_dev, uint32_t in_stride,
                        int32_t partitions, int32_t shards, uint32_t* dense_owner_dev, int64_t dense_limit,
                        uint64_t* shard_ids_dev, uint32_t out_stride, uint32_t* out_lens_dev,
                        bool write_headers, bool dedup, int32_t* pos_dev, volatile int64_t* stage_dev,
                        int64_t seq) {
  const bool dense = dense_owner_dev != nullptr;
  uint64_t cap = 1024;
  while (!dense && cap < (uint64_t)n * 4) cap <<= 1;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const int64_t n_chunks = (n + kFrontChunk - 1) / kFrontChunk;
  const size_t o_owner = 0, o_owner2 = o_owner + (dense ? 0 : al(cap * 4));
  const bool slabs = out_stride != 0u;      // (FrontSlabKernel keeps representatives and places in registers)
  const size_t o_rep = o_owner2 + (dense ? 0 : al(cap * 4));
  const size_t o_place = o_rep + (slabs ? 0 : al((size_t)n * 4));
  const size_t o_base = o_place + (slabs ? 0 : al((size_t)n * 4));
  const size_t o_total = o_base + (slabs ? 0 : al((size_t)n_chunks * shards * 4));
  const size_t bytes = o_total + al(kMaxShards * 4);
  StreamScratch* scratch = ScratchEntry(st);
  std::lock_guard<std::mutex> scratch_lk(scratch->mu);
  uint8_t* buf = nullptr;
  {
    void* p = nullptr;
    const int src = ScratchReserve(scratch, st, bytes, &p);
    if (src != EULER_GPU_OK) return src;
    buf = (uint8_t*)p;
  }
  FrontArgs a{};
  a.d.ids = ids_dev; a.d.n = n; a.d.mask = (uint32_t)(cap - 1);
  a.d.root_mask = root_mask_dev; a.d.root_group = root_group > 0 ? root_group : 1;
  a.d.owner = (uint32_t*)(buf + o_owner);
  a.d.owner2 = (uint32_t*)(buf + o_owner2);
  a.d.in_lens = in_lens_dev; a.d.in_stride = in_stride;
  a.dense_owner = dense_owner_dev;
  a.dense_limit = (uint64_t)dense_limit;
  a.rep = (uint32_t*)(buf + o_rep);
  a.place = (uint32_t*)(buf + o_place);
  a.chunk_base = (uint32_t*)(buf + o_base);
  a.total = (uint32_t*)(buf + o_total);
  a.n_chunks = n_chunks;
  a.partitions = partitions; a.shards = shards;
  a.out_stride = out_stride; a.out_lens = out_lens_dev;
  const int block = 256;
  const int grid = GridFor(n, block);
  if (out_stride != 0u) {
    // slabs: one kernel (two mark kernels before it when the ids are hashed)
    if (!dedup) {
      hipLaunchKernelGGL(FrontSlabKernel<3>, dim3((unsigned)n_chunks), dim3(256), 0, st, a, shard_ids_dev, pos_dev);
    } else if (dense) {
      hipLaunchKernelGGL(FrontMarkDenseKernel, dim3(grid), dim3(block), 0, st, a);
      hipLaunchKernelGGL(FrontSlabKernel<2>, dim3((unsigned)n_chunks), dim3(256), 0, st, a, shard_ids_dev, pos_dev);
    } else {
      hipLaunchKernelGGL(DedupIdsMarkKernel, dim3(grid), dim3(block), 0, st, a.d);
      hipLaunchKernelGGL(DedupIdsMark2Kernel, dim3(grid), dim3(block), 0, st, a.d);
      hipLaunchKernelGGL(FrontSlabKernel<0>, dim3((unsigned)n_chunks), dim3(256), 0, st, a, shard_ids_dev, pos_dev);
    }
    if (write_headers)
      hipLaunchKernelGGL(FrontSlabHeadersKernel, dim3(1), dim3(kMaxShards), 0, st, shard_ids_dev, out_stride,
                         out_lens_dev, shards);
    if (hipGetLastError() != hipSuccess) return Fail(EULER_GPU_EHIP, "front_slabs: launch failed");
    return EULER_GPU_OK;
  }
  if (dense) {
    hipLaunchKernelGGL(FrontMarkDenseKernel, dim3(grid), dim3(block), 0, st, a);
  } else {
    hipLaunchKernelGGL(FrontClearTotalsKernel, dim3(1), dim3(kMaxShards), 0, st, a.total);
    hipLaunchKernelGGL(DedupIdsMarkKernel, dim3(grid), dim3(block), 0, st, a.d);
    hipLaunchKernelGGL(DedupIdsMark2Kernel, dim3(grid), dim3(block), 0, st, a.d);
  }
  hipLaunchKernelGGL(FrontRepHistKernel, dim3((unsigned)n_chunks), dim3(256), 0, st, a);
  hipLaunchKernelGGL(FrontPlaceKernel, dim3(grid), dim3(block), 0, st, a, shard_ids_dev, pos_dev, stage_dev, seq);
  if (hipGetLastError() != hipSuccess) return Fail(EULER_GPU_EHIP, "dedup_split: launch failed");
  // the scratch stays in use until the stream has run these kernels; whoever
  // takes the lock next enqueues on the same stream, i.e. after them
  return EULER_GPU_OK;
}

int euler_gpu_dedup_split_begin(euler_gpu_front* f, void* stream, const uint64_t* ids_dev,
                                int64_t n, const uint8_t* root_mask_dev, int32_t root_group,
                                int32_t partitions, int32_t shards,
                                uint32_t* dense_owner_dev, int64_t dense_limit,
                                uint64_t* shard_ids_dev, int32_t* pos_dev) {
  if (!f || n < 0 || partitions <= 0 || shards <= 0 || shards > kMaxShards)
    return Fail(EULER_GPU_EINVAL, "dedup_split: bad arguments (shards <= 64)");
  if (f->pending) return Fail(EULER_GPU_EINVAL, "dedup_split: the handle has a call in flight");
  f->shards = shards;
  for (int s = 0; s <= shards; ++s) f->stage[s] = 0;
  if (n == 0) return EULER_GPU_OK;
  if (n >= (1LL << 30)) return Fail(EULER_GPU_EINVAL, "dedup_split: n >= 2^30");
  if (!ids_dev || !shard_ids_dev || !pos_dev)
    return Fail(EULER_GPU_EINVAL, "dedup_split: null buffer");
  if (dense_owner_dev != nullptr && (dense_limit <= 0 || dense_limit >= (1LL << 32) - 1))
    return Fail(EULER_GPU_EINVAL, "dedup_split: dense_limit out of range");
  hipStream_t st = (hipStream_t)stream;
  f->seq += 1;
  int rc = FrontEnqueue(st, ids_dev, n, root_mask_dev, root_group, nullptr, 0, partitions, shards,
                        dense_owner_dev, dense_limit, shard_ids_dev, 0, nullptr, false, true, pos_dev,
                        (volatile int64_t*)f->stage_dev, f->seq);
  if (rc == EULER_GPU_OK) {
    const hipError_t e = hipEventRecord(f->done, st);
    if (e != hipSuccess) rc = Fail(EULER_GPU_EHIP, std::string("dedup_split: ") + hipGetErrorString(e));
    else f->pending = 1;
  }
  return rc;
}

}  // extern "C"

namespace euler_gpu {
__global__ void FrontSlabsEmptyKernel(uint64_t* slabs, uint32_t stride, uint32_t* lens, int32_t shards) {
  if ((int)threadIdx.x < shards) { slabs[(int64_t)threadIdx.x * stride] = 0ull; lens[threadIdx.x] = 0u; }
}

// The front end of a hop whose batch and buckets live in SLAB layout, sizes on the device: no
// host wait (csrc/sharded.cc, the enqueued walk).  ids_dev: n_pos positions, every one an id
// (in_lens_dev null: level 0) or slabs of in_stride words with entries 1 .. in_lens_dev[slab].
// Out: slab s of out_slabs_dev (out_stride words: header = the bucket's size, then its ids),
// out_lens_dev[s] the same size, pos_dev[i] = the word of position i's id among the slabs, or
// ~r when position r (>= 0 there) holds the same id.  out_lens_dev [shards] must be ZERO when the
// kernels start (the caller clears a walk's worth of them at once).  write_headers = false
// leaves the headers alone: a lone rank's owner pass reads out_lens_dev itself.  dedup = false:
// every position is sent (equal ids are asked for once each).
int FrontSlabs(hipStream_t st, const uint64_t* ids_dev, int64_t n_pos, const uint32_t* in_lens_dev,
               uint32_t in_stride, int32_t partitions, int32_t shards, uint32_t* dense_owner_dev,
               int64_t dense_limit, uint64_t* out_slabs_dev, uint32_t out_stride, uint32_t* out_lens_dev,
               bool write_headers, bool dedup, int32_t* pos_dev) {
  if (n_pos < 0 || n_pos >= (1LL << 30) || partitions <= 0 || shards <= 0 || shards > kMaxShards ||
      out_stride == 0u || (int64_t)out_stride * shards >= (1LL << 30) || !out_slabs_dev || !out_lens_dev ||
      (in_lens_dev != nullptr && in_stride == 0u))
    return Fail(EULER_GPU_EINVAL, "front_slabs: bad arguments");
  if (n_pos == 0) {
    hipLaunchKernelGGL(FrontSlabsEmptyKernel, dim3(1), dim3(kMaxShards), 0, st, out_slabs_dev, out_stride,
                       out_lens_dev, shards);
    EG_HIP(hipGetLastError());
    return EULER_GPU_OK;
  }
  if (!ids_dev || !pos_dev) return Fail(EULER_GPU_EINVAL, "front_slabs: null buffer");
  if (dense_owner_dev != nullptr && (dense_limit <= 0 || dense_limit >= (1LL << 32) - 1))
    return Fail(EULER_GPU_EINVAL, "front_slabs: dense_limit out of range");
  return FrontEnqueue(st, ids_dev, n_pos, nullptr, 1, in_lens_dev, in_stride, partitions, shards,
                      dense_owner_dev, dense_limit, out_slabs_dev, out_stride, out_lens_dev, write_headers, dedup,
                      pos_dev,
                      nullptr, 0);
}
}  // namespace euler_gpu

extern "C" {

int euler_gpu_dedup_split_end(euler_gpu_front* f, int64_t* shard_off_host) {
  if (!f || !shard_off_host) return Fail(EULER_GPU_EINVAL, "dedup_split_end: null");
  if (f->pending) {
    // the place kernel's first workgroup writes the bucket starts into the pinned buffer and
    // then echoes the call's sequence number: the host polls that word (a few microseconds
    // after the kernel STARTS, where an event fires after it ends and costs a wake-up); the
    // event is the fallback for a runtime that would not make the write visible in time
    volatile int64_t* echo = (volatile int64_t*)f->stage + (kMaxShards + 1);
    bool seen = false;
    for (int64_t spin = 0; spin < (int64_t)1 << 22 && !seen; ++spin) {
      seen = *echo == f->seq;
      if (!seen && (spin & 63) == 63 && hipEventQuery(f->done) == hipSuccess) { seen = *echo == f->seq; break; }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    if (!seen) {
      const hipError_t e = hipEventSynchronize(f->done);
      if (e != hipSuccess) {
        f->pending = 0;
        return Fail(EULER_GPU_EHIP, std::string("dedup_split_end: ") + hipGetErrorString(e));
      }
      if (*echo != f->seq) { f->pending = 0; return Fail(EULER_GPU_EHIP, "dedup_split_end: the bucket sizes did not arrive"); }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    f->pending = 0;
  }
  for (int s = 0; s <= f->shards; ++s) shard_off_host[s] = ((volatile int64_t*)f->stage)[s];
  return EULER_GPU_OK;
}

int euler_gpu_dedup_split(void* stream, const uint64_t* ids_dev, int64_t n,
                          const uint8_t* root_mask_dev, int32_t root_group,
                          int32_t partitions, int32_t shards, uint32_t* dense_owner_dev,
                          int64_t dense_limit, int64_t* shard_off_host,
                          uint64_t* shard_ids_dev, int32_t* pos_dev) {
  if (!shard_off_host) return Fail(EULER_GPU_EINVAL, "dedup_split: null shard_off_host");
  static thread_local euler_gpu_front* f = nullptr;      // one per host thread, kept
  if (f == nullptr) {
    const int rc = euler_gpu_front_create(&f);
    if (rc != EULER_GPU_OK) return rc;
  }
  int rc = euler_gpu_dedup_split_begin(f, stream, ids_dev, n, root_mask_dev, root_group,
                                       partitions, shards, dense_owner_dev, dense_limit,
                                       shard_ids_dev, pos_dev);
  if (rc == EULER_GPU_OK) rc = euler_gpu_dedup_split_end(f, shard_off_host);
  return rc;
}

int euler_gpu_merge_rows(void* stream, const void* in_dev,
                         const int32_t* merge_idx_dev, int64_t n_rows,
                         int64_t row_bytes, void* out_dev) {
  if (n_rows < 0 || row_bytes <= 0 || row_bytes % 4 != 0)
    return Fail(EULER_GPU_EINVAL, "merge_rows: row_bytes must be a multiple of 4");
  if (n_rows == 0) return EULER_GPU_OK;
  const int block = 256;
  const int64_t words = row_bytes / 4;
  hipLaunchKernelGGL(MergeRowsKernel, dim3(GridFor(n_rows * words, block)),
                     dim3(block), 0, (hipStream_t)stream, (const uint32_t*)in_dev,
                     merge_idx_dev, n_rows, words, (uint32_t*)out_dev);
  EG_HIP(hipGetLastError());
  return EULER_GPU_OK;
}

int euler_gpu_sample_node_split(uint64_t seed, uint32_t call_id, int32_t count,
                                const float* shard_weight_host, int32_t shards,
                                int32_t* split_cnt_host) {
  // core/kernels/sample_node_split_op.cc:57-85 with the type weights already
  // summed per shard; remainder draws use RNG domain SPLIT.
  if (count < 0 || shards <= 0 || !shard_weight_host || !split_cnt_host)
    return Fail(EULER_GPU_EINVAL, "sample_node_split: bad arguments");
  std::vector<int32_t> nz;
  int32_t remain = count;
  const float sw1 = shard_weight_host[shards];
  if (std::fabs(sw1 - 0.0) < 0.0000001)
    return Fail(EULER_GPU_EEMPTY, "sample_node_split: node type sum weight is zero");
  for (int32_t i = 0; i < shards; ++i) {
    const float sw0 = shard_weight_host[i];
    split_cnt_host[i] = (int32_t)std::floor(count * sw0 / sw1);
    if (sw0 > 0) nz.push_back(i);
    remain -= split_cnt_host[i];
  }
  for (uint64_t d = 0; remain > 0; ++d, --remain) {
    const double u = RngDraw(seed, call_id, kDomainSplit, 0, d);
    split_cnt_host[nz[(size_t)std::floor(u * nz.size())]] += 1;
  }
  return EULER_GPU_OK;
}

}  // extern "C"
