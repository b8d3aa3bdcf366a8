// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access
// shapes of the sampling kernels: every kernel below moves a
// KNOWN number of bytes in a KNOWN request shape over a 4 GiB working set (far
// beyond the 32 MB of L2 and the 256 MB infinity cache), one launch each:
//   rd16_in_sector   random 16-byte reads, 64-byte aligned: ONE 64-B sector each
//   rd16_straddle    random 16-byte reads at offset 56 of a 128-byte line: both
//                    64-B sectors of ONE 128-B line each
//   rd4_random       random 4-byte reads (one sector each)
//   rd128_line       random whole 128-byte lines (a quad of lanes reads 8 x 16 B)
//   rd_stream        wide coalesced streaming read (16 B per lane) - the guide's
//                    reference pattern (FETCH_SIZE = 1/2 of the bytes)
//   wr4_random       random 4-byte stores (the next-hop owner marks)
//   wr16_stream      coalesced 16-byte stores (the sampler's outputs)
// Usage: rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- tools/ubench_fetch
//   (then WRITE_SIZE, TCC_EA0_RDREQ_sum ... in separate passes); the program
//   prints the bytes each kernel asked for, tools/pmc_round2.py joins the two.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_fetch.hip -o tools/ubench_fetch
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr uint64_t kBytes = 4ULL << 30;
constexpr int kGrid = 256 * 8, kBlock = 256, kPerLane = 64;

__device__ __forceinline__ uint64_t Mix(uint64_t z) {
  z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL; z ^= z >> 27; z *= 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}

__global__ void rd16_in_sector(const uint8_t* base, uint32_t* sink) {
  const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (int i = 0; i < kPerLane; ++i) {
    const uint64_t line = Mix(tid * kPerLane + i) % (kBytes / 128);
    const uint4 q = *reinterpret_cast<const uint4*>(base + line * 128 + 64 * (i & 1));
    acc += q.x ^ q.y ^ q.z ^ q.w;
  }
  if (acc == 0x12345678u) sink[tid] = acc;
}

__global__ void rd16_straddle(const uint8_t* base, uint32_t* sink) {
  const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (int i = 0; i < kPerLane; ++i) {
    const uint64_t line = Mix(tid * kPerLane + i + (1ULL << 40)) % (kBytes / 128);
    // 8 bytes in the first 64-byte sector, 8 in the second (8-byte aligned halves)
    const uint2 a = *reinterpret_cast<const uint2*>(base + line * 128 + 56);
    const uint2 b = *reinterpret_cast<const uint2*>(base + line * 128 + 64);
    acc += a.x ^ a.y ^ b.x ^ b.y;
  }
  if (acc == 0x12345678u) sink[tid] = acc;
}

// the same straddling bytes as ONE unaligned 16-byte load (what the pivot windows do)
typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));
__global__ void rd16_straddle_one_load(const uint8_t* base, uint32_t* sink) {
  const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (int i = 0; i < kPerLane; ++i) {
    const uint64_t line = Mix(tid * kPerLane + i + (2ULL << 40)) % (kBytes / 128);
    const u32x4u q = *reinterpret_cast<const u32x4u*>(base + line * 128 + 56);
    acc += q.x ^ q.y ^ q.z ^ q.w;
  }
  if (acc == 0x12345678u) sink[tid] = acc;
}

__global__ void rd4_random(const uint8_t* base, uint32_t* sink) {
  const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (int i = 0; i < kPerLane; ++i) {
    const uint64_t w = Mix(tid * kPerLane + i + (3ULL << 40)) % (kBytes / 4);
    acc += *reinterpret_cast<const uint32_t*>(base + w * 4);
  }
  if (acc == 0x12345678u) sink[tid] = acc;
}

__global__ void rd128_line(const uint8_t* base, uint32_t* sink) {
  const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (int i = 0; i < kPerLane / 8; ++i) {
    const uint64_t line = Mix(tid * kPerLane + i + (4ULL << 40)) % (kBytes / 128);
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const uint4 q = *reinterpret_cast<const uint4*>(base + line * 128 + 16 * x);
      acc += q.x ^ q.y ^ q.z ^ q.w;
    }
  }
  if (acc == 0x12345678u) sink[tid] = acc;
}

__global__ void rd_stream(const uint8_t* base, uint32_t* sink) {
  const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  const uint64_t n = (uint64_t)kGrid * kBlock;
  uint32_t acc = 0;
  for (int i = 0; i < kPerLane; ++i) {
    const uint4 q = *reinterpret_cast<const uint4*>(base + ((uint64_t)i * n + tid) * 16);
    acc += q.x ^ q.y ^ q.z ^ q.w;
  }
  if (acc == 0x12345678u) sink[tid] = acc;
}

__global__ void wr4_random(uint8_t* base) {
  const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  for (int i = 0; i < kPerLane; ++i) {
    const uint64_t w = Mix(tid * kPerLane + i + (5ULL << 40)) % (kBytes / 4);
    *reinterpret_cast<uint32_t*>(base + w * 4) = (uint32_t)tid;
  }
}

__global__ void wr16_stream(uint8_t* base) {
  const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  const uint64_t n = (uint64_t)kGrid * kBlock;
  for (int i = 0; i < kPerLane; ++i)
    *reinterpret_cast<uint4*>(base + ((uint64_t)i * n + tid) * 16) = make_uint4(tid, i, 0, 0);
}

__global__ void Fill(uint32_t* p, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x) p[i] = (uint32_t)(i * 2654435761u);
}

int main() {
  uint8_t* buf; uint32_t* sink;
  CK(hipMalloc(&buf, kBytes));
  CK(hipMalloc(&sink, (size_t)kGrid * kBlock * 4));
  hipLaunchKernelGGL(Fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)buf, kBytes / 4);
  CK(hipDeviceSynchronize());
  const double lanes = (double)kGrid * kBlock;
#define RUN(K, ...) do { hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); \
    CK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(K, dim3(kGrid), dim3(kBlock), 0, 0, __VA_ARGS__); \
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); \
    printf("%-24s %8.3f ms", #K, ms); } while (0)
  RUN(rd16_in_sector, buf, sink);  printf("  accesses %.0f  asked_bytes %.0f  sectors64 %.0f  lines128 %.0f\n", lanes * kPerLane, lanes * kPerLane * 16, lanes * kPerLane, lanes * kPerLane);
  RUN(rd16_straddle, buf, sink);   printf("  accesses %.0f  asked_bytes %.0f  sectors64 %.0f  lines128 %.0f\n", lanes * kPerLane, lanes * kPerLane * 16, 2 * lanes * kPerLane, lanes * kPerLane);
  RUN(rd16_straddle_one_load, buf, sink); printf("  accesses %.0f  asked_bytes %.0f  sectors64 %.0f  lines128 %.0f\n", lanes * kPerLane, lanes * kPerLane * 16, 2 * lanes * kPerLane, lanes * kPerLane);
  RUN(rd4_random, buf, sink);      printf("  accesses %.0f  asked_bytes %.0f  sectors64 %.0f  lines128 %.0f\n", lanes * kPerLane, lanes * kPerLane * 4, lanes * kPerLane, lanes * kPerLane);
  RUN(rd128_line, buf, sink);      printf("  accesses %.0f  asked_bytes %.0f  sectors64 %.0f  lines128 %.0f\n", lanes * kPerLane / 8, lanes * kPerLane * 16, 2 * lanes * kPerLane / 8, lanes * kPerLane / 8);
  RUN(rd_stream, buf, sink);       printf("  accesses %.0f  asked_bytes %.0f  sectors64 %.0f  lines128 %.0f\n", lanes * kPerLane, lanes * kPerLane * 16, lanes * kPerLane / 4, lanes * kPerLane / 8);
  RUN(wr4_random, buf);            printf("  accesses %.0f  asked_bytes %.0f  sectors64 %.0f  lines128 %.0f\n", lanes * kPerLane, lanes * kPerLane * 4, lanes * kPerLane, lanes * kPerLane);
  RUN(wr16_stream, buf);           printf("  accesses %.0f  asked_bytes %.0f  sectors64 %.0f  lines128 %.0f\n", lanes * kPerLane, lanes * kPerLane * 16, lanes * kPerLane / 4, lanes * kPerLane / 8);
  CK(hipDeviceSynchronize());
  return 0;
}
