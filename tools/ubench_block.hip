// Micro-benchmark (round 6): what does fetching one random 128-byte index block cost on gfx950
// depending on HOW the lanes of a wave ask for it?  The sampler reads a block's keys as three
// 16-byte loads per lane plus a dependent 8-byte id load - four requests for one line per lane.
//   MODE 0: lane-private line, ONE dwordx4 (floor: one request per line)
//   MODE 1: lane-private line, 3 x dwordx4 + a dependent 8-byte load (what WbSamplePair does)
//   MODE 2: lane-private line, 3 x dwordx4 only
//   MODE 3: 8 lanes share a line, each its own 16 bytes (one coalesced request per line,
//           64 lines per 8 instructions), values exchanged with DPP-style shuffles
//   MODE 4: 4 lanes share a line: three take the key chunks, the fourth the id chunk the pick
//           names (dependent), 16 lines per instruction
//   MODE 5: 3 lanes share a line (key chunks only, 21 lines per instruction - groups do not
//           align with the quads of lanes), the owner's dependent id load private (1 lane of 3)
//   MODE 6: MODE 5 with the key chunks sent straight to LDS (global_load_lds_dwordx4) and read
//           back from there - the staging of fanout_plain.h's cooperative build
//   MODE 7: 4 lanes share a line, all four chunks (64 bytes) through LDS-DMA, dependent id private
//   MODE 8: lane-private line, ONE dwordx4 (a predictor: quantized keys) + a dependent ALIGNED
//           dwordx4 of the same line
//   MODE 9: as 8, the dependent dwordx4 at a 4-byte-aligned offset 16 + 12 i (a {sum, id, sum}
//           window that may straddle a 64-byte boundary)
// Independent loads per wave-step = UNROLL lines per lane (MODEs 0-2) so that the memory-level
// parallelism per wave matches the sampler's (a pair of draws per lane).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_block.hip -o gpurun_out/ubench_block
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t Mix(uint64_t z) {
  z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL;
  z ^= z >> 27; z *= 0x94d049bb133111ebULL;
  z ^= z >> 31;
  return z;
}

template <int MODE>
__global__ __launch_bounds__(256) void Fetch(const uint32_t* __restrict__ base, uint64_t n_lines,
                                             int iters, uint32_t* __restrict__ sink) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63;
  uint32_t acc = 0;
  extern __shared__ uint4 stage[];
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t grp3 = (lane * 21846u) >> 16;       // lane / 3
  uint64_t x = Mix(MODE == 3 ? tid >> 3 : (MODE == 4 || MODE == 7) ? tid >> 2
                   : (MODE == 5 || MODE == 6) ? (uint64_t)(tid >> 6) * 32 + grp3 : tid);
  for (int i = 0; i < iters; ++i) {
    if (MODE == 8 || MODE == 9) {
      const uint64_t l0 = (x >> 8) % n_lines, l1 = (Mix(x) >> 8) % n_lines;
      const uint32_t* p0 = base + l0 * 32;
      const uint32_t* p1 = base + l1 * 32;
      const uint4 a = *reinterpret_cast<const uint4*>(p0), b = *reinterpret_cast<const uint4*>(p1);
      uint32_t v = a.x ^ a.w ^ b.y ^ b.z;
      uint4 c, d;
      if (MODE == 8) {
        c = reinterpret_cast<const uint4*>(p0)[1 + (v % 7u)];
        d = reinterpret_cast<const uint4*>(p1)[1 + ((v >> 8) % 7u)];
      } else {
        const uint32_t* q0 = p0 + 4 + 3 * (v % 9u);
        const uint32_t* q1 = p1 + 4 + 3 * ((v >> 8) % 9u);
        c.x = q0[0]; c.y = q0[1]; c.z = q0[2]; c.w = q0[3];
        d.x = q1[0]; d.y = q1[1]; d.z = q1[2]; d.w = q1[3];
      }
      v ^= c.x ^ c.w ^ d.y ^ d.z;
      acc += v;
      x = Mix(x + v + 0x9E3779B97F4A7C15ULL);
    } else if (MODE <= 2) {
      // two independent lines per lane and step (a pair of draws)
      const uint64_t l0 = (x >> 8) % n_lines, l1 = (Mix(x) >> 8) % n_lines;
      const uint4* p0 = reinterpret_cast<const uint4*>(base + l0 * 32);
      const uint4* p1 = reinterpret_cast<const uint4*>(base + l1 * 32);
      uint32_t v;
      if (MODE == 0) {
        const uint4 a = p0[0], b = p1[0];
        v = a.x ^ b.y;
      } else {
        const uint4 a0 = p0[0], a1 = p0[1], a2 = p0[2];
        const uint4 b0 = p1[0], b1 = p1[1], b2 = p1[2];
        v = a0.x ^ a1.y ^ a2.z ^ b0.x ^ b1.y ^ b2.z;
        if (MODE == 1) {
          const uint64_t* i0 = reinterpret_cast<const uint64_t*>(p0) + 6 + (v % 10u);
          const uint64_t* i1 = reinterpret_cast<const uint64_t*>(p1) + 6 + ((v >> 8) % 10u);
          v ^= (uint32_t)(*i0) ^ (uint32_t)(*i1);
        }
      }
      acc += v;
      x = Mix(x + v + 0x9E3779B97F4A7C15ULL);
    } else if (MODE == 3) {
      // 8 lanes per line; two lines per group and step keep 2 x 8 = 16 lines per wave-step...
      // the wave makes 4 such steps where MODE 1 makes one (64 lanes x 2 lines = 128 lines)
      const uint64_t l0 = (x >> 8) % n_lines, l1 = (Mix(x) >> 8) % n_lines;
      const uint4 a = reinterpret_cast<const uint4*>(base + l0 * 32)[lane & 7];
      const uint4 b = reinterpret_cast<const uint4*>(base + l1 * 32)[lane & 7];
      uint32_t v = a.x ^ a.w ^ b.y ^ b.z;
      // reduce over the group (three xor-shuffles), as a count of keys would be
      v ^= __shfl_xor(v, 1); v ^= __shfl_xor(v, 2); v ^= __shfl_xor(v, 4);
      acc += v;
      x = Mix(x + v + 0x9E3779B97F4A7C15ULL);
    } else if (MODE == 5 || MODE == 6 || MODE == 7) {
      const uint64_t l0 = (x >> 8) % n_lines, l1 = (Mix(x) >> 8) % n_lines;
      const uint32_t per = MODE == 7 ? 4u : 3u;
      const uint32_t g = MODE == 7 ? lane >> 2 : grp3, sub = lane - per * g;
      const bool act = MODE == 7 || lane < 63u;
      const uint4* p0 = reinterpret_cast<const uint4*>(base + l0 * 32);
      const uint4* p1 = reinterpret_cast<const uint4*>(base + l1 * 32);
      uint4 a = make_uint4(0, 0, 0, 0), b = a;
      if (MODE == 5) {
        if (act) { a = p0[sub]; b = p1[sub]; }
      } else {
        uint4* st0 = stage + wave * 128;               // [2][64] chunks
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t*)st0);
        const uint32_t d1 = d0 + 1024u;
        if (act) {
          unsigned keep;
          const uint4* s0 = p0 + sub; const uint4* s1 = p1 + sub;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(s0), "s"(d0) : "memory");
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(s1), "s"(d1) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        a = st0[lane]; b = st0[64 + lane];
      }
      uint32_t v = a.x ^ a.w ^ b.y ^ b.z;
      // the group's owner (sub == 0) gathers the other chunks' words and asks for the id
      const uint32_t v1 = __shfl(v, lane + 1), v2 = __shfl(v, lane + 2);
      v ^= v1 ^ v2;
      if (sub == 0 && act) {
        const uint64_t* i0 = reinterpret_cast<const uint64_t*>(p0) + 6 + (v % 10u);
        const uint64_t* i1 = reinterpret_cast<const uint64_t*>(p1) + 6 + ((v >> 8) % 10u);
        v ^= (uint32_t)(*i0) ^ (uint32_t)(*i1);
      }
      v = __shfl(v, lane - sub);
      acc += v;
      x = Mix(x + v + 0x9E3779B97F4A7C15ULL);
    } else {
      const uint64_t l0 = (x >> 8) % n_lines, l1 = (Mix(x) >> 8) % n_lines;
      const uint32_t sub = lane & 3;
      const uint4* p0 = reinterpret_cast<const uint4*>(base + l0 * 32);
      const uint4* p1 = reinterpret_cast<const uint4*>(base + l1 * 32);
      uint4 a = make_uint4(0, 0, 0, 0), b = a;
      if (sub < 3) { a = p0[sub]; b = p1[sub]; }
      uint32_t v = a.x ^ a.w ^ b.y ^ b.z;
      v ^= __shfl_xor(v, 1); v ^= __shfl_xor(v, 2);
      if (sub == 3) {     // the id chunk the pick names
        a = p0[3 + (v % 5u)]; b = p1[3 + ((v >> 8) % 5u)];
        v ^= a.x ^ b.x;
      }
      v = __shfl(v, lane | 3);
      acc += v;
      x = Mix(x + v + 0x9E3779B97F4A7C15ULL);
    }
  }
  if (acc == 0x12345678u) sink[tid] = acc;
}

__global__ void Fill(uint32_t* p, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x)
    p[i] = (uint32_t)Mix(i);
}

template <int MODE>
static void Run(const uint32_t* buf, uint64_t n_lines, const char* name, uint32_t* sink, int wpc) {
  // wpc waves per CU resident: 256 CUs x wpc / 4 workgroups of 256 threads
  const int block = 256, grid = 256 * wpc / 4, iters = MODE == 3 ? 256 : (MODE == 4 || MODE == 7) ? 128 : MODE >= 8 ? 32 : MODE >= 5 ? 96 : 32;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(Fetch<MODE>, dim3(grid), dim3(block), 8192, 0, buf, n_lines, iters, sink);
  CK(hipEventRecord(e0, 0));
  const int reps = 5;
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(Fetch<MODE>, dim3(grid), dim3(block), 8192, 0, buf, n_lines, iters, sink);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double lane_steps = (double)grid * block * iters;
  const double lines = lane_steps * 2.0 / (MODE == 3 ? 8.0 : (MODE == 4 || MODE == 7) ? 4.0 : MODE >= 8 ? 1.0 : MODE >= 5 ? 64.0 / 21.0 : 1.0);
  printf("%-44s waves/CU %2d  %8.3f ms  %7.2f G lines/s\n", name, wpc, ms, lines / ms / 1e6);
}

int main(int argc, char** argv) {
  // optional argument: GiB of index to draw the random lines from (default 16)
  const uint64_t bytes = (uint64_t)(argc > 1 ? atoll(argv[1]) : 16) << 30;
  uint32_t* buf; uint32_t* sink;
  // UB_CONTIG=1: physically contiguous device memory (hipDeviceMallocContiguous) - do larger
  // page-table fragments widen the translation reach?
  if (getenv("UB_CONTIG") && atoi(getenv("UB_CONTIG"))) {
    CK(hipExtMallocWithFlags((void**)&buf, bytes, hipDeviceMallocContiguous));
    printf("# buffer: hipExtMallocWithFlags(hipDeviceMallocContiguous)\n");
  } else {
    CK(hipMalloc(&buf, bytes));
  }
  CK(hipMalloc(&sink, 256 * 32 * 64 * 4));
  hipLaunchKernelGGL(Fill, dim3(4096), dim3(256), 0, 0, buf, bytes / 4);
  CK(hipDeviceSynchronize());
  const uint64_t nl = bytes / 128;
  for (int wpc : {8, 16, 24, 32}) {
    Run<0>(buf, nl, "private line, 1 x dwordx4", sink, wpc);
    Run<2>(buf, nl, "private line, 3 x dwordx4", sink, wpc);
    Run<1>(buf, nl, "private line, 3 x dwordx4 + dependent id", sink, wpc);
    Run<3>(buf, nl, "8 lanes share a line (coalesced)", sink, wpc);
    Run<4>(buf, nl, "4 lanes share a line, id chunk dependent", sink, wpc);
    Run<5>(buf, nl, "3 lanes share a line + private id", sink, wpc);
    Run<6>(buf, nl, "3 lanes share a line via LDS-DMA + private id", sink, wpc);
    Run<7>(buf, nl, "4 lanes, 64 B via LDS-DMA + private id", sink, wpc);
    Run<8>(buf, nl, "private line, 1 x dwordx4 + dependent x4", sink, wpc);
    Run<9>(buf, nl, "private line, 1 x dwordx4 + dep. x4 at 16+12i", sink, wpc);
    printf("\n");
  }
  return 0;
}
