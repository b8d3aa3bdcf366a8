// Wave-level helpers for the reference's sequential f32 running sums (gfx950).
#ifndef EULER_AMD_CSRC_WAVE_SUMS_H_
#define EULER_AMD_CSRC_WAVE_SUMS_H_

#include <hip/hip_runtime.h>

namespace euler_gpu {

__device__ __forceinline__ float ReadLaneF(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// Inclusive integer sum over the wave (mod 2^32): four row_shr steps inside each
// row of 16 lanes, then the three row totals added to the rows above them.
__device__ __forceinline__ uint32_t WaveInclusiveAdd(uint32_t x, int lane) {
  int32_t v = (int32_t)x;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112 /* row_shr:2 */, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114 /* row_shr:4 */, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118 /* row_shr:8 */, 0xf, 0xf, true);
  const int32_t t0 = __builtin_amdgcn_readlane(v, 15);
  const int32_t t1 = __builtin_amdgcn_readlane(v, 31);
  const int32_t t2 = __builtin_amdgcn_readlane(v, 47);
  return (uint32_t)(v + (lane >= 16 ? t0 : 0) + (lane >= 32 ? t1 : 0) + (lane >= 48 ? t2 : 0));
}

// carry + d[0] + ... + d[63] (lane k holds d[k]; lanes past the end hold 0) as the
// reference's sequential f32 adds would give it, WITHOUT adding one by one when the
// sums stay inside the binade of the carry and no add is a rounding tie: then
// fl(carry + d) = (m + n) * ulp with n = d / ulp rounded to nearest, and the chain is an
// integer sum over the mantissa (n2v_kernels.h has the full account and the
// four-entries-per-lane form; tools/n2v_binade_model.py restates it in numpy).  A lane
// the integer sum cannot pass (the sum leaves the binade there, a tie, a negative
// entry, a zero carry) is done by one real add and the lanes after it start over;
// false after four of those: the caller runs its add chain.
__device__ __forceinline__ bool BinadeChunkTotal(float carry, float d, int lane, float* total) {
  int start = 0;
  for (int iter = 0; iter < 4; ++iter) {
    const uint32_t cb = __float_as_uint(carry);
    const uint32_t e = cb >> 23;                     // sign bit set => e >= 256
    const bool range_ok = e >= 30u && e < 254u;
    const uint32_t bb = cb & 0xFF800000u;
    const float B = __uint_as_float(bb);
    const float t = __fadd_rn(B, d);
    const float err = __fsub_rn(d, __fsub_rn(t, B));
    const float half_ulp = __uint_as_float(bb - (24u << 23));
    const bool active = lane >= start;
    const bool ok = d >= 0.f && fabsf(err) != half_ulp && t < __fadd_rn(B, B);
    const uint32_t n = active && ok ? __float_as_uint(t) - bb : 0u;           // < 2^23
    const uint32_t off = (cb - bb) + WaveInclusiveAdd(n, lane);               // < 2^30
    const unsigned long long prob =
        __ballot(active && (!range_ok || !ok || off >= (1u << 23)));
    if (prob == 0) {
      *total = __uint_as_float(bb + (uint32_t)__builtin_amdgcn_readlane((int)off, 63));
      return true;
    }
    const int c = __ffsll((long long)prob) - 1;
    const float before =
        c == start ? carry
                   : __uint_as_float(bb + (uint32_t)__builtin_amdgcn_readlane((int)off, c - 1));
    carry = __fadd_rn(before, ReadLaneF(d, c));
    start = c + 1;
    if (start == 64) { *total = carry; return true; }
  }
  return false;
}

}  // namespace euler_gpu

#endif  // EULER_AMD_CSRC_WAVE_SUMS_H_
