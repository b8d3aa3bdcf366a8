// Counter-based RNG of the sampling path (host + gfx950 device).
//
// The reference draws every random number from one function,
// `double euler::common::ThreadLocalRandom()` (euler/common/random.cc:22-27),
// a thread_local minstd engine seeded with time(0): sequential, unseedable and
// dependent on which pool thread runs the query.  A GPU sampler needs the
// opposite: a stateless map (who, which draw) -> uniform double that any lane
// can evaluate.  This header defines that map; DESIGN.md "RNG contract" is the
// prose version and oracle/eo_rng.h is the independently written checker.
//
//   key  = (seed_lo, seed_hi ^ kDomainSalt[domain])
//   ctr  = (call_id, stream_lo, stream_hi, draw_idx >> 1)
//   w[4] = Philox4x32-10(ctr, key)
//   draw d -> (a, b) = (w[2*(d&1)], w[2*(d&1)+1])
//   u = ((a >> 5) * 2^26 + (b >> 6)) * 2^-53                      in [0, 1)
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define EG_HD __host__ __device__ __forceinline__
#else
#define EG_HD inline
#endif

namespace euler_gpu {

enum RngDomain : uint32_t {
  kDomainNeighbor = 0,  // stream = root node id
  kDomainNode = 1,      // stream = 0, draws in Graph::SampleNode program order
  kDomainWalk = 2,      // stream = walker index (node2vec step)
  kDomainSplit = 3,     // SAMPLE_NODE_SPLIT remainder
  kDomainRoot = 4,      // API_SAMPLE_ROOT: stream = batch row, 2 draws per sample
  kDomainLayer = 5,     // API_SAMPLE_L: stream = POSITION in the root list (the
                        // same node drawn twice samples twice), draws of one
                        // Node::SampleNeighbor(count = 1)
  kDomainLocalLayer = 6 // API_LOCAL_SAMPLE_L: stream = batch row, draw j = sample j
};

EG_HD uint32_t DomainSalt(uint32_t domain) {
  return domain == 0 ? 0x00000000u
       : domain == 1 ? 0x9E3779B9u
       : domain == 2 ? 0x7F4A7C15u
       : domain == 3 ? 0xF39CC060u
       : domain == 4 ? 0x6A09E667u
       : domain == 5 ? 0xB5C0FBCFu
                     : 0x3C6EF372u;
}

struct Philox4 {
  uint32_t w[4];
};

// One 32 x 32 -> 64 multiply per half round: on gfx950 a single v_mad_u64_u32 yields both
// words (v_mul_hi_u32 + v_mul_lo_u32 are two quarter-rate instructions; ten rounds make 40 of
// them per block against 20)
EG_HD Philox4 Philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                            uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0;
    const uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  Philox4 out;
  out.w[0] = c0; out.w[1] = c1; out.w[2] = c2; out.w[3] = c3;
  return out;
}

// One Philox block = two uniform doubles (draws 2*block and 2*block+1).
EG_HD Philox4 RngBlock(uint64_t seed, uint32_t call_id, uint32_t domain,
                       uint64_t stream, uint32_t block) {
  return Philox4x32_10(call_id, (uint32_t)stream, (uint32_t)(stream >> 32),
                       block, (uint32_t)seed,
                       (uint32_t)(seed >> 32) ^ DomainSalt(domain));
}

EG_HD double UnitFromWords(uint32_t a, uint32_t b) {
  // exact: 27 + 26 = 53 mantissa bits
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) *
         (1.0 / 9007199254740992.0);
}

EG_HD double RngDraw(uint64_t seed, uint32_t call_id, uint32_t domain,
                     uint64_t stream, uint64_t draw_idx) {
  const Philox4 b = RngBlock(seed, call_id, domain, stream,
                             (uint32_t)(draw_idx >> 1));
  const int h = (int)(draw_idx & 1);
  return UnitFromWords(b.w[2 * h], b.w[2 * h + 1]);
}

}  // namespace euler_gpu
