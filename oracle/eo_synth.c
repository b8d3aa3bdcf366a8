/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * Deterministic synthetic power-law graph used by the benchmark configs
 * (SURVEY.md §8d configs 2/3): every quantity is a pure function of
 * (seed, node id, edge slot), so the CPU baseline and the GPU run sample the
 * SAME graph and any row can be regenerated on the host for a spot check.
 * The device mirror is euler_amd/csrc/graph_build.hip; tests compare the two
 * bit-for-bit.
 *
 * Model: RMAT(a,b,c,d = 0.57,0.19,0.19,0.05) marginals.  In RMAT the source
 * id's bits are independent with P(bit=1) = c+d, so the expected out-degree of
 * node x is proportional to (a+b)^(#0 bits) (c+d)^(#1 bits); the destination's
 * bit i is 1 with probability d/(c+d) if the source bit is 1 and b/(a+b)
 * otherwise.  We draw the degree as 1 + round-by-hash(lambda[popcount(x)])
 * (minimum degree 1, so uniformly drawn roots are never isolated) and the
 * destination bits from a 64-bit mixer in 16-bit slices.  No floating-point
 * transcendental is evaluated per node: lambda comes from a 64-entry table
 * filled once on the host and handed to both implementations.
 */
#include <math.h>
#include <string.h>

#include "euler_oracle.h"

static uint64_t sy_mix64(uint64_t z) {
  z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL;
  z ^= z >> 27; z *= 0x94d049bb133111ebULL;
  z ^= z >> 31;
  return z;
}

static uint64_t sy_hash(uint64_t seed, uint64_t node, uint64_t j, uint64_t c) {
  uint64_t h = sy_mix64(seed + node * 0x9E3779B97F4A7C15ULL);
  h = sy_mix64(h + j * 0xD1B54A32D192ED03ULL + c * 0x8CB92BA72F3D8DD7ULL);
  return h;
}

/* hashed_ids: the id a node is known by outside - arbitrary u64 ids, as a .dat dataset has.
 * mix64 is a bijection of u64 with mix64(0) = 0, so ids are distinct and never the sentinel 0. */
uint64_t eo_synth_external_id(const eo_synth_params* p, uint64_t node_id) {
  return p->hashed_ids ? sy_mix64(node_id) : node_id;
}

void eo_synth_fill_table(eo_synth_params* p) {
  const double p1 = 0.24, p0 = 0.76;
  const int S = p->scale;
  uint64_t mask = S >= 64 ? ~0ULL : ((1ULL << S) - 1);
  /* count nodes per popcount among x = 0..n_nodes-1 */
  double cnt[65];
  for (int z = 0; z <= 64; ++z) cnt[z] = 0;
  for (int64_t x = 0; x < p->n_nodes; ++x)
    cnt[__builtin_popcountll((uint64_t)x & mask)] += 1.0;
  double wz[65], norm = 0;
  for (int z = 0; z <= S; ++z) {
    wz[z] = pow(p0, S - z) * pow(p1, z);
    norm += cnt[z] * wz[z];
  }
  double extra = (double)(p->n_edges_target - p->n_nodes);
  if (extra < 0) extra = 0;
  for (int z = 0; z < 64; ++z)
    p->deg_table[z] = (z <= S && norm > 0) ? extra * wz[z] / norm : 0.0;
}

int64_t eo_synth_degree(const eo_synth_params* p, uint64_t node_id) {
  uint64_t x = node_id - 1;
  uint64_t mask = p->scale >= 64 ? ~0ULL : ((1ULL << p->scale) - 1);
  double lam = p->deg_table[__builtin_popcountll(x & mask) & 63];
  double fl = floor(lam);
  double frac = lam - fl;
  uint64_t h = sy_hash(p->seed, node_id, ~0ULL, 0);
  double u = (double)(h >> 11) * (1.0 / 9007199254740992.0);
  return 1 + (int64_t)fl + (u < frac ? 1 : 0);
}

uint64_t eo_synth_neighbor(const eo_synth_params* p, uint64_t node_id,
                           int64_t j) {
  uint64_t x = node_id - 1;
  uint64_t bits = 0;
  uint64_t h = 0;
  for (int i = 0; i < p->scale; ++i) {
    if ((i & 3) == 0) h = sy_hash(p->seed, node_id, (uint64_t)j, 1 + (i >> 2));
    uint32_t slice = (uint32_t)(h >> (16 * (i & 3))) & 0xFFFFu;
    uint32_t thr = ((x >> i) & 1) ? 13653u : 16384u;
    if (slice < thr) bits |= 1ULL << i;
  }
  return bits % (uint64_t)p->n_nodes + 1;
}

float eo_synth_weight(const eo_synth_params* p, uint64_t node_id, int64_t j) {
  if (!p->weighted) return 1.0f;
  uint64_t h = sy_hash(p->seed, node_id, (uint64_t)j, 0);
  float x = (float)(uint32_t)(h >> 40) * (1.0f / 16777216.0f);
  float y = 7.5f * x;
  return 0.5f + y;
}

/* Edge slots of a row are split evenly over the types:
 * type t owns slots [deg*t/T, deg*(t+1)/T). */
int32_t eo_synth_type(const eo_synth_params* p, uint64_t node_id, int64_t j) {
  int64_t deg = eo_synth_degree(p, node_id);
  for (int32_t t = 0; t < p->n_types; ++t)
    if (j < deg * (t + 1) / p->n_types) return t;
  return p->n_types - 1;
}

/* Materialise rows [row_begin, row_end) (ids row+1) into reference-format
 * arrays.  Pass nbr == NULL to only count: returns the edge total. */
int64_t eo_synth_build(const eo_synth_params* p, int64_t row_begin,
                       int64_t row_end, int64_t* row_ptr, int32_t* type_end,
                       uint64_t* nbr, float* prefix_w, float* type_prefix) {
  const int32_t T = p->n_types;
  int64_t off = 0;
  for (int64_t r = row_begin; r < row_end; ++r) {
    uint64_t id = (uint64_t)r + 1;
    int64_t deg = eo_synth_degree(p, id);
    int64_t i = r - row_begin;
    if (row_ptr) row_ptr[i] = off;
    if (nbr) {
      float sum = 0, tsum = 0;
      int32_t t = 0;
      float tw = 0;
      for (int64_t j = 0; j < deg; ++j) {
        while (t < T - 1 && j >= deg * (t + 1) / T) {
          type_end[i * T + t] = (int32_t)j;
          tsum += tw; type_prefix[i * T + t] = tsum; tw = 0; ++t;
        }
        float w = eo_synth_weight(p, id, j);
        sum += w; tw += w;
        nbr[off + j] = eo_synth_external_id(p, eo_synth_neighbor(p, id, j));
        prefix_w[off + j] = sum;
      }
      while (t < T) {
        type_end[i * T + t] = (int32_t)deg;
        tsum += tw; type_prefix[i * T + t] = tsum; tw = 0; ++t;
      }
    }
    off += deg;
  }
  if (row_ptr) row_ptr[row_end - row_begin] = off;
  return off;
}
