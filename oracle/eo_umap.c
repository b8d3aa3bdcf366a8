/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * API_LOCAL_SAMPLE_L (euler/core/kernels/local_sample_layer_op.cc:43-146) hands
 * its sampler the distinct (neighbour id, edge type) pairs of a batch row in the
 * ITERATION ORDER of a std::unordered_map<std::string, ...>.  That order is not in
 * /root/reference: it belongs to the C++ standard library the reference is built
 * against - here libstdc++ of GCC 11.4.0 (GLIBCXX_3.4.30, the toolchain of this
 * image).  This file restates the published algorithm of that dependency:
 *   std::hash<std::string>      = _Hash_bytes(data, len, 0xc70f6907), the 64-bit
 *                                 Murmur-style hash of libsupc++/hash_bytes.cc;
 *   _Prime_rehash_policy        = max load factor 1, growth factor 2, bucket
 *                                 counts from __prime_list (first entries below),
 *                                 __fast_bkt for requests < 14, 11 as the first
 *                                 minimum (include/bits/hashtable_policy.h,
 *                                 src/c++11/hashtable_c++0x.cc);
 *   _Hashtable (unique keys)    = one singly linked list; a node entering an empty
 *                                 bucket goes to the FRONT of the list, a node
 *                                 entering a non-empty bucket goes right after the
 *                                 bucket's before-node; a rehash relinks the nodes
 *                                 in list order by the same two rules
 *                                 (include/bits/hashtable.h: _M_insert_bucket_begin,
 *                                 _M_rehash_aux(unique)).
 * Pinned by tests/test_oracle_vs_ref.py against the real container inside
 * oracle/_ref (random key sequences and the op's outputs).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "euler_oracle.h"

static const unsigned long eo_prime_list[] = {
  2UL, 3UL, 5UL, 7UL, 11UL, 13UL, 17UL, 19UL, 23UL, 29UL,
  31UL, 37UL, 41UL, 43UL, 47UL, 53UL, 59UL, 61UL, 67UL, 71UL,
  73UL, 79UL, 83UL, 89UL, 97UL, 103UL, 109UL, 113UL, 127UL, 137UL,
  139UL, 149UL, 157UL, 167UL, 179UL, 193UL, 199UL, 211UL, 227UL, 241UL,
  257UL, 277UL, 293UL, 313UL, 337UL, 359UL, 383UL, 409UL, 439UL, 467UL,
  503UL, 541UL, 577UL, 619UL, 661UL, 709UL, 761UL, 823UL, 887UL, 953UL,
  1031UL, 1109UL, 1193UL, 1289UL, 1381UL, 1493UL, 1613UL, 1741UL, 1879UL, 2029UL,
  2179UL, 2357UL, 2549UL, 2753UL, 2971UL, 3209UL, 3469UL, 3739UL, 4027UL, 4349UL,
  4703UL, 5087UL, 5503UL, 5953UL, 6427UL, 6949UL, 7517UL, 8123UL, 8783UL, 9497UL,
  10273UL, 11113UL, 12011UL, 12983UL, 14033UL, 15173UL, 16411UL, 17749UL, 19183UL, 20753UL,
  22447UL, 24281UL, 26267UL, 28411UL, 30727UL, 33223UL, 35933UL, 38873UL, 42043UL, 45481UL,
  49201UL, 53201UL, 57557UL, 62233UL, 67307UL, 72817UL, 78779UL, 85229UL, 92203UL, 99733UL,
  107897UL, 116731UL, 126271UL, 136607UL, 147793UL, 159871UL, 172933UL, 187091UL, 202409UL, 218971UL,
  236897UL, 256279UL, 277261UL, 299951UL, 324503UL, 351061UL, 379787UL, 410857UL, 444487UL, 480881UL
};
#define EO_PRIME_COUNT (sizeof(eo_prime_list) / sizeof(eo_prime_list[0]))

static uint64_t eo_shift_mix(uint64_t v) { return v ^ (v >> 47); }

/* libsupc++ hash_bytes.cc, the size_t == 8 implementation */
uint64_t eo_std_hash_bytes(const void* ptr, uint64_t len) {
  const uint64_t mul = (((uint64_t)0xc6a4a793UL) << 32) + (uint64_t)0x5bd1e995UL;
  const unsigned char* buf = (const unsigned char*)ptr;
  const uint64_t len_aligned = len & ~(uint64_t)0x7;
  const unsigned char* end = buf + len_aligned;
  uint64_t hash = 0xc70f6907UL ^ (len * mul);
  for (const unsigned char* p = buf; p != end; p += 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    const uint64_t data = eo_shift_mix(w * mul) * mul;
    hash ^= data;
    hash *= mul;
  }
  if ((len & 0x7) != 0) {
    uint64_t data = 0;
    for (int i = (int)(len & 0x7) - 1; i >= 0; --i) data = (data << 8) + end[i];
    hash ^= data;
    hash *= mul;
  }
  hash = eo_shift_mix(hash) * mul;
  hash = eo_shift_mix(hash);
  return hash;
}

typedef struct {
  uint64_t next_resize;    /* _Prime_rehash_policy::_M_next_resize */
} eo_policy;

static uint64_t eo_next_bkt(eo_policy* p, uint64_t n) {
  static const unsigned char fast_bkt[] = {2, 2, 2, 3, 5, 5, 7, 7, 11, 11, 11, 11, 13, 13};
  if (n < sizeof(fast_bkt)) {
    if (n == 0) return 1;
    p->next_resize = (uint64_t)floor(fast_bkt[n] * 1.0);
    return fast_bkt[n];
  }
  /* lower_bound(__prime_list + 6, last, n) */
  size_t lo = 6, hi = EO_PRIME_COUNT- 1;
  while (lo < hi) {
    size_t mid = (lo + hi) / 2;
    if (eo_prime_list[mid] < n) lo = mid + 1; else hi = mid;
  }
  p->next_resize = (uint64_t)floor(eo_prime_list[lo] * 1.0);
  return eo_prime_list[lo];
}

/* returns the new bucket count, or 0 for "no rehash" */
static uint64_t eo_need_rehash(eo_policy* p, uint64_t n_bkt, uint64_t n_elt, uint64_t n_ins) {
  if (n_elt + n_ins > p->next_resize) {
    uint64_t want = n_elt + n_ins;
    if (p->next_resize == 0 && want < 11) want = 11;
    double min_bkts = (double)want / 1.0;
    if (min_bkts >= (double)n_bkt) {
      uint64_t a = (uint64_t)floor(min_bkts) + 1, b = n_bkt * 2;
      return eo_next_bkt(p, a > b ? a : b);
    }
    p->next_resize = (uint64_t)floor((double)n_bkt * 1.0);
    return 0;
  }
  return 0;
}

/* Iteration order of a std::unordered_map after inserting n DISTINCT keys with
 * the given hash codes in index order: order[k] = index of the k-th element. */
void eo_umap_iteration_order(const uint64_t* hash, int64_t n, int64_t* order) {
  /* node i: next[i]; node index n = the before-begin sentinel */
  int64_t* next = (int64_t*)malloc((size_t)(n + 1) * sizeof(int64_t));
  int64_t n_bkt = 1;
  int64_t* bucket = (int64_t*)malloc(sizeof(int64_t));   /* before-node of a bucket, -1 = empty */
  bucket[0] = -1;
  const int64_t BB = n;
  next[BB] = -1;
  eo_policy pol = {0};
  for (int64_t i = 0; i < n; ++i) {
    uint64_t nb = eo_need_rehash(&pol, (uint64_t)n_bkt, (uint64_t)i, 1);
    if (nb) {                                            /* _M_rehash_aux(nb, unique) */
      int64_t* nbuckets = (int64_t*)malloc((size_t)nb * sizeof(int64_t));
      for (uint64_t b = 0; b < nb; ++b) nbuckets[b] = -1;
      int64_t p = next[BB];
      next[BB] = -1;
      uint64_t bbegin_bkt = 0;
      while (p >= 0) {
        int64_t nx = next[p];
        uint64_t b = hash[p] % nb;
        if (nbuckets[b] < 0) {
          next[p] = next[BB];
          next[BB] = p;
          nbuckets[b] = BB;
          if (next[p] >= 0) nbuckets[bbegin_bkt] = p;
          bbegin_bkt = b;
        } else {
          next[p] = next[nbuckets[b]];
          next[nbuckets[b]] = p;
        }
        p = nx;
      }
      free(bucket);
      bucket = nbuckets;
      n_bkt = (int64_t)nb;
    }
    uint64_t b = hash[i] % (uint64_t)n_bkt;              /* _M_insert_bucket_begin */
    if (bucket[b] >= 0) {
      next[i] = next[bucket[b]];
      next[bucket[b]] = i;
    } else {
      next[i] = next[BB];
      next[BB] = i;
      if (next[i] >= 0) bucket[hash[next[i]] % (uint64_t)n_bkt] = i;
      bucket[b] = BB;
    }
  }
  int64_t k = 0;
  for (int64_t p = next[BB]; p >= 0; p = next[p]) order[k++] = p;
  free(next);
  free(bucket);
}

/* API_LOCAL_SAMPLE_L (local_sample_layer_op.cc:43-146).  idx [batch*n, 2] / ids /
 * w / t: the API_GET_NB_NODE result; out_* [batch * m].  RNG: domain LOCAL_LAYER,
 * stream = batch row. */
void eo_local_sample_layer(uint64_t seed, uint32_t call_id, const int32_t* idx,
                           int64_t idx_elems, const uint64_t* ids, const float* w,
                           const int32_t* t, int32_t n, int32_t m, int32_t take_sqrt,
                           int64_t default_node, uint64_t* o_nb, float* o_w, int32_t* o_t) {
  const int32_t batch = (int32_t)(idx_elems / (n * 2));
  for (int32_t i = 0; i < batch; ++i) {
    const int32_t begin = idx[(int64_t)i * n * 2];
    const int32_t end = i < batch - 1 ? idx[(int64_t)(i + 1) * n * 2] : idx[idx_elems - 1];
    const int64_t cap = end > begin ? end - begin : 0;
    /* distinct (id, type) pairs in first-occurrence order, weights accumulated */
    uint64_t* u_id = (uint64_t*)malloc((size_t)(cap + 1) * 8);
    int32_t* u_t = (int32_t*)malloc((size_t)(cap + 1) * 4);
    float* u_w = (float*)malloc((size_t)(cap + 1) * 4);
    uint64_t* u_h = (uint64_t*)malloc((size_t)(cap + 1) * 8);
    int64_t tcap = 16;
    while (tcap < 2 * cap + 2) tcap <<= 1;
    int64_t* slot = (int64_t*)malloc((size_t)tcap * 8);   /* oracle-side lookup, any hash */
    for (int64_t s = 0; s < tcap; ++s) slot[s] = -1;
    int64_t nu = 0;
    for (int32_t j = begin; j < end; ++j) {
      char key[48];
      /* std::to_string(dst_id) + std::to_string(type): same key <=> same pair
       * EXCEPT where the digits run together (id 12, type 3 vs id 1, type 23),
       * which the reference merges too - so the lookup is by the string */
      int len = snprintf(key, sizeof key, "%llu%d", (unsigned long long)ids[j], t[j]);
      uint64_t h = eo_std_hash_bytes(key, (uint64_t)len);
      int64_t s = (int64_t)(h & (uint64_t)(tcap - 1));
      int64_t found = -1;
      while (slot[s] >= 0) {
        int64_t e = slot[s];
        if (u_h[e] == h) {
          char k2[48];
          int l2 = snprintf(k2, sizeof k2, "%llu%d", (unsigned long long)u_id[e], u_t[e]);
          if (l2 == len && memcmp(k2, key, (size_t)len) == 0) { found = e; break; }
        }
        s = (s + 1) & (tcap - 1);
      }
      if (found < 0) {
        slot[s] = nu; u_id[nu] = ids[j]; u_t[nu] = t[j]; u_w[nu] = w[j]; u_h[nu] = h; ++nu;
      } else {
        u_w[found] += w[j];
      }
    }
    int64_t* order = (int64_t*)malloc((size_t)(nu + 1) * 8);
    eo_umap_iteration_order(u_h, nu, order);
    float* sum_w = (float*)malloc((size_t)(nu + 1) * 4);
    float acc = 0.0f;
    for (int64_t k = 0; k < nu; ++k) {
      float x = u_w[order[k]];
      if (take_sqrt) x = sqrtf(x);
      u_w[order[k]] = x;
      acc += x;
      sum_w[k] = acc;
    }
    if (nu == 0 || acc == 0) {
      memset(o_nb + (int64_t)i * m, (int)default_node, sizeof(uint64_t) * (size_t)m);
      memset(o_w + (int64_t)i * m, 0, sizeof(float) * (size_t)m);
      memset(o_t + (int64_t)i * m, 0, sizeof(int32_t) * (size_t)m);
    } else {
      eo_rng_ctx rng = {seed, call_id, EO_DOMAIN_LOCAL_LAYER, (uint64_t)i, 0};
      for (int32_t j = 0; j < m; ++j) {
        int64_t mid = eo_random_select(sum_w, 0, (uint64_t)(nu - 1), eo_next_uniform(&rng));
        int64_t e = order[mid];
        o_nb[(int64_t)i * m + j] = u_id[e];
        o_w[(int64_t)i * m + j] = u_w[e];
        o_t[(int64_t)i * m + j] = u_t[e];
      }
    }
    free(u_id); free(u_t); free(u_w); free(u_h); free(slot); free(order); free(sum_w);
  }
}
