/*
 * ORACLE (test infrastructure, NOT product code) - see euler_oracle.h.
 * Plain-C restatement of the reference algorithms; every function cites the
 * reference file:line it follows (paths relative to /root/reference).
 * Build: gcc -std=c99 -O2 -ffp-contract=off (oracle/Makefile).
 */
#include "euler_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ utils */

static uint64_t eo_mix64(uint64_t z) {
  z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL;
  z ^= z >> 27; z *= 0x94d049bb133111ebULL;
  z ^= z >> 31;
  return z;
}

void eo_philox_kat(const uint32_t ctr[4], const uint32_t key[2],
                   uint32_t out[4]) {
  eo_philox4x32_10(ctr, key, out);
}

double eo_uniform_at(uint64_t seed, uint32_t call_id, uint32_t domain,
                     uint64_t stream, uint64_t draw_idx) {
  eo_rng_ctx c = {seed, call_id, domain, stream, draw_idx};
  return eo_next_uniform(&c);
}

/* ------------------------------------------------------------------ graph */

eo_graph* eo_graph_create(int64_t n_rows, int32_t n_types,
                          const uint64_t* row_id, const int64_t* row_ptr,
                          const int32_t* type_end, const uint64_t* nbr,
                          const float* prefix_w, const float* type_prefix) {
  eo_graph* g = (eo_graph*)calloc(1, sizeof(eo_graph));
  g->n_rows = n_rows; g->n_types = n_types; g->row_id = row_id;
  g->row_ptr = row_ptr; g->type_end = type_end; g->nbr = nbr;
  g->prefix_w = prefix_w; g->type_prefix = type_prefix;
  uint64_t cap = 16;
  while (cap < (uint64_t)n_rows * 2 + 1) cap <<= 1;
  g->hash_cap = cap;
  g->hash_key = (uint64_t*)malloc(cap * sizeof(uint64_t));
  g->hash_row = (int64_t*)malloc(cap * sizeof(int64_t));
  for (uint64_t i = 0; i < cap; ++i) g->hash_row[i] = -1;
  /* Graph::AddNode: node_map_[id] = n (graph.cc:162-166): last insert wins */
  for (int64_t r = 0; r < n_rows; ++r) {
    uint64_t h = eo_mix64(row_id[r]) & (cap - 1);
    while (g->hash_row[h] >= 0 && g->hash_key[h] != row_id[r])
      h = (h + 1) & (cap - 1);
    g->hash_key[h] = row_id[r];
    g->hash_row[h] = r;
  }
  return g;
}

void eo_graph_destroy(eo_graph* g) {
  if (!g) return;
  free(g->hash_key); free(g->hash_row); free(g);
}

/* Graph::GetNodeByID (graph.h:87-92): miss -> nullptr (-1 here). */
int64_t eo_graph_find_row(const eo_graph* g, uint64_t id) {
  uint64_t h = eo_mix64(id) & (g->hash_cap - 1);
  while (g->hash_row[h] >= 0) {
    if (g->hash_key[h] == id) return g->hash_row[h];
    h = (h + 1) & (g->hash_cap - 1);
  }
  return -1;
}

/* Node::Init (node.cc:46-66): sum_weight runs across ALL types of the node in
 * f32; type_weight per type in f32; CompactWeightedCollection::Init
 * (compact_weighted_collection.h:84-100) then running-sums the type weights. */
void eo_build_prefix(int64_t n_rows, int32_t n_types, const int64_t* seg_ptr,
                     const float* w, int64_t* row_ptr, int32_t* type_end,
                     float* prefix_w, float* type_prefix) {
  for (int64_t i = 0; i < n_rows; ++i) {
    float sum_weight = 0;
    float type_sum = 0;
    int64_t base = seg_ptr[i * n_types];
    row_ptr[i] = base;
    for (int32_t t = 0; t < n_types; ++t) {
      float type_weight = 0;
      for (int64_t j = seg_ptr[i * n_types + t];
           j < seg_ptr[i * n_types + t + 1]; ++j) {
        sum_weight += w[j];
        type_weight += w[j];
        prefix_w[j] = sum_weight;
      }
      type_end[i * n_types + t] = (int32_t)(seg_ptr[i * n_types + t + 1] - base);
      type_sum += type_weight;
      type_prefix[i * n_types + t] = type_sum;
    }
  }
  row_ptr[n_rows] = seg_ptr[n_rows * n_types];
}

/* ---------------------------------------------------------- RandomSelect */

/* RandomSelect<T> (compact_weighted_collection.h:30-52) with the uniform
 * draw passed in.  size_t arithmetic kept (Q4); returns last `mid` on the
 * fall-through (Q3). */
int64_t eo_random_select(const float* sum_weights, uint64_t begin_pos,
                         uint64_t end_pos, double u) {
  float limit_begin = begin_pos == 0 ? 0 : sum_weights[begin_pos - 1];
  float limit_end = sum_weights[end_pos];
  double r = u * (limit_end - limit_begin) + limit_begin;
  uint64_t low = begin_pos, high = end_pos, mid = 0;
  int finish = 0;
  while (low <= high && !finish) {
    mid = (low + high) / 2;
    float interval_begin = mid == 0 ? 0 : sum_weights[mid - 1];
    float interval_end = sum_weights[mid];
    if (interval_begin <= r && r < interval_end) {
      finish = 1;
    } else if (interval_begin > r) {
      high = mid - 1;
    } else if (interval_end <= r) {
      low = mid + 1;
    }
  }
  return (int64_t)mid;
}

/* ------------------------------------------------- Node::SampleNeighbor */

/* CompactWeightedCollection<int32_t>::Get(t) weight = running-sum diff
 * (compact_weighted_collection.h:127-141). */
static float eo_type_weight(const float* tp, int32_t t) {
  return tp[t] - (t > 0 ? tp[t - 1] : 0);
}

/* Node::__SampleNeighbor (node.cc:98-161) for one row.  Returns `count`, or
 * 0 for the error / empty returns. */
static int32_t eo_sample_row(const eo_graph* g, int64_t row,
                             const int32_t* edge_types, int32_t k,
                             int32_t count, eo_rng_ctx* rng, uint64_t* out_id,
                             float* out_w, int32_t* out_t) {
  const int32_t T = g->n_types;
  const int32_t* groups_idx = g->type_end + row * T;
  const float* tp = g->type_prefix + row * T;
  const uint64_t* nbr = g->nbr + g->row_ptr[row];
  const float* nw = g->prefix_w + g->row_ptr[row];
  float sub_sum[64];
  int32_t sub_ids[64];
  int use_sub = (k > 1 && k < T);
  if (use_sub) {                                   /* node.cc:106-121 */
    if (k > 64) return 0;
    float s = 0;
    for (int32_t i = 0; i < k; ++i) {
      int32_t et = edge_types[i];
      if (et >= 0 && et < T) {
        sub_ids[i] = et;
        s += eo_type_weight(tp, et);
        sub_sum[i] = s;
      } else {
        return 0;                                  /* err_vec */
      }
    }
  }
  for (int32_t i = 0; i < count; ++i) {            /* node.cc:123-159 */
    int32_t edge_type = 0;
    if (k == 1) {
      edge_type = edge_types[0];
      if (edge_type < 0 || edge_type >= T) return 0;
      int32_t pre_idx = edge_type == 0 ? 0 : groups_idx[edge_type - 1];
      int32_t cur_idx = groups_idx[edge_type] - 1;
      if (cur_idx < pre_idx) return 0;
    } else if (use_sub) {
      if (sub_sum[k - 1] == 0) return 0;
      int64_t m = eo_random_select(sub_sum, 0, (uint64_t)(k - 1),
                                   eo_next_uniform(rng));
      edge_type = sub_ids[m];
    } else {
      if (tp[T - 1] == 0) return 0;
      edge_type = (int32_t)eo_random_select(tp, 0, (uint64_t)(T - 1),
                                            eo_next_uniform(rng));
    }
    int32_t b = edge_type == 0 ? 0 : groups_idx[edge_type - 1];
    int32_t e = groups_idx[edge_type] - 1;
    /* int32 -> size_t conversion as in the call at node.cc:153-154 */
    int64_t mid = eo_random_select(nw, (uint64_t)(int64_t)b,
                                   (uint64_t)(int64_t)e,
                                   eo_next_uniform(rng));
    float pre = mid <= 0 ? 0 : nw[mid - 1];
    out_id[i] = nbr[mid];
    out_w[i] = nw[mid] - pre;
    out_t[i] = edge_type;
  }
  return count;
}

/* api.cc:223-236 loop + sample_neighbor_op.cc:134-143 fill + FillNeighbor
 * (common.cc:275-334).  RNG stream per root = (call_id, node id). */
int64_t eo_sample_neighbor_core(const eo_graph* g, uint64_t seed,
                                uint32_t call_id, const uint64_t* ids,
                                int64_t n, const int32_t* edge_types,
                                int32_t k, int32_t count, int32_t* idx,
                                uint64_t* out_id, float* out_w,
                                int32_t* out_t) {
  int64_t off = 0;
  for (int64_t i = 0; i < n; ++i) {
    int32_t got = 0;
    int64_t row = eo_graph_find_row(g, ids[i]);
    if (row >= 0 && count > 0) {
      eo_rng_ctx rng = {seed, call_id, EO_DOMAIN_NEIGHBOR, ids[i], 0};
      got = eo_sample_row(g, row, edge_types, k, count, &rng, out_id + off,
                          out_w + off, out_t + off);
    }
    if (got == 0) {
      for (int32_t j = 0; j < count; ++j) {
        out_id[off + j] = 0; out_w[off + j] = 0; out_t[off + j] = 0;
      }
    }
    if (idx) { idx[2 * i] = (int32_t)off; idx[2 * i + 1] = (int32_t)(off + count); }
    off += count;
  }
  return off;
}

/* TF dense repack (tf_euler/kernels/sample_neighbor_op.cc:79-81,110-122):
 * prefill default_node / 0.0 / -1, copy the row unless its FIRST id is the
 * sentinel 0 (Q1). */
static void eo_tf_repack(const uint64_t* cid, const float* cw,
                         const int32_t* ct, int64_t n, int32_t count,
                         int64_t default_node, int64_t* out_n, float* out_w,
                         int32_t* out_t) {
  for (int64_t i = 0; i < n * count; ++i) {
    out_n[i] = default_node; out_w[i] = 0.0f; out_t[i] = -1;
  }
  if (count <= 0) return;
  for (int64_t i = 0; i < n; ++i) {
    if (cid[i * count] != 0) {
      for (int32_t j = 0; j < count; ++j) {
        out_n[i * count + j] = (int64_t)cid[i * count + j];
        out_w[i * count + j] = cw[i * count + j];
        out_t[i * count + j] = ct[i * count + j];
      }
    }
  }
}

void eo_sample_neighbor_tf(const eo_graph* g, uint64_t seed, uint32_t call_id,
                           const int64_t* nodes, int64_t n,
                           const int32_t* edge_types, int32_t k,
                           int32_t count, int64_t default_node,
                           int64_t* out_n, float* out_w, int32_t* out_t) {
  int64_t tot = n * (int64_t)count;
  uint64_t* cid = (uint64_t*)malloc((tot + 1) * 8);
  float* cw = (float*)malloc((tot + 1) * 4);
  int32_t* ct = (int32_t*)malloc((tot + 1) * 4);
  eo_sample_neighbor_core(g, seed, call_id, (const uint64_t*)nodes, n,
                          edge_types, k, count, NULL, cid, cw, ct);
  eo_tf_repack(cid, cw, ct, n, count, default_node, out_n, out_w, out_t);
  free(cid); free(cw); free(ct);
}

/* TF SampleFanout (tf_euler/kernels/sample_fanout_op.cc:60-145): one GQL
 * chaining the hops on the CORE id tensors (sentinel-0 rows included); hop h
 * uses call_id + h. edge_types is [layers, k]. */
void eo_sample_fanout_tf(const eo_graph* g, uint64_t seed, uint32_t call_id,
                         const int64_t* nodes, int64_t n,
                         const int32_t* edge_types, int32_t k,
                         const int32_t* counts, int32_t layers,
                         int64_t default_node, int64_t** out_n,
                         float** out_w, int32_t** out_t) {
  uint64_t* roots = (uint64_t*)malloc((n + 1) * 8);
  memcpy(roots, nodes, n * 8);
  int64_t m = n;
  for (int32_t h = 0; h < layers; ++h) {
    int64_t tot = m * (int64_t)counts[h];
    uint64_t* cid = (uint64_t*)malloc((tot + 1) * 8);
    float* cw = (float*)malloc((tot + 1) * 4);
    int32_t* ct = (int32_t*)malloc((tot + 1) * 4);
    eo_sample_neighbor_core(g, seed, call_id + (uint32_t)h, roots, m,
                            edge_types + h * k, k, counts[h], NULL, cid, cw,
                            ct);
    eo_tf_repack(cid, cw, ct, m, counts[h], default_node, out_n[h], out_w[h],
                 out_t[h]);
    free(roots); free(cw); free(ct);
    roots = cid;
    m = tot;
  }
  free(roots);
}

/* Node::__GetFullNeighbor (node.cc:175-197) via api.cc:208-221. */
int64_t eo_get_full_neighbor(const eo_graph* g, const uint64_t* ids,
                             int64_t n, const int32_t* edge_types, int32_t k,
                             int32_t* idx, uint64_t* out_id, float* out_w,
                             int32_t* out_t) {
  int64_t off = 0;
  const int32_t T = g->n_types;
  for (int64_t i = 0; i < n; ++i) {
    if (idx) idx[2 * i] = (int32_t)off;
    int64_t row = eo_graph_find_row(g, ids[i]);
    if (row >= 0) {
      const int32_t* gi = g->type_end + row * T;
      const uint64_t* nbr = g->nbr + g->row_ptr[row];
      const float* nw = g->prefix_w + g->row_ptr[row];
      for (int32_t a = 0; a < k; ++a) {
        int32_t et = edge_types[a];
        if (et >= 0 && et < T) {
          int32_t b = et == 0 ? 0 : gi[et - 1];
          for (int32_t j = b; j < gi[et]; ++j) {
            if (out_id) {
              float pre = j == 0 ? 0 : nw[j - 1];
              out_id[off] = nbr[j]; out_w[off] = nw[j] - pre; out_t[off] = et;
            }
            ++off;
          }
        }
      }
    }
    if (idx) idx[2 * i + 1] = (int32_t)off;
  }
  return off;
}

/* ------------------------------------------- ID_UNIQUE / *_GATHER ops */

/* IdUnique (core/kernels/id_unique_op.cc:35-64): first-occurrence order. */
int64_t eo_id_unique(const uint64_t* ids, int64_t n, uint64_t* unique_ids,
                     int32_t* gather_idx) {
  uint64_t cap = 16;
  while (cap < (uint64_t)n * 2 + 1) cap <<= 1;
  uint64_t* key = (uint64_t*)malloc(cap * 8);
  int32_t* val = (int32_t*)malloc(cap * 4);
  for (uint64_t i = 0; i < cap; ++i) val[i] = -1;
  int32_t cnt = 0;
  for (int64_t i = 0; i < n; ++i) {
    uint64_t h = eo_mix64(ids[i]) & (cap - 1);
    while (val[h] >= 0 && key[h] != ids[i]) h = (h + 1) & (cap - 1);
    if (val[h] < 0) { key[h] = ids[i]; val[h] = cnt; unique_ids[cnt++] = ids[i]; }
    gather_idx[i] = val[h];
  }
  free(key); free(val);
  return cnt;
}

/* IdxGather (core/kernels/idx_gather_op.cc:33-55). */
void eo_idx_gather(const int32_t* idx, const int32_t* gather_idx, int64_t n,
                   int32_t* out) {
  int32_t base = 0;
  for (int64_t i = 0; i < n; ++i) {
    int32_t a = gather_idx[i] * 2;
    out[2 * i] = base;
    out[2 * i + 1] = base + idx[a + 1] - idx[a];
    base = out[2 * i + 1];
  }
}

/* DataGather (core/kernels/data_gather_op.cc:33-46). Returns elements. */
int64_t eo_data_gather(const void* data, int32_t elem_size,
                       const int32_t* idx, const int32_t* gather_idx,
                       int64_t n, void* out) {
  int64_t base = 0;
  for (int64_t i = 0; i < n; ++i) {
    int32_t a = gather_idx[i] * 2;
    int32_t b = idx[a], e = idx[a + 1];
    if (out)
      memcpy((char*)out + base * elem_size,
             (const char*)data + (int64_t)b * elem_size,
             (size_t)(e - b) * elem_size);
    base += e - b;
  }
  return base;
}

/* ---------------------------------------------------------------- alias */

/* AliasMethod::Init (alias_method.cc:23-63): LIFO small/large stacks, avg is
 * double, prob_ float, weights_ updated in float. */
void eo_alias_init(const float* weights, int64_t n, float* prob,
                   int64_t* alias) {
  int64_t* small = (int64_t*)malloc((n + 1) * 8);
  int64_t* large = (int64_t*)malloc((n + 1) * 8);
  float* w = (float*)malloc((n + 1) * 4);
  int64_t ns = 0, nl = 0;
  memcpy(w, weights, n * 4);
  for (int64_t i = 0; i < n; ++i) alias[i] = 0;   /* vector::resize zero-fills */
  for (int64_t i = 0; i < n; ++i) prob[i] = 0;
  double avg = 1 / (double)n;
  for (int64_t i = 0; i < n; ++i) {
    if (w[i] > avg) large[nl++] = i; else small[ns++] = i;
  }
  while (nl > 0 && ns > 0) {
    int64_t less = small[--ns];
    int64_t more = large[--nl];
    prob[less] = w[less] * (float)(uint64_t)n;   /* float * size_t */
    alias[less] = more;
    w[more] = (float)((double)(w[more] + w[less]) - avg);
    if (w[more] > avg) large[nl++] = more; else small[ns++] = more;
  }
  while (ns > 0) prob[small[--ns]] = 1.0f;
  while (nl > 0) prob[large[--nl]] = 1.0f;
  free(small); free(large); free(w);
}

/* AliasMethod::Next (alias_method.cc:66-78): column = floor(n * u1);
 * u2 < prob[column] ? column : alias[column]. */
static int64_t eo_alias_next(const float* prob, const int64_t* alias,
                             int64_t n, eo_rng_ctx* rng) {
  int64_t column = (int64_t)floor((double)n * eo_next_uniform(rng));
  int toss = eo_next_uniform(rng) < prob[column];
  return toss ? column : alias[column];
}

/* Graph::BuildGlobalSampler (graph.cc:333-370) + FastWeightedCollection::Init
 * (fast_weighted_collection.h:55-75): weights normalised by the f32 type sum,
 * then normalised AGAIN by the f32 sum of the normalised weights. */
eo_node_sampler* eo_node_sampler_create(int64_t n, const uint64_t* ids,
                                        const int32_t* types,
                                        const float* weights,
                                        int32_t n_types) {
  eo_node_sampler* s = (eo_node_sampler*)calloc(1, sizeof(eo_node_sampler));
  s->n_types = n_types;
  s->type_off = (int64_t*)calloc(n_types + 1, 8);
  s->ids = (uint64_t*)malloc((n + 1) * 8);
  s->prob = (float*)malloc((n + 1) * 4);
  s->alias = (int64_t*)malloc((n + 1) * 8);
  s->type_sum = (float*)calloc(n_types, 4);
  s->sampler_sum = (float*)calloc(n_types, 4);
  s->tc_prob = (float*)calloc(n_types, 4);
  s->tc_alias = (int64_t*)calloc(n_types, 8);
  float* norm = (float*)malloc((n + 1) * 4);
  int64_t* fill = (int64_t*)calloc(n_types + 1, 8);
  for (int64_t i = 0; i < n; ++i) s->type_off[types[i] + 1]++;
  for (int32_t t = 0; t < n_types; ++t) s->type_off[t + 1] += s->type_off[t];
  for (int64_t i = 0; i < n; ++i) {
    int32_t t = types[i];
    int64_t p = s->type_off[t] + fill[t]++;
    s->ids[p] = ids[i];
    norm[p] = weights[i];
    s->type_sum[t] += weights[i];
  }
  for (int32_t t = 0; t < n_types; ++t) {
    int64_t b = s->type_off[t], e = s->type_off[t + 1];
    for (int64_t i = b; i < e; ++i) norm[i] /= s->type_sum[t];
    float sum = 0;
    for (int64_t i = b; i < e; ++i) sum += norm[i];
    s->sampler_sum[t] = sum;
    for (int64_t i = b; i < e; ++i) norm[i] /= sum;
    eo_alias_init(norm + b, e - b, s->prob + b, s->alias + b);
  }
  /* node_type_collection_.Init(node_type_ids, node_weight_sums_) */
  float tsum = 0;
  for (int32_t t = 0; t < n_types; ++t) tsum += s->type_sum[t];
  s->tc_sum = tsum;
  float* tnorm = (float*)malloc((n_types + 1) * 4);
  for (int32_t t = 0; t < n_types; ++t) tnorm[t] = s->type_sum[t] / tsum;
  eo_alias_init(tnorm, n_types, s->tc_prob, s->tc_alias);
  free(tnorm); free(norm); free(fill);
  return s;
}

void eo_node_sampler_destroy(eo_node_sampler* s) {
  if (!s) return;
  free(s->type_off); free(s->ids); free(s->prob); free(s->alias);
  free(s->type_sum); free(s->sampler_sum); free(s->tc_prob); free(s->tc_alias);
  free(s);
}

/* euler::SampleNode (api.cc:32-37) -> Graph::SampleNode (graph.cc:221-275).
 * One RNG stream per call (domain NODE, stream 0), draws in program order. */
static int64_t eo_sample_node_stream(const eo_node_sampler* s, uint64_t seed,
                                     uint32_t call_id, uint64_t stream,
                                     const int32_t* node_types, int32_t k,
                                     int32_t count, uint64_t* out);

int64_t eo_sample_node(const eo_node_sampler* s, uint64_t seed,
                       uint32_t call_id, const int32_t* node_types, int32_t k,
                       int32_t count, uint64_t* out) {
  return eo_sample_node_stream(s, seed, call_id, 0, node_types, k, count, out);
}

/* API_SAMPLE_N_WITH_TYPES (core/kernels/sample_n_with_types_op.cc:44-52) as the
 * TF kernel calls it (tf_euler/kernels/sample_n_with_types_op.cc:50-57: the same
 * count for every type): one SampleNode({type}, count) per listed type, its RNG
 * stream = the index of the call.  Returns 0, or -1 where the TF kernel would
 * abort ("samples size error, invalid node types!"). */
int eo_sample_n_with_types(const eo_node_sampler* s, uint64_t seed,
                           uint32_t call_id, const int32_t* types, int64_t n,
                           int32_t count, uint64_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    int64_t got = eo_sample_node_stream(s, seed, call_id, (uint64_t)i, types + i, 1,
                                        count, out + i * count);
    if (got != count) return -1;
  }
  return 0;
}

static int64_t eo_sample_node_stream(const eo_node_sampler* s, uint64_t seed,
                                     uint32_t call_id, uint64_t stream,
                                     const int32_t* node_types, int32_t k,
                                     int32_t count, uint64_t* out) {
  eo_rng_ctx rng = {seed, call_id, EO_DOMAIN_NODE, stream, 0};
  const int32_t T = s->n_types;
  if (k == 1) {
    int32_t type = node_types[0];
    if (type == -1) {
      if (s->tc_sum == 0) return 0;
      for (int32_t i = 0; i < count; ++i) {
        int32_t t = (int32_t)eo_alias_next(s->tc_prob, s->tc_alias, T, &rng);
        int64_t b = s->type_off[t];
        out[i] = s->ids[b + eo_alias_next(s->prob + b, s->alias + b,
                                          s->type_off[t + 1] - b, &rng)];
      }
      return count;
    }
    if (type < 0 || type >= T) return -1;
    if (s->sampler_sum[type] == 0 || s->type_off[type + 1] == s->type_off[type])
      return 0;
    int64_t b = s->type_off[type];
    for (int32_t i = 0; i < count; ++i)
      out[i] = s->ids[b + eo_alias_next(s->prob + b, s->alias + b,
                                        s->type_off[type + 1] - b, &rng)];
    return count;
  }
  /* type-list overload (graph.cc:247-275): sub collection in ascending type
   * order over the listed SET, CDF draw for the type. */
  float sub_sum[64]; int32_t sub_ids[64]; int32_t m = 0; float acc = 0;
  for (int32_t t = 0; t < T && m < 64; ++t) {
    int in = 0;
    for (int32_t j = 0; j < k; ++j) if (node_types[j] == t) in = 1;
    if (in) { acc += s->type_sum[t]; sub_ids[m] = t; sub_sum[m] = acc; ++m; }
  }
  if (m == 0 || !(sub_sum[m - 1] > 0)) return 0;
  for (int32_t i = 0; i < count; ++i) {
    int32_t t = sub_ids[eo_random_select(sub_sum, 0, (uint64_t)(m - 1),
                                         eo_next_uniform(&rng))];
    int64_t b = s->type_off[t];
    out[i] = s->ids[b + eo_alias_next(s->prob + b, s->alias + b,
                                      s->type_off[t + 1] - b, &rng)];
  }
  return count;
}

/* ----------------------------------------------------------- RandomWalk */

/* tf_euler/kernels/random_walk_op.cc: TraditionalRandomWalk (:207-247) when
 * |p-1|,|q-1| <= 1e-6, else node2vec RWCallback (:83-138) + BuildWeights
 * (:140-168). edge_types is [walk_len, k]. */
int eo_random_walk(const eo_graph* g, uint64_t seed, uint32_t call_id,
                   const int64_t* nodes, int64_t n, const int32_t* edge_types,
                   int32_t k, int32_t walk_len, float p, float q,
                   int64_t default_node, int64_t* out) {
  const int64_t L = walk_len + 1;
  for (int64_t i = 0; i < n; ++i) out[i * L] = nodes[i];
  const float kEps = 1.0e-6;
  if (fabs(p - 1.0) <= kEps && fabs(q - 1.0) <= kEps) {
    for (int64_t i = 0; i < n; ++i) {
      uint64_t cur = (uint64_t)nodes[i];
      for (int32_t s = 0; s < walk_len; ++s) {
        uint64_t id = 0; float w; int32_t t; int32_t got = 0;
        int64_t row = eo_graph_find_row(g, cur);
        if (row >= 0) {
          eo_rng_ctx rng = {seed, call_id + (uint32_t)s, EO_DOMAIN_NEIGHBOR,
                            cur, 0};
          got = eo_sample_row(g, row, edge_types + s * k, k, 1, &rng, &id, &w,
                              &t);
        }
        if (!got) id = 0;
        out[i * L + s + 1] = id == 0 ? (int64_t)(uint64_t)default_node
                                     : (int64_t)id;
        cur = id;
      }
    }
    return 0;
  }
  int64_t* parent_ids = (int64_t*)malloc((n + 1) * 8);
  int64_t* cur = (int64_t*)malloc((n + 1) * 8);
  int64_t* pn_off = (int64_t*)calloc(n + 2, 8);   /* parent neighbor lists */
  uint64_t* pn = NULL;
  memcpy(parent_ids, nodes, n * 8);
  memcpy(cur, nodes, n * 8);
  for (int32_t s = 0; s < walk_len; ++s) {
    const int32_t* et = edge_types + s * k;
    int64_t tot = eo_get_full_neighbor(g, (const uint64_t*)cur, n, et, k, NULL,
                                       NULL, NULL, NULL);
    int32_t* idx = (int32_t*)malloc((2 * n + 2) * 4);
    uint64_t* cn = (uint64_t*)malloc((tot + 1) * 8);
    float* w = (float*)malloc((tot + 1) * 4);
    int32_t* tt = (int32_t*)malloc((tot + 1) * 4);
    float* sums = (float*)malloc((tot + 1) * 4);
    eo_get_full_neighbor(g, (const uint64_t*)cur, n, et, k, idx, cn, w, tt);
    int64_t* next = (int64_t*)malloc((n + 1) * 8);
    for (int64_t i = 0; i < n; ++i) {
      int64_t b = idx[2 * i], e = idx[2 * i + 1];
      int64_t sample_id = default_node;
      if (e > b) {
        int64_t parent_id = parent_ids[i];
        const int64_t* c = (const int64_t*)cn + b;   /* ids read as int64 */
        const int64_t* pp = pn ? (const int64_t*)pn + pn_off[i] : NULL;
        int64_t np = pn ? pn_off[i + 1] - pn_off[i] : 0;
        float* wi = w + b;
        int64_t nc = e - b, j = 0, kk = 0;
        while (j < nc && kk < np) {
          if (c[j] < pp[kk]) {
            if (c[j] != parent_id) wi[j] /= q; else wi[j] /= p;
            ++j;
          } else if (c[j] == pp[kk]) {
            ++kk; ++j;
          } else {
            ++kk;
          }
        }
        while (j < nc) {
          if (c[j] != parent_id) wi[j] /= q; else wi[j] /= p;
          ++j;
        }
        /* CompactWeightedCollection::Init + Sample (:116-119) */
        float acc = 0;
        for (int64_t x = 0; x < nc; ++x) { acc += wi[x]; sums[x] = acc; }
        eo_rng_ctx rng = {seed, call_id + (uint32_t)s, EO_DOMAIN_WALK,
                          (uint64_t)i, 0};
        int64_t mid = eo_random_select(sums, 0, (uint64_t)(nc - 1),
                                       eo_next_uniform(&rng));
        sample_id = c[mid];
      }
      out[i * L + s + 1] = sample_id;
      next[i] = sample_id;
    }
    /* parent_neighbors_ = neighbors; parent_ids_ = this step's nodes */
    free(pn);
    pn = cn;
    for (int64_t i = 0; i < n; ++i) { pn_off[i] = idx[2 * i]; }
    pn_off[n] = tot;
    memcpy(parent_ids, cur, n * 8);
    memcpy(cur, next, n * 8);
    free(next); free(idx); free(w); free(tt); free(sums);
  }
  free(pn); free(pn_off); free(parent_ids); free(cur);
  return 0;
}

/* GenPair (tf_euler/kernels/gen_pair_op.cc:42-95). */
int64_t eo_gen_pair_count(int64_t path_len, int32_t left, int32_t right) {
  int64_t pair_count = path_len * (left + right);
  for (int i = left, j = 0; i > 0 && j < path_len; --i, ++j) pair_count -= i;
  for (int i = right, j = 0; i > 0 && j < path_len; --i, ++j) pair_count -= i;
  return pair_count;
}

void eo_gen_pair(const int64_t* paths, int64_t batch, int64_t path_len,
                 int32_t left, int32_t right, int64_t* out) {
  int64_t pc = eo_gen_pair_count(path_len, left, right);
  for (int64_t i = 0; i < batch; ++i) {
    const int64_t* path = paths + i * path_len;
    int64_t* o = out + i * pc * 2;
    for (int64_t j = 0; j < path_len; ++j) {
      int k = 0;
      while ((j - k - 1) >= 0 && k < left) {
        *o++ = path[j]; *o++ = path[j - k - 1]; ++k;
      }
      k = 0;
      while ((j + k + 1) < path_len && k < right) {
        *o++ = path[j]; *o++ = path[j + k + 1]; ++k;
      }
    }
  }
}

/* ------------------------------------------------ message passing ops */

/* ScatterAddOp (tf_euler/kernels/scatter_op.cc:32-56): zero init, adds in
 * input order (fp32, order matters). */
void eo_scatter_add(const float* updates, const int32_t* indices, int64_t e,
                    int64_t d, int32_t size, float* out) {
  for (int64_t i = 0; i < (int64_t)size * d; ++i) out[i] = 0;
  for (int64_t i = 0; i < e; ++i)
    for (int64_t j = 0; j < d; ++j)
      out[(int64_t)indices[i] * d + j] += updates[i * d + j];
}

/* ScatterMaxOp (scatter_op.cc:64-92): init -1e9 (Q11). */
void eo_scatter_max(const float* updates, const int32_t* indices, int64_t e,
                    int64_t d, int32_t size, float* out) {
  for (int64_t i = 0; i < (int64_t)size * d; ++i) out[i] = (float)-1e9;
  for (int64_t i = 0; i < e; ++i)
    for (int64_t j = 0; j < d; ++j) {
      int64_t o = (int64_t)indices[i] * d + j;
      if (updates[i * d + j] > out[o]) out[o] = updates[i * d + j];
    }
}

/* GatherOp (tf_euler/kernels/gather_op.cc:31-52). */
void eo_gather(const float* params, const int32_t* indices, int64_t e,
               int64_t d, float* out) {
  for (int64_t i = 0; i < e; ++i)
    memcpy(out + i * d, params + (int64_t)indices[i] * d, (size_t)d * 4);
}

/* ----------------------------------------------------------- shard ops */

/* IDSplit::GetShardId (core/kernels/id_split_op.cc:46-49). */
int32_t eo_shard_of(uint64_t id, int32_t partitions, int32_t shards) {
  return (int32_t)((id % (uint64_t)partitions) % (uint64_t)shards);
}

/* IDSplit::Compute node-id branch (id_split_op.cc:57-98): stable bucket by
 * owner; merge_idx = original positions. shard_off is [shards+1]. */
void eo_id_split(const uint64_t* ids, int64_t n, int32_t partitions,
                 int32_t shards, int64_t* shard_off, uint64_t* shard_ids,
                 int32_t* merge_idx) {
  for (int32_t s = 0; s <= shards; ++s) shard_off[s] = 0;
  for (int64_t i = 0; i < n; ++i)
    shard_off[eo_shard_of(ids[i], partitions, shards) + 1]++;
  for (int32_t s = 0; s < shards; ++s) shard_off[s + 1] += shard_off[s];
  int64_t* fill = (int64_t*)calloc(shards + 1, 8);
  for (int64_t i = 0; i < n; ++i) {
    int32_t s = eo_shard_of(ids[i], partitions, shards);
    int64_t p = shard_off[s] + fill[s]++;
    shard_ids[p] = ids[i];
    merge_idx[p] = (int32_t)i;
  }
  free(fill);
}

/* SampleNodeSplit (core/kernels/sample_node_split_op.cc:57-85) for a single
 * (already summed) type weight per shard; shard_weight[shards] = total. */
void eo_sample_node_split(uint64_t seed, uint32_t call_id, int32_t count,
                          const float* shard_weight, int32_t shards,
                          int32_t* split_cnt) {
  int32_t remain = count;
  int32_t nz[1024]; int32_t nnz = 0;
  for (int32_t i = 0; i < shards; ++i) {
    float sw0 = 0, sw1 = 0;
    sw0 += shard_weight[i];
    sw1 += shard_weight[shards];
    split_cnt[i] = (int32_t)floor(count * sw0 / sw1);
    if (sw0 > 0 && nnz < 1024) nz[nnz++] = i;
    remain -= split_cnt[i];
  }
  eo_rng_ctx rng = {seed, call_id, EO_DOMAIN_SPLIT, 0, 0};
  for (; remain > 0; --remain) {
    int32_t z = nz[(size_t)floor(eo_next_uniform(&rng) * (size_t)nnz)];
    split_cnt[z] += 1;
  }
}

/* -------------------------------------------------------- CPU baseline */

static double eo_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double eo_bench_fanout(const eo_graph* g, uint64_t seed,
                       const uint64_t* roots, int64_t batch, int32_t iters,
                       const int32_t* counts, int32_t hops, int32_t threads,
                       int64_t* edges) {
  int64_t total = 0;
  int32_t et = 0;
  double t0 = eo_now();
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) reduction(+ : total) schedule(dynamic, 1)
#endif
  for (int32_t it = 0; it < iters; ++it) {
    int64_t m = batch;
    uint64_t* frontier = (uint64_t*)malloc((m + 1) * 8);
    memcpy(frontier, roots + (int64_t)it * batch, m * 8);
    for (int32_t h = 0; h < hops; ++h) {
      int64_t tot = m * counts[h];
      uint64_t* cid = (uint64_t*)malloc((tot + 1) * 8);
      float* cw = (float*)malloc((tot + 1) * 4);
      int32_t* ct = (int32_t*)malloc((tot + 1) * 4);
      eo_sample_neighbor_core(g, seed, (uint32_t)(it * hops + h), frontier, m,
                              &et, 1, counts[h], NULL, cid, cw, ct);
      total += tot;
      free(frontier); free(cw); free(ct);
      frontier = cid; m = tot;
    }
    free(frontier);
  }
  double t1 = eo_now();
  (void)threads;
  *edges = total;
  return t1 - t0;
}


/* tf_euler/kernels/get_dense_feature_op.cc:63-125: outputs are zero filled
 * (:70-72); the GQL values() step returns, per node, the (begin, end) offsets
 * of feature `fid` in the node's float_features_ (GET_NODE_FEATURE,
 * core/graph/node.cc:330-352: fid out of range -> length 0; unknown node ->
 * empty); the callback copies end - begin values to row j (:113-120). */
int eo_get_dense_feature(const eo_graph* g, const eo_features* f,
                         const uint64_t* ids, int64_t n, int32_t fid,
                         int32_t dim, float* out) {
  for (int64_t i = 0; i < n * (int64_t)dim; ++i) out[i] = 0.0f;
  for (int64_t j = 0; j < n; ++j) {
    int64_t row = eo_graph_find_row(g, ids[j]);
    if (row < 0) continue;
    if (fid < 0 || fid >= f->n_float) continue;
    const int32_t* idx = f->feat_idx + row * f->n_float;
    int32_t pre = fid == 0 ? 0 : idx[fid - 1];
    int32_t now = idx[fid];
    if (now - pre > dim) return -2;
    const float* src = f->feat_val + f->feat_ptr[row] + pre;
    for (int32_t c = 0; c < now - pre; ++c) out[j * (int64_t)dim + c] = src[c];
  }
  return 0;
}


/* core/kernels/get_neighbor_op.cc:117-168.  order_by sorts every row with
 * std::sort and `a <= b` (asc) / `!(a <= b)` (desc) on the key: for distinct
 * keys that is ascending / descending order; equal keys have no defined order
 * there (non-strict comparator), here they keep storage order (insertion sort,
 * stable).  limit k truncates every row to its first k entries. */
int64_t eo_neighbor_post_process(int64_t n, int32_t* idx, uint64_t* ids, float* w,
                                 int32_t* t, int32_t order_by, int32_t desc,
                                 int64_t limit) {
  int64_t out = 0;
  for (int64_t i = 0; i < n; ++i) {
    int64_t b = idx[2 * i], e = idx[2 * i + 1];
    if (order_by == 1 || order_by == 2) {
      for (int64_t x = b + 1; x < e; ++x) {
        uint64_t ki = ids[x]; float kw = w[x]; int32_t kt = t[x];
        int64_t y = x - 1;
        while (y >= b) {
          int before;   /* does element y stay before the moving element? */
          if (order_by == 1) before = desc ? ids[y] >= ki : ids[y] <= ki;
          else before = desc ? w[y] >= kw : w[y] <= kw;
          if (before) break;
          ids[y + 1] = ids[y]; w[y + 1] = w[y]; t[y + 1] = t[y];
          --y;
        }
        ids[y + 1] = ki; w[y + 1] = kw; t[y + 1] = kt;
      }
    }
    int64_t len = e - b;
    if (limit >= 0 && len > limit) len = limit;
    for (int64_t x = 0; x < len; ++x) {
      ids[out + x] = ids[b + x]; w[out + x] = w[b + x]; t[out + x] = t[b + x];
    }
    idx[2 * i] = (int32_t)out;
    idx[2 * i + 1] = (int32_t)(out + len);
    out += len;
  }
  return out;
}

void eo_neighbor_to_dense(int64_t n, const int32_t* idx, const uint64_t* ids,
                          const float* w, const int32_t* t, int32_t k,
                          int64_t default_node, int64_t* out_id, float* out_w,
                          int32_t* out_t) {
  for (int64_t i = 0; i < n * (int64_t)k; ++i) {
    out_id[i] = default_node; out_w[i] = 0.0f; out_t[i] = -1;
  }
  for (int64_t i = 0; i < n; ++i) {
    int64_t b = idx[2 * i], e = idx[2 * i + 1];
    for (int64_t j = b; j < e && j - b < k; ++j) {
      out_id[i * k + j - b] = (int64_t)ids[j];
      out_w[i * k + j - b] = w[j];
      out_t[i * k + j - b] = t[j];
    }
  }
}

/* ------------------------------------------------ layerwise sampling ops
 * The DAG of `sampleLNB(edge_types, n, m, default_node)` (parser/translator.cc:
 * 338-386,489-527).  The `sqrt` variant (API_LOCAL_SAMPLE_L) is not restated:
 * its candidate order is std::unordered_map<std::string,...> iteration order. */

/* API_GET_EDGE_SUM_WEIGHT (core/kernels/get_edge_sum_weight_op.cc:51-62):
 * `float sum_weight = 0; for (iw : GetFullNeighbor({root}, edge_types)[0])
 * sum_weight += weight`. */
void eo_get_edge_sum_weight(const eo_graph* g, const uint64_t* ids, int64_t n,
                            const int32_t* edge_types, int32_t k, float* out_w) {
  const int32_t T = g->n_types;
  for (int64_t i = 0; i < n; ++i) {
    float sum = 0;
    int64_t row = eo_graph_find_row(g, ids[i]);
    if (row >= 0) {
      const int32_t* gi = g->type_end + row * T;
      const float* nw = g->prefix_w + g->row_ptr[row];
      for (int32_t a = 0; a < k; ++a) {
        int32_t et = edge_types[a];
        if (et >= 0 && et < T) {
          int32_t b = et == 0 ? 0 : gi[et - 1];
          for (int32_t j = b; j < gi[et]; ++j) {
            float pre = j == 0 ? 0 : nw[j - 1];
            float w = nw[j] - pre;
            sum += w;
          }
        }
      }
    }
    out_w[i] = sum;
  }
}

/* API_SAMPLE_ROOT (core/kernels/sample_root_op.cc:42-86): per batch row a
 * FastWeightedCollection (fast_weighted_collection.h:55-75: f32 sum, weights
 * divided by it, AliasMethod::Init) over the row's n roots; m draws
 * (AliasMethod::Next: 2 uniforms each); zero sum -> default_node.
 * RNG: domain ROOT, stream = batch row. */
void eo_sample_root(uint64_t seed, uint32_t call_id, const uint64_t* roots,
                    const float* weights, int64_t batch, int32_t n, int32_t m,
                    int64_t default_node, uint64_t* out) {
  float* norm = (float*)malloc((size_t)(n + 1) * 4);
  float* prob = (float*)malloc((size_t)(n + 1) * 4);
  int64_t* alias = (int64_t*)malloc((size_t)(n + 1) * 8);
  for (int64_t b = 0; b < batch; ++b) {
    const float* w = weights + b * n;
    float sum = 0.0f;
    for (int32_t i = 0; i < n; ++i) sum += w[i];
    if (sum == 0) {
      for (int32_t j = 0; j < m; ++j) out[b * m + j] = (uint64_t)default_node;
      continue;
    }
    for (int32_t i = 0; i < n; ++i) norm[i] = w[i] / sum;
    eo_alias_init(norm, n, prob, alias);
    eo_rng_ctx rng = {seed, call_id, EO_DOMAIN_ROOT, (uint64_t)b, 0};
    for (int32_t j = 0; j < m; ++j)
      out[b * m + j] = roots[b * n + eo_alias_next(prob, alias, n, &rng)];
  }
  free(norm); free(prob); free(alias);
}

/* API_SAMPLE_L (core/kernels/sample_layer_op.cc:54-70): one
 * SampleNeighbor({root}, edge_types, 1) per position; empty -> (default_node,
 * 0, 0).  RNG: domain LAYER, stream = position. */
void eo_sample_layer(const eo_graph* g, uint64_t seed, uint32_t call_id,
                     const uint64_t* roots, int64_t n, const int32_t* edge_types,
                     int32_t k, int64_t default_node, uint64_t* out_id,
                     float* out_w, int32_t* out_t) {
  eo_sample_layer_at(g, seed, call_id, roots, 0, n, edge_types, k, default_node, out_id,
                     out_w, out_t);
}

/* ... with explicit RNG streams (pos == NULL: the index): what a shard of the
 * multi-GPU path computes for the positions its requester sent along. */
void eo_sample_layer_at(const eo_graph* g, uint64_t seed, uint32_t call_id,
                        const uint64_t* roots, const int64_t* pos, int64_t n,
                        const int32_t* edge_types, int32_t k, int64_t default_node,
                        uint64_t* out_id, float* out_w, int32_t* out_t) {
  for (int64_t i = 0; i < n; ++i) {
    int32_t got = 0;
    int64_t row = eo_graph_find_row(g, roots[i]);
    if (row >= 0) {
      eo_rng_ctx rng = {seed, call_id, EO_DOMAIN_LAYER, (uint64_t)(pos ? pos[i] : i), 0};
      got = eo_sample_row(g, row, edge_types, k, 1, &rng, out_id + i, out_w + i,
                          out_t + i);
    }
    if (got == 0) {
      out_id[i] = (uint64_t)default_node; out_w[i] = 0; out_t[i] = 0;
    }
  }
}

/* EdgeExist(EdgeId(src, dst, type)) (core/api/api.cc:46-48) answered from the
 * adjacency rows (the Edge map holds the same triples; checked against the
 * reference's loaded Edge records in tests/test_oracle_vs_ref.py). */
static int eo_edge_exist(const eo_graph* g, uint64_t src, uint64_t dst,
                         int32_t type) {
  const int32_t T = g->n_types;
  if (type < 0 || type >= T) return 0;
  int64_t row = eo_graph_find_row(g, src);
  if (row < 0) return 0;
  const int32_t* gi = g->type_end + row * T;
  const uint64_t* nbr = g->nbr + g->row_ptr[row];
  for (int32_t j = type == 0 ? 0 : gi[type - 1]; j < gi[type]; ++j)
    if (nbr[j] == dst) return 1;
  return 0;
}

/* API_SPARSE_GEN_ADJ + API_SPARSE_GET_ADJ (core/kernels/sparse_gen_adj_op.cc:
 * 52-61, sparse_get_adj_op.cc:55-91): root r of batch row r / n keeps the
 * candidates l_nb[b*m .. b*m+m) it has an edge of a listed type to.  Pass
 * out_id == NULL to size.  Returns the total. */
int64_t eo_sparse_get_adj(const eo_graph* g, const uint64_t* roots,
                          const uint64_t* l_nb, int64_t batch, int32_t n,
                          int32_t m, const int32_t* edge_types, int32_t k,
                          int32_t* idx, uint64_t* out_id) {
  int64_t off = 0;
  for (int64_t r = 0; r < batch * n; ++r) {
    int64_t b = r / n;
    if (idx) idx[2 * r] = (int32_t)off;
    for (int32_t j = 0; j < m; ++j) {
      uint64_t nb = l_nb[b * m + j];
      int exist = 0;
      for (int32_t a = 0; a < k; ++a)
        exist = exist || eo_edge_exist(g, roots[r], nb, edge_types[a]);
      if (exist) {
        if (out_id) out_id[off] = nb;
        ++off;
      }
    }
    if (idx) idx[2 * r + 1] = (int32_t)off;
  }
  return off;
}

/* The sparse-tensor assembly shared by the TF kernels SparseGetAdj
 * (tf_euler/kernels/sparse_get_adj_op.cc:92-124) and
 * SampleNeighborLayerwiseWithAdj (sample_neighbor_layerwise_with_adj_op.cc:
 * 112-140), from the API_SPARSE_GET_ADJ result (idx, vals): per batch row a
 * std::set of (src id, nb id) pairs, then for every (j, c) a 1 where the pair
 * is in the set, and a 0 at (j, c) = (n-1, m-1) otherwise.  indices [nnz,3],
 * values [nnz], shape[3] (= max index + 1 per dimension, SparseTensorBuilder).
 * Pass indices == NULL to size. */
int64_t eo_adj_to_sparse(const uint64_t* nodes, const uint64_t* nb_nodes,
                         int64_t batch, int32_t n, int32_t m, const int32_t* idx,
                         const uint64_t* vals, int64_t* indices, int64_t* values,
                         int64_t* shape) {
  int64_t nnz = 0;
  if (shape) shape[0] = shape[1] = shape[2] = 0;
  for (int64_t i = 0; i < batch; ++i) {
    for (int32_t j = 0; j < n; ++j) {
      uint64_t src = nodes[j + (int64_t)n * i];
      for (int32_t c = 0; c < m; ++c) {
        uint64_t dst = nb_nodes[c + (int64_t)m * i];
        /* relation_set.find((src, dst)): the set holds (nodes[j'], v) for every
         * j' of this batch row and every v in its result slice */
        int found = 0;
        for (int64_t jj = (int64_t)n * i; jj < (int64_t)n * (i + 1) && !found; ++jj) {
          if (nodes[jj] != src) continue;
          for (int32_t p = idx[2 * jj]; p < idx[2 * jj + 1]; ++p)
            if (vals[p] == dst) { found = 1; break; }
        }
        int emit = found || (j == n - 1 && c == m - 1);
        if (!emit) continue;
        if (indices) {
          indices[3 * nnz] = i; indices[3 * nnz + 1] = j; indices[3 * nnz + 2] = c;
          values[nnz] = found ? 1 : 0;
        }
        if (shape) {
          if (i + 1 > shape[0]) shape[0] = i + 1;
          if (j + 1 > shape[1]) shape[1] = j + 1;
          if (c + 1 > shape[2]) shape[2] = c + 1;
        }
        ++nnz;
      }
    }
  }
  return nnz;
}

/* TF GetSparseFeature for one feature (tf_euler/kernels/get_sparse_feature_op.cc:
 * 84-113): the GQL result "fea:2i" / "fea:2i+1" (idx pairs + values of slot fid,
 * GET_NODE_FEATURE, core/graph/node.cc:330-351) turned into SparseTensorBuilder
 * entries: node j with no value -> ((j, 0), default); else ((j, k), value k).
 * Pass indices == NULL to size.  shape[2] = max index + 1 per dimension. */
int64_t eo_get_sparse_feature(const eo_graph* g, const eo_u64_features* f,
                              const uint64_t* ids, int64_t n, int32_t fid,
                              int64_t default_value, int64_t* indices,
                              int64_t* values, int64_t* shape) {
  int64_t nnz = 0;
  if (shape) shape[0] = shape[1] = 0;
  for (int64_t j = 0; j < n; ++j) {
    int32_t len = 0;
    const uint64_t* src = 0;
    int64_t row = eo_graph_find_row(g, ids[j]);
    if (row >= 0 && fid >= 0 && fid < f->n_u64) {
      const int32_t* idx = f->feat_idx + row * f->n_u64;
      int32_t pre = fid == 0 ? 0 : idx[fid - 1];
      len = idx[fid] - pre;
      src = f->feat_val + f->feat_ptr[row] + pre;
    }
    int32_t emit = len > 0 ? len : 1;
    for (int32_t k = 0; k < emit; ++k) {
      if (indices) {
        indices[2 * nnz] = j; indices[2 * nnz + 1] = k;
        values[nnz] = len > 0 ? (int64_t)src[k] : default_value;
      }
      if (shape) {
        if (j + 1 > shape[0]) shape[0] = j + 1;
        if (k + 1 > shape[1]) shape[1] = k + 1;
      }
      ++nnz;
    }
  }
  return nnz;
}
