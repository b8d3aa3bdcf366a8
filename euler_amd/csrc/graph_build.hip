// Graph construction: host CSR -> HBM layout, synthetic power-law generator
// on device, global node sampler (alias tables), row export, and the graph
// life-cycle entry points of the C ABI.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>

#include "device_fns.h"
#include "wb_index.h"

namespace euler_gpu {


namespace {
thread_local std::string g_last_error;
std::mutex g_default_mu;
euler_gpu_graph* g_default_graph = nullptr;
}  // namespace

void SetError(const std::string& msg) { g_last_error = msg; }
int Fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

namespace {

struct GraphBuilder {
  std::unique_ptr<euler_gpu_graph> g{new euler_gpu_graph()};
  int rc = EULER_GPU_OK;

  template <typename T>
  T* Alloc(size_t count) {
    void* p = nullptr;
    const size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      rc = Fail(EULER_GPU_ENOMEM, std::string("hipMalloc(") +
                                      std::to_string(bytes) +
                                      "): " + hipGetErrorString(e));
      return nullptr;
    }
    g->allocations.push_back(p);
    g->bytes += (int64_t)bytes;
    return (T*)p;
  }

  template <typename T>
  T* Upload(const T* host, size_t count) {
    T* d = Alloc<T>(count);
    if (!d) return nullptr;
    if (count > 0) {
      hipError_t e = hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice);
      if (e != hipSuccess) {
        rc = Fail(EULER_GPU_EHIP, std::string("hipMemcpy H2D: ") +
                                      hipGetErrorString(e));
        return nullptr;
      }
    }
    return d;
  }
};

// ---- sampling index (EdgeBlock + skip levels), built from the flat arrays --
__global__ void BuildBlocksKernel(const float* __restrict__ pw,
                                  const uint64_t* __restrict__ nbr, int64_t E,
                                  int64_t n_blk, EdgeBlock* __restrict__ blk,
                                  float* __restrict__ skip1) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_blk) return;
  EdgeBlock b;
  const int64_t base = i * kEdgesPerBlock;
#pragma unroll
  for (int x = 0; x < kEdgesPerBlock; ++x) {
    const int64_t m = base + x < E ? base + x : E - 1;   // tail: repeat the last edge
    b.pw[x] = pw[m];
    b.nbr[x] = nbr[m];
  }
  b.prev_last = base > 0 ? pw[base - 1] : 0.f;
  b.pad = 0;
  blk[i] = b;
  skip1[i] = b.pw[kEdgesPerBlock - 1];
}

// upper[q] = lower[min(fan * q + fan - 1, n_lower - 1)]
__global__ void BuildPivotKernel(const float* __restrict__ lower, int64_t n_lower,
                                 int32_t fan, int64_t n_upper,
                                 float* __restrict__ upper) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_upper) return;
  int64_t x = q * fan + (fan - 1);
  if (x > n_lower - 1) x = n_lower - 1;
  upper[q] = lower[x];
}

int BuildPivotLevels(GraphBuilder* b) {
  GraphView& v = b->g->view;
  const int64_t E = v.n_edges;
  if (E >= (int64_t)0x7fffffff * 4 - 64)
    return Fail(EULER_GPU_EINVAL, "graph too large for 32-bit pivot indices");
  int64_t n[kPivotLevels + 1];
  int64_t total = 0;
  n[0] = E;
  for (int k = 1; k <= kPivotLevels; ++k) {
    const int fan = k == 1 ? 4 : 5;
    n[k] = (n[k - 1] + fan - 1) / fan;
    v.piv_off[k] = total;
    total += n[k] + 4;          // a 4-float window may run past the level's end
  }
  v.piv_off[0] = 0;
  float* piv = b->Alloc<float>((size_t)total + 8);
  if (b->rc != EULER_GPU_OK) return b->rc;
  EG_HIP(hipMemset(piv, 0, ((size_t)total + 8) * sizeof(float)));
  const int block = 256;
  const float* lower = v.prefix_w;
  for (int k = 1; k <= kPivotLevels && E > 0; ++k) {
    const int fan = k == 1 ? 4 : 5;
    hipLaunchKernelGGL(BuildPivotKernel, dim3((n[k] + block - 1) / block), dim3(block),
                       0, 0, lower, n[k - 1], fan, n[k], piv + v.piv_off[k]);
    lower = piv + v.piv_off[k];
  }
  EG_HIP(hipGetLastError());
  EG_HIP(hipDeviceSynchronize());
  v.pivots = piv;
  return EULER_GPU_OK;
}

// flag[0] stays 1 iff every non-empty row's type_prefix[0] == its last sum (T == 1)
__global__ void VerifyTotalsKernel(GraphView g, int32_t* flag) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= g.n_rows) return;
  const uint4 q = *reinterpret_cast<const uint4*>(g.row_meta + row * 16);
  const int64_t row_ptr = (int64_t)(((uint64_t)q.y << 32) | q.x);
  const int32_t deg = (int32_t)q.z;
  if (deg > 0 && __float_as_uint(g.prefix_w[row_ptr + deg - 1]) != q.w) flag[0] = 0;
}

int VerifyTotals(GraphBuilder* b) {
  GraphView& v = b->g->view;
  v.total_in_meta = 0;
  if (v.T != 1 || v.n_rows == 0) return EULER_GPU_OK;
  int32_t* flag = nullptr;
  EG_HIP(hipMalloc((void**)&flag, 16));
  const int32_t one = 1;
  EG_HIP(hipMemcpy(flag, &one, 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(VerifyTotalsKernel, dim3((v.n_rows + 255) / 256), dim3(256), 0, 0, v,
                     flag);
  int32_t res = 0;
  EG_HIP(hipMemcpy(&res, flag, 4, hipMemcpyDeviceToHost));
  EG_HIP(hipFree(flag));
  v.total_in_meta = res;
  return EULER_GPU_OK;
}

int BuildBlockedIndex(GraphBuilder* b);

// At creation: the pivot levels over the flat arrays (1.25 bytes per edge; variant 5 and every
// search's last resort).  The weight-bucket index and the EdgeBlock copy with its block pivots
// are built by the first sampling call (sample_kernels.hip: SamplingView) - and the EdgeBlocks
// (13 bytes per edge) only for graphs the weight-bucket index does not serve.
int BuildSearchIndex(GraphBuilder* b) {
  int rc = VerifyTotals(b);
  if (rc == EULER_GPU_OK) rc = BuildPivotLevels(b);
  return rc;
}

int BuildBlockedIndex(GraphBuilder* b) {
  GraphView& v = b->g->view;
  const int64_t E = v.n_edges;
  v.n_blk = (E + kEdgesPerBlock - 1) / kEdgesPerBlock;
  if (E >= (int64_t)kEdgesPerBlock * 0x7fffff00LL)
    return Fail(EULER_GPU_EINVAL, "graph too large for 32-bit block indices");
  EdgeBlock* blk = b->Alloc<EdgeBlock>((size_t)v.n_blk);
  float* s1 = b->Alloc<float>((size_t)v.n_blk + 8);   // windows may run 3 past the end
  if (b->rc != EULER_GPU_OK) return b->rc;
  const int block = 256;
  if (E > 0) {
    hipLaunchKernelGGL(BuildBlocksKernel, dim3((v.n_blk + block - 1) / block),
                       dim3(block), 0, 0, v.prefix_w, v.nbr, E, v.n_blk, blk, s1);
    EG_HIP(hipGetLastError());
    EG_HIP(hipDeviceSynchronize());
  }
  // block pivot levels over skip1 (fanout 5)
  {
    int64_t n[kPivotLevels + 1];
    int64_t total = 0;
    n[1] = v.n_blk;
    v.bpiv_off[0] = v.bpiv_off[1] = 0;
    for (int k = 2; k <= kPivotLevels; ++k) {
      n[k] = (n[k - 1] + 4) / 5;
      v.bpiv_off[k] = total;
      total += n[k] + 4;
    }
    float* bp = b->Alloc<float>((size_t)total + 8);
    if (b->rc != EULER_GPU_OK) return b->rc;
    EG_HIP(hipMemset(bp, 0, ((size_t)total + 8) * sizeof(float)));
    const float* lower = s1;
    for (int k = 2; k <= kPivotLevels && E > 0; ++k) {
      hipLaunchKernelGGL(BuildPivotKernel, dim3((n[k] + block - 1) / block), dim3(block),
                         0, 0, lower, n[k - 1], 5, n[k], bp + v.bpiv_off[k]);
      lower = bp + v.bpiv_off[k];
    }
    EG_HIP(hipGetLastError());
    EG_HIP(hipDeviceSynchronize());
    v.bpiv = bp;
  }
  v.blk = blk; v.skip1 = s1;
  return EULER_GPU_OK;
}

std::mutex g_blocked_mu;

// Stream-ordered scratch (hipMallocAsync) is used per call by several entry
// points; keep freed blocks in the pool instead of returning them to the
// driver at every synchronisation (the default threshold is 0).
void KeepPoolMemory(int device) {
  hipMemPool_t pool;
  if (hipDeviceGetDefaultMemPool(&pool, device) == hipSuccess) {
    uint64_t keep = ~0ULL;
    (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
  }
}

void DestroyGraph(euler_gpu_graph* g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  for (void* p : g->allocations) (void)hipFree(p);
  for (auto& kv : g->ws) (void)hipFree(kv.second.first);
  for (auto& kv : g->flow_tables) (void)hipFree(kv.second.p);
  delete g;
}

// AliasMethod::Init (euler/common/alias_method.cc:23-63), restated for the
// table build: LIFO small/large stacks, `avg` in double, weights updated in
// float, prob in float.  Element order is the caller's (Q7).
void AliasBuild(const std::vector<float>& weights, std::vector<float>* prob,
                std::vector<int64_t>* alias) {
  const size_t n = weights.size();
  prob->assign(n, 0.f);
  alias->assign(n, 0);
  std::vector<int64_t> small, large;
  std::vector<float> w(weights);
  const double avg = 1 / static_cast<double>(n);
  for (size_t i = 0; i < n; ++i) {
    if (w[i] > avg) large.push_back((int64_t)i); else small.push_back((int64_t)i);
  }
  while (!large.empty() && !small.empty()) {
    const int64_t less = small.back(); small.pop_back();
    const int64_t more = large.back(); large.pop_back();
    (*prob)[less] = w[less] * n;
    (*alias)[less] = more;
    w[more] = w[more] + w[less] - avg;
    if (w[more] > avg) large.push_back(more); else small.push_back(more);
  }
  while (!small.empty()) { (*prob)[small.back()] = 1.0; small.pop_back(); }
  while (!large.empty()) { (*prob)[large.back()] = 1.0; large.pop_back(); }
}

// Graph::BuildGlobalSampler (core/graph/graph.cc:333-370) +
// FastWeightedCollection::Init (common/fast_weighted_collection.h:55-75).
int BuildNodeSampler(GraphBuilder* b, const std::vector<uint64_t>& ids,
                     const std::vector<int32_t>& types,
                     const std::vector<float>& weights, int32_t n_types) {
  if (n_types <= 0 || n_types > kMaxNodeTypes)
    return Fail(EULER_GPU_EINVAL, "node sampler: 1 .. 32 node types");
  const size_t n = ids.size();
  // Everything is validated and built on the side; the graph's sampler is replaced only when the
  // new table is on the device (a failed rebuild must leave the old sampler intact).
  for (size_t i = 0; i < n; ++i) {
    if (types[i] < 0 || types[i] >= n_types)
      return Fail(EULER_GPU_EINVAL, "node sampler: node type out of range");
    if (!(weights[i] >= 0.f) || !std::isfinite(weights[i]))
      return Fail(EULER_GPU_EINVAL, "node sampler: node weights must be finite and >= 0");
  }
  NodeSamplerView s;
  std::memset(&s, 0, sizeof(s));
  s.n_types = n_types;
  std::vector<std::vector<uint64_t>> tid(n_types);
  std::vector<std::vector<float>> tw(n_types);
  std::vector<float> sums(n_types, 0.f);
  for (size_t i = 0; i < n; ++i) {
    const int32_t t = types[i];
    tid[t].push_back(ids[i]);
    tw[t].push_back(weights[i]);
    sums[t] += weights[i];
  }
  for (int32_t t = 0; t < n_types; ++t)
    if (!std::isfinite(sums[t]))
      return Fail(EULER_GPU_EINVAL, "node sampler: a node type's weights sum to infinity");
  std::vector<AliasEntry> entries(n);
  size_t off = 0;
  for (int32_t t = 0; t < n_types; ++t) {
    s.type_off[t] = (int64_t)off;
    std::vector<float>& w = tw[t];
    // a type whose nodes all weigh 0 is never drawn (its type sum is 0: Graph::SampleNode returns
    // nothing for it, graph.cc:221-245); its table is filled evenly instead of with 0 / 0
    const bool dead = !w.empty() && !(sums[t] > 0.f);
    for (auto& x : w) x = dead ? 1.0f / (float)w.size() : x / sums[t];   // graph.cc:355-358
    float sum = 0.f;
    for (auto x : w) sum += x;                       // FWC::Init sum_weight_
    std::vector<float> norm(w);
    for (auto& x : norm) x /= sum;
    std::vector<float> prob;
    std::vector<int64_t> alias;
    AliasBuild(norm, &prob, &alias);
    for (size_t i = 0; i < w.size(); ++i) {
      AliasEntry e{};
      e.id_self = tid[t][i];
      e.id_alias = tid[t][(size_t)alias[i]];
      e.prob = prob[i];
      entries[off + i] = e;
    }
    s.type_sum[t] = sums[t];
    s.sampler_sum[t] = (w.empty() || dead) ? 0.f : sum;
    off += w.size();
  }
  s.type_off[n_types] = (int64_t)off;
  // node_type_collection_.Init(node_type_ids, node_weight_sums_)
  float tsum = 0.f;
  for (int32_t t = 0; t < n_types; ++t) tsum += sums[t];
  s.tc_sum = tsum;
  std::vector<float> tnorm(sums);
  for (auto& x : tnorm) x = tsum > 0.f ? x / tsum : 1.0f / (float)n_types;
  std::vector<float> tprob;
  std::vector<int64_t> talias;
  AliasBuild(tnorm, &tprob, &talias);
  for (int32_t t = 0; t < n_types; ++t) {
    s.tc_prob[t] = tprob[t];
    s.tc_alias[t] = (int32_t)talias[t];
  }
  s.entries = b->Upload(entries.data(), entries.size());
  if (!s.entries) return b->rc;
  b->g->sampler = s;
  b->g->has_sampler = true;
  b->g->n_node_types = n_types;
  b->g->node_weight_sums = sums;
  return EULER_GPU_OK;
}

}  // namespace

// ------------------------------------------------------------------------
// Host CSR -> device graph (optionally only the rows a shard owns).
// ------------------------------------------------------------------------
int BuildGraphFromHost(const euler_gpu_host_csr* c, int device,
                       int32_t partitions, int32_t shard_index, int32_t shards,
                       euler_gpu_graph** out) {
  if (!c || !out) return Fail(EULER_GPU_EINVAL, "graph_create: null argument");
  if (c->n_rows < 0 || c->n_edge_types <= 0 || c->n_edge_types > kMaxListedTypes)
    return Fail(EULER_GPU_EINVAL, "graph_create: need 1..32 edge types");
  if (shards <= 0 || shard_index < 0 || shard_index >= shards || partitions <= 0)
    return Fail(EULER_GPU_EINVAL, "graph_create: bad shard arguments");
  if (c->n_rows > 0 && (!c->row_id || !c->row_ptr || !c->type_end ||
                        !c->type_prefix))
    return Fail(EULER_GPU_EINVAL, "graph_create: null CSR array");
  EG_HIP(hipSetDevice(device));
  const int32_t T = c->n_edge_types;
  // rows kept by this shard
  std::vector<int64_t> keep;
  keep.reserve((size_t)c->n_rows / shards + 1);
  for (int64_t r = 0; r < c->n_rows; ++r) {
    const uint64_t id = c->row_id[r];
    if ((int32_t)((id % (uint64_t)partitions) % (uint64_t)shards) == shard_index)
      keep.push_back(r);
  }
  const int64_t n = (int64_t)keep.size();
  KeepPoolMemory(device);
  GraphBuilder b;
  b.g->device = device;
  GraphView& v = b.g->view;
  v.T = T;
  v.meta_stride = 8 + 8 * T;
  v.n_rows = n;
  // gather rows
  std::vector<uint64_t> row_id(n);
  std::vector<uint8_t> meta((size_t)n * v.meta_stride + 16, 0);
  int64_t E = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t r = keep[i];
    E += c->row_ptr[r + 1] - c->row_ptr[r];
  }
  std::vector<uint64_t> nbr((size_t)E);
  std::vector<float> pw((size_t)E);
  int64_t off = 0;
  bool zero_nbr = false;
  bool monotone = true;
  bool uniform = true;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t r = keep[i];
    row_id[i] = c->row_id[r];
    const int64_t b0 = c->row_ptr[r], deg = c->row_ptr[r + 1] - b0;
    if (deg < 0 || c->type_end[r * T + T - 1] != deg)
      return Fail(EULER_GPU_EINVAL, "graph_create: type_end does not match row_ptr");
    uint8_t* rec = meta.data() + (size_t)i * v.meta_stride;
    std::memcpy(rec, &off, 8);
    std::memcpy(rec + 8, c->type_end + r * T, 4 * T);
    std::memcpy(rec + 8 + 4 * T, c->type_prefix + r * T, 4 * T);
    if (deg > 0) {
      std::memcpy(nbr.data() + off, c->nbr + b0, (size_t)deg * 8);
      std::memcpy(pw.data() + off, c->prefix_w + b0, (size_t)deg * 4);
      float prev = 0.f;
      for (int64_t j = 0; j < deg; ++j) {
        zero_nbr |= c->nbr[b0 + j] == 0;
        const float cur = c->prefix_w[b0 + j];
        monotone &= cur >= prev;           // false for NaN too
        uniform &= cur == (float)(j + 1) && j + 1 < (1 << 24);
        prev = cur;
      }
    }
    off += deg;
  }
  v.n_edges = E;
  v.has_zero_nbr = zero_nbr ? 1 : 0;
  v.monotone = monotone ? 1 : 0;
  v.uniform_w = (uniform && monotone && E > 0) ? 1 : 0;   // (j + 1 is exact in f32 below 2^24)
  v.row_meta = b.Upload(meta.data(), meta.size());
  v.nbr = b.Upload(nbr.data(), nbr.size());
  v.prefix_w = b.Upload(pw.data(), pw.size());
  v.row_id = b.Upload(row_id.data(), row_id.size());
  if (b.rc != EULER_GPU_OK) { DestroyGraph(b.g.release()); return b.rc; }
  // id map: strided identity when the ids allow it, else a hash table
  bool identity = n > 0;
  uint64_t base = n > 0 ? row_id[0] : 0, stride = 1;
  if (n > 1) {
    stride = row_id[1] - row_id[0];
    if (row_id[1] <= row_id[0]) identity = false;
  }
  for (int64_t i = 0; identity && i < n; ++i)
    identity = row_id[i] == base + (uint64_t)i * stride;
  for (int64_t i = 0; i < n; ++i)
    if (row_id[i] > b.g->max_id) b.g->max_id = row_id[i];
  if (identity) {
    v.map_mode = 0; v.id_base = base; v.id_stride = stride;
  } else {
    uint64_t cap = 16;
    while (cap < (uint64_t)n * 2 + 1) cap <<= 1;
    std::vector<uint64_t> slots(2 * cap);
    for (uint64_t i = 0; i < cap; ++i) { slots[2 * i] = 0; slots[2 * i + 1] = ~0ULL; }
    for (int64_t i = 0; i < n; ++i) {
      uint64_t h = Mix64(row_id[i]) & (cap - 1);
      // duplicate ids: the later row wins (Graph::AddNode overwrites,
      // core/graph/graph.cc:162-166)
      while ((int64_t)slots[2 * h + 1] >= 0 && slots[2 * h] != row_id[i])
        h = (h + 1) & (cap - 1);
      slots[2 * h] = row_id[i];
      slots[2 * h + 1] = (uint64_t)i;
    }
    v.map_mode = 1; v.hash_mask = cap - 1; v.id_base = 0; v.id_stride = 1;
    v.hash_slots = b.Upload(slots.data(), slots.size());
    if (b.rc != EULER_GPU_OK) { DestroyGraph(b.g.release()); return b.rc; }
  }
  {
    int rc = BuildSearchIndex(&b);
    if (rc != EULER_GPU_OK) { DestroyGraph(b.g.release()); return rc; }
  }
  // dense float features of the kept rows
  if (c->n_float_features > 0) {
    if (!c->feat_ptr || !c->feat_idx || (c->feat_ptr[c->n_rows] > 0 && !c->feat_val)) {
      DestroyGraph(b.g.release());
      return Fail(EULER_GPU_EINVAL, "graph_create: null feature array");
    }
    const int32_t F = c->n_float_features;
    std::vector<int64_t> fptr((size_t)n + 1, 0);
    std::vector<int32_t> fidx((size_t)n * F);
    int64_t tot = 0;
    bool uniform = n > 0;
    for (int64_t i = 0; i < n; ++i) {
      const int64_t r = keep[i];
      const int64_t len = c->feat_ptr[r + 1] - c->feat_ptr[r];
      if (len < 0 || c->feat_idx[r * F + F - 1] != len) {
        DestroyGraph(b.g.release());
        return Fail(EULER_GPU_EINVAL, "graph_create: feature index does not cover values");
      }
      tot += len;
    }
    std::vector<float> fval((size_t)tot);
    int64_t foff = 0;
    for (int64_t i = 0; i < n; ++i) {
      const int64_t r = keep[i];
      const int64_t len = c->feat_ptr[r + 1] - c->feat_ptr[r];
      fptr[i] = foff;
      std::memcpy(fidx.data() + i * F, c->feat_idx + r * F, (size_t)F * 4);
      if (len > 0)
        std::memcpy(fval.data() + foff, c->feat_val + c->feat_ptr[r], (size_t)len * 4);
      uniform &= std::memcmp(fidx.data() + i * F, fidx.data(), (size_t)F * 4) == 0;
      foff += len;
    }
    fptr[n] = foff;
    v.n_float = F;
    v.feat_uniform = uniform ? 1 : 0;
    v.feat_stride = uniform ? (int64_t)fidx[F - 1] : 0;
    bool aligned = uniform;
    for (int32_t f = 0; aligned && f + 1 < F; ++f) aligned = fidx[f] % 4 == 0;
    b.g->feat_slot_aligned = aligned;
    v.feat_ptr = b.Upload(fptr.data(), fptr.size());
    v.feat_idx = b.Upload(fidx.data(), fidx.size());
    v.feat_val = b.Upload(fval.data(), fval.size());
    if (b.rc != EULER_GPU_OK) { DestroyGraph(b.g.release()); return b.rc; }
  }
  // global node sampler over this shard's nodes
  {
    std::vector<uint64_t> ids;
    std::vector<int32_t> types;
    std::vector<float> weights;
    ids.reserve(n); types.reserve(n); weights.reserve(n);
    auto push = [&](int64_t r) {
      ids.push_back(c->row_id[r]);
      types.push_back(c->node_type ? c->node_type[r] : 0);
      weights.push_back(c->node_weight ? c->node_weight[r] : 1.0f);
    };
    if (c->sampler_order) {
      std::map<uint64_t, int64_t> pos;
      for (int64_t i = 0; i < n; ++i) pos[row_id[i]] = keep[i];
      for (int64_t i = 0; i < c->n_rows; ++i) {
        auto it = pos.find(c->sampler_order[i]);
        if (it != pos.end()) push(it->second);
      }
    } else {
      for (int64_t i = 0; i < n; ++i) push(keep[i]);
    }
    int32_t n_types = c->n_node_types > 0 ? c->n_node_types : 1;
    int rc = BuildNodeSampler(&b, ids, types, weights, n_types);
    if (rc != EULER_GPU_OK) { DestroyGraph(b.g.release()); return rc; }
  }
  // sparse (uint64) features of the kept rows
  if (c->n_u64_features > 0) {
    if (!c->ufeat_ptr || !c->ufeat_idx || (c->ufeat_ptr[c->n_rows] > 0 && !c->ufeat_val)) {
      DestroyGraph(b.g.release());
      return Fail(EULER_GPU_EINVAL, "graph_create: null uint64 feature array");
    }
    const int32_t U = c->n_u64_features;
    std::vector<int64_t> uptr((size_t)n + 1, 0);
    std::vector<int32_t> uidx((size_t)n * U);
    std::vector<uint64_t> uval;
    for (int64_t i = 0; i < n; ++i) {
      const int64_t r = keep[i];
      const int64_t len = c->ufeat_ptr[r + 1] - c->ufeat_ptr[r];
      if (len < 0 || c->ufeat_idx[r * U + U - 1] != len) {
        DestroyGraph(b.g.release());
        return Fail(EULER_GPU_EINVAL, "graph_create: uint64 feature index does not cover values");
      }
      uptr[i] = (int64_t)uval.size();
      std::memcpy(uidx.data() + i * U, c->ufeat_idx + r * U, (size_t)U * 4);
      uval.insert(uval.end(), c->ufeat_val + c->ufeat_ptr[r], c->ufeat_val + c->ufeat_ptr[r] + len);
    }
    uptr[n] = (int64_t)uval.size();
    b.g->n_u64 = U;
    b.g->ufeat_ptr = b.Upload(uptr.data(), uptr.size());
    b.g->ufeat_idx = b.Upload(uidx.data(), uidx.size());
    if (uval.empty()) uval.push_back(0);      // keep the pointer valid
    b.g->ufeat_val = b.Upload(uval.data(), uval.size());
    if (b.rc != EULER_GPU_OK) { DestroyGraph(b.g.release()); return b.rc; }
  }
  // node types in row order (API_GET_NODE_T); graphs without types keep nullptr
  if (c->node_type && n > 0) {
    std::vector<int32_t> row_type((size_t)n);
    for (int64_t i = 0; i < n; ++i) row_type[i] = c->node_type[keep[i]];
    b.g->node_type_dev = b.Upload(row_type.data(), row_type.size());
    if (b.rc != EULER_GPU_OK) { DestroyGraph(b.g.release()); return b.rc; }
  }
  *out = b.g.release();
  return EULER_GPU_OK;
}

// ------------------------------------------------------------------------
// Synthetic power-law graph, generated directly in HBM (bench configs 2/3).
// Mirror of oracle/eo_synth.c: every value is a pure integer/IEEE function of
// (seed, node id, slot), so host and device agree bit-for-bit.
// ------------------------------------------------------------------------
struct SynthView {
  uint64_t seed;
  int64_t n_nodes;
  int32_t scale;
  int32_t n_types;
  int32_t weighted;
  int32_t hashed_ids;     // node x carries the external id Mix64(x) (oracle/eo_synth.c: eo_synth_external_id)
  double deg_table[64];
};

__host__ __device__ __forceinline__ uint64_t SynthHash(uint64_t seed,
                                                       uint64_t node,
                                                       uint64_t j, uint64_t c) {
  uint64_t h = Mix64(seed + node * 0x9E3779B97F4A7C15ULL);
  h = Mix64(h + j * 0xD1B54A32D192ED03ULL + c * 0x8CB92BA72F3D8DD7ULL);
  return h;
}

__device__ __forceinline__ int64_t SynthDegree(const SynthView& p, uint64_t node_id) {
  const uint64_t x = node_id - 1;
  const uint64_t mask = p.scale >= 64 ? ~0ULL : ((1ULL << p.scale) - 1);
  const double lam = p.deg_table[__popcll(x & mask) & 63];
  const double fl = floor(lam);
  const double frac = __dsub_rn(lam, fl);
  const uint64_t h = SynthHash(p.seed, node_id, ~0ULL, 0);
  const double u = __dmul_rn((double)(h >> 11), 1.0 / 9007199254740992.0);
  return 1 + (int64_t)fl + (u < frac ? 1 : 0);
}

__device__ __forceinline__ uint64_t SynthNeighbor(const SynthView& p,
                                                  uint64_t node_id, int64_t j) {
  const uint64_t x = node_id - 1;
  uint64_t bits = 0, h = 0;
  for (int i = 0; i < p.scale; ++i) {
    if ((i & 3) == 0) h = SynthHash(p.seed, node_id, (uint64_t)j, 1 + (i >> 2));
    const uint32_t slice = (uint32_t)(h >> (16 * (i & 3))) & 0xFFFFu;
    const uint32_t thr = ((x >> i) & 1) ? 13653u : 16384u;
    if (slice < thr) bits |= 1ULL << i;
  }
  return bits % (uint64_t)p.n_nodes + 1;
}

__device__ __forceinline__ float SynthWeight(const SynthView& p, uint64_t node_id,
                                             int64_t j) {
  if (!p.weighted) return 1.0f;
  const uint64_t h = SynthHash(p.seed, node_id, (uint64_t)j, 0);
  const float x = __fmul_rn((float)(uint32_t)(h >> 40), 1.0f / 16777216.0f);
  return __fadd_rn(0.5f, __fmul_rn(7.5f, x));
}

// row r of a shard holds node id = id_base + r * id_stride
__global__ void SynthDegreeKernel(SynthView p, uint64_t id_base, uint64_t id_stride,
                                  int64_t n_rows, int64_t* deg) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n_rows) deg[r] = SynthDegree(p, id_base + (uint64_t)r * id_stride);
}

// One lane per edge: neighbour id + RAW weight (prefix summed afterwards).
__global__ __launch_bounds__(256) void SynthEdgeKernel(
    SynthView p, uint64_t id_base, uint64_t id_stride, int64_t n_rows,
    const int64_t* row_ptr, int64_t n_edges, uint64_t* nbr, float* w) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges;
       e += stride) {
    // row = last r with row_ptr[r] <= e
    int64_t lo = 0, hi = n_rows;
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (row_ptr[mid] <= e) lo = mid; else hi = mid;
    }
    const uint64_t id = id_base + (uint64_t)lo * id_stride;
    const int64_t j = e - row_ptr[lo];
    const uint64_t nb = SynthNeighbor(p, id, j);
    nbr[e] = p.hashed_ids ? Mix64(nb) : nb;
    w[e] = SynthWeight(p, id, j);
  }
}

// hashed ids: the external id of every row, and the open-addressing id map
// (device_fns.h: FindRow) filled on the device - keys are distinct and never 0, so a slot
// is claimed by a CAS on its key word; the row number follows.
__global__ void SynthRowIdKernel(uint64_t id_base, uint64_t id_stride, int64_t n_rows,
                                 uint64_t* row_id) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n_rows) row_id[r] = Mix64(id_base + (uint64_t)r * id_stride);
}

__global__ void HashInitKernel(uint64_t* slots, uint64_t cap) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += stride) {
    slots[2 * i] = 0; slots[2 * i + 1] = ~0ULL;
  }
}

__global__ void HashInsertKernel(const uint64_t* row_id, int64_t n_rows, uint64_t* slots,
                                 uint64_t mask) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const uint64_t key = row_id[r];
  uint64_t h = Mix64(key) & mask;
  for (uint64_t probes = 0; probes <= mask; ++probes) {
    const unsigned long long old =
        atomicCAS(reinterpret_cast<unsigned long long*>(slots + 2 * h), 0ULL, (unsigned long long)key);
    if (old == 0ULL) { slots[2 * h + 1] = (uint64_t)r; return; }
    h = (h + 1) & mask;
  }
}

// Node::Init (core/graph/node.cc:46-66) per row: strictly sequential f32
// running sum across all types, per-type sums, row_meta record.  One lane per
// row (the adds of a row cannot be reordered without changing the floats).
__global__ __launch_bounds__(256) void SynthPrefixKernel(
    int32_t T, int32_t meta_stride, int64_t n_rows, const int64_t* row_ptr,
    float* w_inout, uint8_t* row_meta) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const int64_t b = row_ptr[r];
  const int64_t deg = row_ptr[r + 1] - b;
  uint8_t* rec = row_meta + r * (int64_t)meta_stride;
  *reinterpret_cast<int64_t*>(rec) = b;
  int32_t* type_end = reinterpret_cast<int32_t*>(rec + 8);
  float* type_prefix = reinterpret_cast<float*>(rec + 8 + 4 * T);
  float sum = 0.f, tsum = 0.f, tw = 0.f;
  int32_t t = 0;
  for (int64_t j = 0; j < deg; ++j) {
    while (t < T - 1 && j >= deg * (t + 1) / T) {
      type_end[t] = (int32_t)j;
      tsum = __fadd_rn(tsum, tw); type_prefix[t] = tsum; tw = 0.f; ++t;
    }
    const float w = w_inout[b + j];
    sum = __fadd_rn(sum, w);
    tw = __fadd_rn(tw, w);
    w_inout[b + j] = sum;
  }
  while (t < T) {
    type_end[t] = (int32_t)deg;
    tsum = __fadd_rn(tsum, tw); type_prefix[t] = tsum; tw = 0.f; ++t;
  }
}

int BuildGraphSynthetic(const euler_gpu_synth_params* sp, int device,
                        int32_t partitions, int32_t shard_index, int32_t shards,
                        euler_gpu_graph** out) {
  if (!sp || !out) return Fail(EULER_GPU_EINVAL, "graph_create_synthetic: null");
  if (sp->n_nodes <= 0 || sp->n_types <= 0 || sp->n_types > kMaxListedTypes ||
      sp->scale <= 0 || sp->scale > 40)
    return Fail(EULER_GPU_EINVAL, "graph_create_synthetic: bad parameters");
  if (shards <= 0 || shard_index < 0 || shard_index >= shards || partitions <= 0)
    return Fail(EULER_GPU_EINVAL, "graph_create_synthetic: bad shard arguments");
  if (shards > 1 && partitions % shards != 0)
    return Fail(EULER_GPU_EINVAL,
                "graph_create_synthetic: partitions must be a multiple of shards");
  EG_HIP(hipSetDevice(device));
  // owner(id) = (id % P) % S = id % S when S | P: the shard owns the ids
  // congruent to shard_index mod S -> strided identity id map.
  const uint64_t stride = (uint64_t)shards;
  uint64_t base = (uint64_t)shard_index;
  if (base == 0) base = stride;       // ids start at 1
  if (shards == 1) base = 1;
  const int64_t n_rows =
      base > (uint64_t)sp->n_nodes ? 0
                                   : (int64_t)(((uint64_t)sp->n_nodes - base) / stride + 1);
  KeepPoolMemory(device);
  GraphBuilder b;
  b.g->device = device;
  GraphView& v = b.g->view;
  const int32_t T = sp->n_types;
  v.T = T; v.meta_stride = 8 + 8 * T; v.n_rows = n_rows;
  v.map_mode = 0; v.id_base = base; v.id_stride = stride; v.row_id = nullptr;
  b.g->max_id = n_rows > 0 ? base + (uint64_t)(n_rows - 1) * stride : 0;
  v.has_zero_nbr = 0;
  v.monotone = 1;     // weights are >= 0.5: f32 running sums never decrease
  v.uniform_w = sp->weighted ? 0 : 1;   // unweighted: every weight is 1.0f, sums 1, 2, 3, ...
  SynthView p{};
  p.seed = sp->seed; p.n_nodes = sp->n_nodes; p.scale = sp->scale;
  p.n_types = sp->n_types; p.weighted = sp->weighted;
  p.hashed_ids = sp->hashed_ids != 0 ? 1 : 0;
  if (p.hashed_ids && shards != 1)
    return Fail(EULER_GPU_EINVAL, "graph_create_synthetic: hashed ids are not sharded (owner(id) "
                                  "would scatter the rows: build the shard from arrays)");
  std::memcpy(p.deg_table, sp->deg_table, sizeof(p.deg_table));
  const int block = 256;
  // degrees -> row_ptr (temporary, dropped after row_meta is built)
  int64_t* deg = nullptr;
  int64_t* row_ptr = nullptr;
  EG_HIP(hipMalloc((void**)&deg, (size_t)(n_rows + 1) * 8 + 16));
  EG_HIP(hipMalloc((void**)&row_ptr, (size_t)(n_rows + 1) * 8 + 16));
  EG_HIP(hipMemset(deg, 0, (size_t)(n_rows + 1) * 8));
  if (n_rows > 0)
    hipLaunchKernelGGL(SynthDegreeKernel, dim3((n_rows + block - 1) / block),
                       dim3(block), 0, 0, p, base, stride, n_rows, deg);
  {
    size_t tmp_bytes = 0;
    EG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, deg, row_ptr,
                                            n_rows + 1));
    void* tmp = nullptr;
    EG_HIP(hipMalloc(&tmp, tmp_bytes + 16));
    EG_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, deg, row_ptr,
                                            n_rows + 1));
    EG_HIP(hipDeviceSynchronize());
    EG_HIP(hipFree(tmp));
  }
  int64_t E = 0;
  EG_HIP(hipMemcpy(&E, row_ptr + n_rows, 8, hipMemcpyDeviceToHost));
  EG_HIP(hipFree(deg));
  v.n_edges = E;
  uint64_t* nbr = b.Alloc<uint64_t>((size_t)E);
  float* pw = b.Alloc<float>((size_t)E);
  uint8_t* meta = b.Alloc<uint8_t>((size_t)n_rows * v.meta_stride + 16);
  if (b.rc != EULER_GPU_OK) { (void)hipFree(row_ptr); DestroyGraph(b.g.release()); return b.rc; }
  if (E > 0)
    hipLaunchKernelGGL(SynthEdgeKernel, dim3(GridFor(E, block)), dim3(block), 0, 0,
                       p, base, stride, n_rows, row_ptr, E, nbr, pw);
  if (n_rows > 0)
    hipLaunchKernelGGL(SynthPrefixKernel, dim3((n_rows + block - 1) / block),
                       dim3(block), 0, 0, T, v.meta_stride, n_rows, row_ptr, pw,
                       meta);
  EG_HIP(hipGetLastError());
  EG_HIP(hipDeviceSynchronize());
  EG_HIP(hipFree(row_ptr));
  v.nbr = nbr; v.prefix_w = pw; v.row_meta = meta;
  if (p.hashed_ids && n_rows > 0) {
    uint64_t cap = 16;
    while (cap < (uint64_t)n_rows * 2 + 1) cap <<= 1;
    uint64_t* rid = b.Alloc<uint64_t>((size_t)n_rows);
    uint64_t* slots = b.Alloc<uint64_t>((size_t)(2 * cap));
    if (b.rc != EULER_GPU_OK) { DestroyGraph(b.g.release()); return b.rc; }
    hipLaunchKernelGGL(SynthRowIdKernel, dim3((n_rows + block - 1) / block), dim3(block), 0, 0,
                       base, stride, n_rows, rid);
    hipLaunchKernelGGL(HashInitKernel, dim3(GridFor((int64_t)cap, block)), dim3(block), 0, 0, slots, cap);
    hipLaunchKernelGGL(HashInsertKernel, dim3((n_rows + block - 1) / block), dim3(block), 0, 0,
                       rid, n_rows, slots, cap - 1);
    EG_HIP(hipGetLastError());
    EG_HIP(hipDeviceSynchronize());
    v.row_id = rid; v.map_mode = 1; v.hash_slots = slots; v.hash_mask = cap - 1;
    v.id_base = 0; v.id_stride = 1;
    b.g->max_id = ~0ULL;          // the ids are spread over the whole u64 range
  }
  {
    int rc = BuildSearchIndex(&b);
    if (rc != EULER_GPU_OK) { DestroyGraph(b.g.release()); return rc; }
  }
  b.g->has_sampler = false;   // uniform roots are drawn by the caller
  b.g->n_node_types = 1;
  *out = b.g.release();
  return EULER_GPU_OK;
}

// ---- weight-bucket index (wb_index.h) ------------------------------------------------
__global__ void WbCountKernel(GraphView g, uint32_t* nbk) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row > g.n_rows) return;
  uint32_t n = 0;
  if (row < g.n_rows) {
    const RowMeta m = LoadRowMeta(g, row);
    n = WbBuckets((uint32_t)m.type_end[g.T - 1]);
  }
  nbk[row] = n;
}

// the per-row records {wb_lo, row_lo, type_end[T], lim[T], type_sum[T] (T > 1)} (plain graphs read them as WbRec)
__global__ void WbRecKernel(GraphView g, const uint32_t* wb_lo, uint8_t* wbg, int32_t wbg_stride) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= g.n_rows) return;
  const RowMeta m = LoadRowMeta(g, row);
  uint8_t* out = wbg + row * (int64_t)wbg_stride;
  reinterpret_cast<uint32_t*>(out)[0] = wb_lo != nullptr ? wb_lo[row] : 0u;
  reinterpret_cast<uint32_t*>(out)[1] = (uint32_t)m.row_ptr;
  int32_t* te = reinterpret_cast<int32_t*>(out + 8);
  float* lim = reinterpret_cast<float*>(out + 8 + 4 * g.T);
  float* tsum = lim + g.T;                          // (T > 1 only)
  for (int32_t t = 0; t < g.T; ++t) {
    const int32_t e = m.type_end[t];
    te[t] = e;
    lim[t] = e > 0 ? g.prefix_w[m.row_ptr + e - 1] : 0.f;
    if (g.T > 1) tsum[t] = m.type_prefix[t];
  }
}

constexpr int kWbOverflowKeep = 4096;

// one lane per block: its row is the last one with wb_lo[row] <= block (rows without edges
// have no blocks and share their successor's offset)
__global__ __launch_bounds__(256) void WbFillKernel(GraphView g, const uint32_t* wb_lo,
                                                    int64_t n_wb, EdgeBlock* wb,
                                                    unsigned long long* overflows) {
  // overflows[0] = buckets that overflow their block, [1] = entries taken in the list of their
  // rows that follows (kWbOverflowKeep uint32 row numbers: what a test draws roots from)
  uint32_t* ovf_rows = reinterpret_cast<uint32_t*>(overflows + 2);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long mine = 0;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_wb; b += stride) {
    int64_t lo = 0, hi = g.n_rows;            // wb_lo[lo] <= b < wb_lo[hi]
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)wb_lo[mid] <= b) lo = mid; else hi = mid;
    }
    const RowMeta m = LoadRowMeta(g, lo);
    const uint32_t deg = (uint32_t)m.type_end[g.T - 1];
    EdgeBlock e;
    if (WbBuildBlock(g.prefix_w, g.nbr, (uint32_t)m.row_ptr, deg, g.prefix_w[m.row_ptr + deg - 1],
                     (uint32_t)(b - (int64_t)wb_lo[lo]), &e)) {
      ++mine;
      const unsigned long long k = atomicAdd(overflows + 1, 1ull);
      if (k < (unsigned long long)kWbOverflowKeep) ovf_rows[k] = (uint32_t)lo;
    }
    wb[b] = e;
  }
  if (mine != 0) atomicAdd(overflows, mine);
}

// 64-byte hash slots carrying the row record (common.h: GraphView::fat)
__global__ __launch_bounds__(256) void FatFillKernel(GraphView g, const uint8_t* trec, int32_t trec_stride,
                                                     uint8_t* fat) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h <= g.hash_mask; h += stride) {
    const uint64_t key = g.hash_slots[2 * h];
    const int64_t row = (int64_t)g.hash_slots[2 * h + 1];
    uint4 a = make_uint4(0u, 0u, 0u, 0u), b = make_uint4(0u, 0xFFFFFFFFu, 0u, 0u), c = make_uint4(0u, 0u, 0u, 0u);
    if (row >= 0) {
      const uint8_t* rec = trec + row * (int64_t)trec_stride;
      const uint32_t* hd = reinterpret_cast<const uint32_t*>(rec);
      const int32_t* te = reinterpret_cast<const int32_t*>(rec + 8);
      const float* lim = reinterpret_cast<const float*>(rec + 8 + 4 * g.T);
      const RowMeta m = LoadRowMeta(g, row);
      const int32_t t1 = g.T - 1;                        // T == 1: both entries are group 0
      a = make_uint4((uint32_t)key, (uint32_t)(key >> 32), hd[0], hd[1]);
      b = make_uint4((uint32_t)te[0], (uint32_t)te[t1], __float_as_uint(lim[0]), __float_as_uint(lim[t1]));
      c = make_uint4(__float_as_uint(m.type_prefix[0]), __float_as_uint(m.type_prefix[t1]),
                     (uint32_t)(uint64_t)row, (uint32_t)((uint64_t)row >> 32));
    }
    uint4* out = reinterpret_cast<uint4*>(fat + h * 64);
    out[0] = a; out[1] = b; out[2] = c; out[3] = make_uint4(0u, 0u, 0u, 0u);
  }
}

bool WbFits(size_t need);

// after the row records exist (v.trec): the fat slots of a hashed graph with T <= 2
int BuildFatSlots(GraphBuilder* b) {
  GraphView& v = b->g->view;
  if (v.map_mode != 1 || v.T > 2 || v.trec == nullptr || v.hash_slots == nullptr) return EULER_GPU_OK;
  const size_t bytes = ((size_t)v.hash_mask + 1) * 64;
  if (!WbFits(bytes)) return EULER_GPU_OK;
  uint8_t* fat = b->Alloc<uint8_t>(bytes);
  if (b->rc != EULER_GPU_OK) return b->rc;
  hipLaunchKernelGGL(FatFillKernel, dim3(GridFor((int64_t)v.hash_mask + 1, 256)), dim3(256), 0, 0, v, v.trec,
                     v.trec_stride, fat);
  EG_HIP(hipGetLastError());
  EG_HIP(hipDeviceSynchronize());
  v.fat = fat;
  return EULER_GPU_OK;
}

// How much HBM the weight-bucket index may take: at most `g_wb_budget_frac` of what
// is free when it is built (default one half - the rest stays for the caller's features, model
// and activations, which are usually allocated AFTER the first sampling call) and at most
// `g_wb_budget_bytes` (default: no absolute cap).  Process-wide, set through
// euler_gpu_set_index_budget or the environment (EULER_GPU_WB_INDEX=0 turns the index off,
// EULER_GPU_WB_INDEX_MAX_GB / EULER_GPU_WB_INDEX_MAX_FRACTION bound it).
std::atomic<int64_t> g_wb_budget_bytes{-2};      // -2 = read the environment first, -1 = no cap
std::atomic<int64_t> g_wb_budget_ppm{-2};        // fraction of the free HBM, parts per million

void WbBudgetFromEnv() {
  if (g_wb_budget_bytes.load() != -2) return;
  int64_t bytes = -1, ppm = 500000;
  if (const char* e = getenv("EULER_GPU_WB_INDEX")) {
    if (e[0] == '0') bytes = 0;
  }
  if (const char* e = getenv("EULER_GPU_WB_INDEX_MAX_GB")) {
    const double gb = atof(e);
    if (gb >= 0 && bytes != 0) bytes = (int64_t)(gb * 1073741824.0);
  }
  if (const char* e = getenv("EULER_GPU_WB_INDEX_MAX_FRACTION")) {
    const double f = atof(e);
    if (f >= 0 && f <= 1) ppm = (int64_t)(f * 1e6);
  }
  g_wb_budget_ppm.store(ppm);
  g_wb_budget_bytes.store(bytes);
}

// true when an index of `need` bytes may be built now
bool WbFits(size_t need) {
  WbBudgetFromEnv();
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return false; }
  const int64_t cap = g_wb_budget_bytes.load();
  if (cap >= 0 && need > (size_t)cap) return false;
  const double frac = (double)g_wb_budget_ppm.load() * 1e-6;
  if ((double)need > frac * (double)free_b) return false;
  return need + ((size_t)1 << 30) <= free_b;
}

int BuildWbIndex(GraphBuilder* b) {
  GraphView& v = b->g->view;
  // uniform weights, several edge-type groups: the row records alone (a draw there is an index
  // computation - PivotSample's H1 - and needs no buckets; the typed hops of the one-kernel
  // fanout read the group ends and the type sums from the record)
  if (v.monotone != 0 && v.uniform_w != 0 && (v.T > 1 || v.map_mode != 0) && v.n_rows > 0 && v.n_edges > 0 &&
      v.n_edges < ((int64_t)1 << 31)) {
    const int32_t stride = v.T == 1 ? 16 : 8 + 12 * v.T;
    if (!WbFits((size_t)v.n_rows * stride)) return EULER_GPU_OK;
    uint8_t* trec = b->Alloc<uint8_t>((size_t)v.n_rows * stride + 16);
    if (b->rc != EULER_GPU_OK) return b->rc;
    hipLaunchKernelGGL(WbRecKernel, dim3((v.n_rows + 255) / 256), dim3(256), 0, 0, v, (const uint32_t*)nullptr,
                       trec, stride);
    EG_HIP(hipGetLastError());
    EG_HIP(hipDeviceSynchronize());
    v.trec = trec; v.trec_stride = stride;
    return BuildFatSlots(b);
  }
  // rows must be non-decreasing (the keys decide by counting) and worth a search at all;
  // 32-bit edge and block numbers
  if (v.monotone == 0 || v.uniform_w != 0 || v.n_rows <= 0 || v.n_edges <= 0 ||
      v.n_edges >= ((int64_t)1 << 32) - 16 ||
      v.n_edges / 4 + v.n_rows >= ((int64_t)1 << 32) - 16)
    return EULER_GPU_OK;
  // (the 16-byte record of plain graphs: unsigned 32-bit edge and block numbers - the kernel of
  // fanout_plain.h reads them as such; the lean builds of fanout_local.h stop at 2^31 edges)
  const bool plain = v.T == 1 && v.total_in_meta != 0 && v.map_mode == 0;
  const int block = 256;
  // temporaries, released on every way out (EG_HIP returns early)
  struct Temps {
    uint32_t* nbk = nullptr; uint32_t* wb_lo = nullptr; void* scan = nullptr; unsigned long long* ovf = nullptr;
    ~Temps() {
      if (nbk) (void)hipFree(nbk);
      if (wb_lo) (void)hipFree(wb_lo);
      if (scan) (void)hipFree(scan);
      if (ovf) (void)hipFree(ovf);
    }
  } tmp;
  if (hipMalloc((void**)&tmp.nbk, ((size_t)v.n_rows + 1) * 4 + 16) != hipSuccess ||
      hipMalloc((void**)&tmp.wb_lo, ((size_t)v.n_rows + 1) * 4 + 16) != hipSuccess) {
    (void)hipGetLastError();
    return EULER_GPU_OK;          // no room: the pivot-level search stays
  }
  uint32_t* nbk = tmp.nbk;
  uint32_t* wb_lo = tmp.wb_lo;
  hipLaunchKernelGGL(WbCountKernel, dim3((v.n_rows + 1 + block - 1) / block), dim3(block), 0, 0, v, nbk);
  {
    size_t tmp_bytes = 0;
    EG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, nbk, wb_lo, v.n_rows + 1));
    EG_HIP(hipMalloc(&tmp.scan, tmp_bytes + 16));
    EG_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.scan, tmp_bytes, nbk, wb_lo, v.n_rows + 1));
    EG_HIP(hipDeviceSynchronize());
  }
  uint32_t n_wb32 = 0;
  EG_HIP(hipMemcpy(&n_wb32, wb_lo + v.n_rows, 4, hipMemcpyDeviceToHost));
  const int64_t n_wb = (int64_t)n_wb32;
  const int32_t stride = v.T == 1 ? 16 : 8 + 12 * v.T;
  // (checked before allocating: an index that does not fit is an optimisation declined)
  if (!WbFits((size_t)n_wb * sizeof(EdgeBlock) + (size_t)v.n_rows * stride)) return EULER_GPU_OK;
  uint8_t* wbg = b->Alloc<uint8_t>((size_t)v.n_rows * stride + 16);
  static_assert(offsetof(WbRec, lo) == 4 && offsetof(WbRec, deg) == 8 && offsetof(WbRec, total) == 12,
                "WbRec is the general record at T = 1");
  const WbRec* rec = plain ? reinterpret_cast<const WbRec*>(wbg) : nullptr;
  EdgeBlock* wb = b->Alloc<EdgeBlock>((size_t)n_wb);
  if (b->rc != EULER_GPU_OK) return b->rc;
  hipLaunchKernelGGL(WbRecKernel, dim3((v.n_rows + block - 1) / block), dim3(block), 0, 0, v, wb_lo,
                     wbg, stride);
  EG_HIP(hipMalloc((void**)&tmp.ovf, 16 + 4 * kWbOverflowKeep));
  unsigned long long* ovf = tmp.ovf;
  EG_HIP(hipMemset(ovf, 0, 16 + 4 * kWbOverflowKeep));
  if (n_wb > 0)
    hipLaunchKernelGGL(WbFillKernel, dim3(GridFor(n_wb, block)), dim3(block), 0, 0, v, wb_lo, n_wb, wb, ovf);
  EG_HIP(hipGetLastError());
  EG_HIP(hipDeviceSynchronize());
  unsigned long long n_ovf = 0;
  EG_HIP(hipMemcpy(&n_ovf, ovf, 8, hipMemcpyDeviceToHost));
  {
    const size_t keep = (size_t)std::min<unsigned long long>(n_ovf, (unsigned long long)kWbOverflowKeep);
    b->g->wb_overflow_rows.resize(keep);
    if (keep > 0) EG_HIP(hipMemcpy(b->g->wb_overflow_rows.data(), ovf + 2, keep * 4, hipMemcpyDeviceToHost));
  }
  v.wrec = rec; v.wb = wb; v.n_wb = n_wb; v.wbg = wbg; v.wbg_stride = stride;
  v.trec = wbg; v.trec_stride = stride;
  // (i.i.d. uniform weights: 1e-4; lognormal sigma 2, Pareto alpha 0.7: ~5e-2)
  v.wb_lean_ok = (double)n_ovf <= 0.002 * (double)n_wb ? 1 : 0;
  return BuildFatSlots(b);
}

int EnsureWbIndex(const euler_gpu_graph* cg) {
  euler_gpu_graph* g = const_cast<euler_gpu_graph*>(cg);
  if (g->wb_tried.load(std::memory_order_acquire) != 0) return EULER_GPU_OK;
  std::lock_guard<std::mutex> lk(g_blocked_mu);
  if (g->wb_tried.load(std::memory_order_acquire) != 0) return EULER_GPU_OK;
  int prev = 0;
  EG_HIP(hipGetDevice(&prev));
  EG_HIP(hipSetDevice(g->device));
  GraphBuilder b;
  b.g.reset(g);                 // borrow the graph: allocations land in its list
  const size_t n_alloc = g->allocations.size();
  const int64_t bytes0 = g->bytes;
  const int rc = BuildWbIndex(&b);
  b.g.release();
  if (rc != EULER_GPU_OK) {
    // The index is an optimisation: a failure while building it (no memory for a temporary, a
    // failed launch) DECLINES it - the view's fields are only set on success, what was
    // allocated for it is returned, the caller's sampling call goes on over the pivot levels.
    (void)hipGetLastError();
    (void)hipDeviceSynchronize();
    (void)hipGetLastError();
    while (g->allocations.size() > n_alloc) { (void)hipFree(g->allocations.back()); g->allocations.pop_back(); }
    g->bytes = bytes0;
    GraphView& v = g->view;
    v.wb = nullptr; v.wbg = nullptr; v.wrec = nullptr; v.n_wb = 0; v.wb_lean_ok = 0;
    v.trec = nullptr; v.trec_stride = 0; v.fat = nullptr;
  }
  (void)hipSetDevice(prev);
  g->wb_tried.store(1, std::memory_order_release);
  return EULER_GPU_OK;
}

// test hook (euler_gpu_set_tuning key 56): the next N builds of the EdgeBlocks fail as an
// allocation would (tests/test_gpu_parity.py: test_index_builds_declined)
std::atomic<int> g_blk_fail_next{0};

// The EdgeBlocks are an optimisation as the weight-bucket index is: when they cannot
// be built - no memory for 13 bytes per edge after the caller's features / model took the HBM -
// what the attempt allocated is returned, the decision is remembered (no rebuild per call) and
// the samplers run WITHOUT them: a view with neither index makes BlockPivotSample bisect the flat
// running sums and LoadSegment read the limits from them (k1_search.h) - same results, slower.
int EnsureBlockedIndex(const euler_gpu_graph* cg) {
  euler_gpu_graph* g = const_cast<euler_gpu_graph*>(cg);
  if (g->blk_ready.load(std::memory_order_acquire) != 0) return EULER_GPU_OK;
  std::lock_guard<std::mutex> lk(g_blocked_mu);
  if (g->blk_ready.load(std::memory_order_acquire) != 0) return EULER_GPU_OK;
  int prev = 0;
  EG_HIP(hipGetDevice(&prev));
  EG_HIP(hipSetDevice(g->device));
  GraphBuilder b;
  b.g.reset(g);                 // borrow the graph: allocations land in its list
  const size_t n_alloc = g->allocations.size();
  const int64_t bytes0 = g->bytes;
  int rc;
  if (g_blk_fail_next.load() > 0) {
    g_blk_fail_next.fetch_sub(1);
    (void)b.Alloc<uint8_t>(4096);        // (something to roll back)
    rc = Fail(EULER_GPU_ENOMEM, "EdgeBlocks: allocation failure injected (tuning key 56)");
  } else {
    rc = BuildBlockedIndex(&b);
  }
  b.g.release();
  if (rc != EULER_GPU_OK) {
    (void)hipGetLastError();
    (void)hipDeviceSynchronize();
    (void)hipGetLastError();
    while (g->allocations.size() > n_alloc) { (void)hipFree(g->allocations.back()); g->allocations.pop_back(); }
    g->bytes = bytes0;
    GraphView& v = g->view;
    v.blk = nullptr; v.skip1 = nullptr; v.bpiv = nullptr; v.n_blk = 0;
  }
  (void)hipSetDevice(prev);
  // 1 = built, 2 = declined: either way this graph is not asked again
  g->blk_ready.store(rc == EULER_GPU_OK ? 1 : 2, std::memory_order_release);
  return EULER_GPU_OK;
}

// rows -> host (spot checks)
__global__ void ExportMetaKernel(GraphView g, const uint64_t* ids, int64_t n,
                                 int64_t* deg, int64_t* src_off, int32_t* type_end,
                                 float* type_prefix) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t row = FindRow(g, ids[i]);
  if (row < 0) {
    deg[i] = 0; src_off[i] = 0;
    for (int t = 0; t < g.T; ++t) { type_end[i * g.T + t] = 0; type_prefix[i * g.T + t] = 0.f; }
    return;
  }
  const RowMeta m = LoadRowMeta(g, row);
  deg[i] = m.type_end[g.T - 1];
  src_off[i] = m.row_ptr;
  for (int t = 0; t < g.T; ++t) {
    type_end[i * g.T + t] = m.type_end[t];
    type_prefix[i * g.T + t] = m.type_prefix[t];
  }
}

}  // namespace euler_gpu

using namespace euler_gpu;

extern "C" {

const char* euler_gpu_last_error(void) { return g_last_error.c_str(); }
const char* euler_gpu_version(void) { return "euler-gpu 0.1 (gfx950)"; }

int euler_gpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int euler_gpu_set_index_budget(int64_t max_bytes, double max_free_fraction) {
  if (max_free_fraction > 1.0) return Fail(EULER_GPU_EINVAL, "set_index_budget: fraction > 1");
  WbBudgetFromEnv();
  if (max_bytes >= 0) g_wb_budget_bytes.store(max_bytes);
  if (max_free_fraction >= 0.0) g_wb_budget_ppm.store((int64_t)(max_free_fraction * 1e6));
  return EULER_GPU_OK;
}

int euler_gpu_graph_create(const euler_gpu_host_csr* csr, int device,
                           euler_gpu_graph** out) {
  return BuildGraphFromHost(csr, device, 1, 0, 1, out);
}

int euler_gpu_graph_create_shard(const euler_gpu_host_csr* csr, int device,
                                 int32_t partitions, int32_t shard_index,
                                 int32_t shards, euler_gpu_graph** out) {
  return BuildGraphFromHost(csr, device, partitions, shard_index, shards, out);
}

int euler_gpu_graph_create_synthetic(const euler_gpu_synth_params* p, int device,
                                     int32_t partitions, int32_t shard_index,
                                     int32_t shards, euler_gpu_graph** out) {
  return BuildGraphSynthetic(p, device, partitions, shard_index, shards, out);
}

int euler_gpu_graph_load(const char* data_path, int device, int32_t shard_index,
                         int32_t shards, euler_gpu_graph** out) {
  if (!data_path || !out) return Fail(EULER_GPU_EINVAL, "graph_load: null argument");
  DatGraph d;
  int rc = LoadDatDirectory(data_path, shard_index, shards, &d);
  if (rc != EULER_GPU_OK) return rc;
  euler_gpu_host_csr c{};
  d.Describe(&c);
  // the loader already kept only this shard's partitions
  rc = BuildGraphFromHost(&c, device, 1, 0, 1, out);
  if (rc == EULER_GPU_OK) (*out)->partitions = d.partitions;
  return rc;
}

int euler_gpu_dat_open(const char* data_path, int32_t shard_index,
                       int32_t shards, euler_gpu_host_csr* csr,
                       int32_t* partitions, void** owner) {
  if (!data_path || !csr || !owner)
    return Fail(EULER_GPU_EINVAL, "dat_open: null argument");
  std::unique_ptr<DatGraph> o(new DatGraph());
  int rc = LoadDatDirectory(data_path, shard_index, shards, o.get());
  if (rc != EULER_GPU_OK) return rc;
  o->Describe(csr);
  if (partitions) *partitions = o->partitions;
  *owner = o.release();
  return EULER_GPU_OK;
}

void euler_gpu_dat_close(void* owner) { delete static_cast<DatGraph*>(owner); }

int euler_gpu_dat_verify_edges(const char* data_path, int32_t shard_index, int32_t shards,
                               int64_t* edge_records, int64_t* not_in_rows,
                               int64_t* row_triples) {
  if (!data_path) return Fail(EULER_GPU_EINVAL, "dat_verify_edges: null path");
  DatGraph d;
  int rc = LoadDatDirectory(data_path, shard_index, shards, &d);
  if (rc != EULER_GPU_OK) return rc;
  return VerifyEdgeFiles(data_path, shard_index, shards, d, edge_records, not_in_rows,
                         row_triples);
}

void euler_gpu_graph_destroy(euler_gpu_graph* g) {
  {
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (g == g_default_graph) g_default_graph = nullptr;
  }
  DestroyGraph(g);
}

int64_t euler_gpu_graph_num_nodes(const euler_gpu_graph* g) { return g ? g->view.n_rows : -1; }
int64_t euler_gpu_graph_num_edges(const euler_gpu_graph* g) { return g ? g->view.n_edges : -1; }
int32_t euler_gpu_graph_num_edge_types(const euler_gpu_graph* g) { return g ? g->view.T : -1; }
int32_t euler_gpu_graph_num_node_types(const euler_gpu_graph* g) { return g ? g->n_node_types : -1; }
int euler_gpu_graph_device(const euler_gpu_graph* g) { return g ? g->device : -1; }
int32_t euler_gpu_graph_num_float_features(const euler_gpu_graph* g) {
  return g ? g->view.n_float : 0;
}
int64_t euler_gpu_graph_bytes(const euler_gpu_graph* g) { return g ? g->bytes : -1; }
int32_t euler_gpu_graph_partitions(const euler_gpu_graph* g) { return g ? g->partitions : 0; }

int euler_gpu_graph_id_range(const euler_gpu_graph* g, uint64_t* max_id_host,
                             int32_t* identity_host) {
  if (!g || !max_id_host || !identity_host)
    return Fail(EULER_GPU_EINVAL, "graph_id_range: null");
  *max_id_host = g->max_id;
  *identity_host = g->view.map_mode == 0 ? 1 : 0;
  return EULER_GPU_OK;
}

int euler_gpu_graph_set_node_sampler(euler_gpu_graph* g, int64_t n, const uint64_t* ids_host,
                                     const int32_t* types_host, const float* weights_host,
                                     int32_t n_node_types) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "set_node_sampler: null graph");
  if (n <= 0 || n_node_types <= 0 || n_node_types > kMaxNodeTypes)
    return Fail(EULER_GPU_EINVAL, "set_node_sampler: bad arguments");
  const GraphView& v = g->view;
  if (!ids_host && (v.map_mode != 0 || n != v.n_rows))
    return Fail(EULER_GPU_EINVAL, "set_node_sampler: ids required (the id map is not the strided "
                                  "identity, or n != the graph's rows)");
  std::vector<uint64_t> ids((size_t)n);
  std::vector<int32_t> types((size_t)n, 0);
  std::vector<float> weights((size_t)n, 1.0f);
  for (int64_t i = 0; i < n; ++i)
    ids[(size_t)i] = ids_host ? ids_host[i] : v.id_base + v.id_stride * (uint64_t)i;
  if (types_host) std::memcpy(types.data(), types_host, (size_t)n * sizeof(int32_t));
  if (weights_host) std::memcpy(weights.data(), weights_host, (size_t)n * sizeof(float));
  int prev = 0;
  EG_HIP(hipGetDevice(&prev));
  EG_HIP(hipSetDevice(g->device));
  const AliasEntry* old = g->has_sampler ? g->sampler.entries : nullptr;
  const int64_t old_bytes = g->has_sampler
      ? (int64_t)std::max<size_t>((size_t)g->sampler.type_off[g->sampler.n_types] * sizeof(AliasEntry), 16) : 0;
  const size_t n_alloc = g->allocations.size();
  const int64_t bytes0 = g->bytes;
  GraphBuilder b;
  b.g.reset(g);                 // borrow the graph: the table lands in its allocation list
  const int rc = BuildNodeSampler(&b, ids, types, weights, n_node_types);
  b.g.release();
  if (rc != EULER_GPU_OK) {
    // nothing of the graph was touched except, possibly, an allocation whose upload failed
    while (g->allocations.size() > n_alloc) { (void)hipFree(g->allocations.back()); g->allocations.pop_back(); }
    g->bytes = bytes0;
  } else if (old != nullptr) {
    auto it = std::find(g->allocations.begin(), g->allocations.end(), (void*)old);
    if (it != g->allocations.end()) {
      (void)hipDeviceSynchronize();      // (no launch may still read the old table)
      (void)hipFree(*it);
      g->allocations.erase(it);
      g->bytes -= old_bytes;
    }
  }
  (void)hipSetDevice(prev);
  return rc;
}

int euler_gpu_graph_index_overflow_rows(const euler_gpu_graph* g, uint64_t* ids_host, int64_t cap,
                                        int64_t* n_host) {
  if (!g || !n_host || cap < 0 || (cap > 0 && !ids_host))
    return Fail(EULER_GPU_EINVAL, "index_overflow_rows: bad arguments");
  *n_host = 0;
  const int rc = EnsureWbIndex(g);
  if (rc != EULER_GPU_OK) return rc;
  const GraphView& v = g->view;
  if (v.map_mode != 0) return EULER_GPU_OK;        // (ids of a hashed graph are not kept per row on the host)
  int64_t n = 0;
  for (uint32_t row : g->wb_overflow_rows) {
    if (n >= cap) break;
    ids_host[n++] = v.id_base + v.id_stride * (uint64_t)row;
  }
  *n_host = n;
  return EULER_GPU_OK;
}

int euler_gpu_graph_node_weight_sums(const euler_gpu_graph* g, float* out_host) {
  if (!g || !out_host) return Fail(EULER_GPU_EINVAL, "node_weight_sums: null");
  for (size_t i = 0; i < g->node_weight_sums.size(); ++i)
    out_host[i] = g->node_weight_sums[i];
  return EULER_GPU_OK;
}

int euler_gpu_graph_export_rows(const euler_gpu_graph* g, const uint64_t* ids_host,
                                int64_t n, int64_t* row_ptr_host,
                                int32_t* type_end_host, uint64_t* nbr_host,
                                float* prefix_w_host, float* type_prefix_host) {
  if (!g) return Fail(EULER_GPU_ENOGRAPH, "export_rows: null graph");
  if (n < 0 || !row_ptr_host) return Fail(EULER_GPU_EINVAL, "export_rows: bad args");
  EG_HIP(hipSetDevice(g->device));
  row_ptr_host[0] = 0;
  if (n == 0) return EULER_GPU_OK;
  const int32_t T = g->view.T;
  uint64_t* ids_dev = nullptr;
  int64_t* deg_dev = nullptr;
  EG_HIP(hipMalloc((void**)&ids_dev, n * 8));
  EG_HIP(hipMalloc((void**)&deg_dev, n * 16 + (size_t)n * T * 8));
  int64_t* off_dev = deg_dev + n;
  int32_t* te_dev = (int32_t*)(off_dev + n);
  float* tp_dev = (float*)(te_dev + n * T);
  EG_HIP(hipMemcpy(ids_dev, ids_host, n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(ExportMetaKernel, dim3((n + 255) / 256), dim3(256), 0, 0,
                     g->view, ids_dev, n, deg_dev, off_dev, te_dev, tp_dev);
  std::vector<int64_t> deg(n), off(n);
  EG_HIP(hipMemcpy(deg.data(), deg_dev, n * 8, hipMemcpyDeviceToHost));
  EG_HIP(hipMemcpy(off.data(), off_dev, n * 8, hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < n; ++i) row_ptr_host[i + 1] = row_ptr_host[i] + deg[i];
  if (nbr_host) {
    EG_HIP(hipMemcpy(type_end_host, te_dev, (size_t)n * T * 4, hipMemcpyDeviceToHost));
    EG_HIP(hipMemcpy(type_prefix_host, tp_dev, (size_t)n * T * 4, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; ++i) {
      if (deg[i] == 0) continue;
      EG_HIP(hipMemcpy(nbr_host + row_ptr_host[i], g->view.nbr + off[i],
                       (size_t)deg[i] * 8, hipMemcpyDeviceToHost));
      EG_HIP(hipMemcpy(prefix_w_host + row_ptr_host[i], g->view.prefix_w + off[i],
                       (size_t)deg[i] * 4, hipMemcpyDeviceToHost));
    }
  }
  EG_HIP(hipFree(ids_dev));
  EG_HIP(hipFree(deg_dev));
  return EULER_GPU_OK;
}

// tf_euler/utils/init_query_proxy.cc:19-37: "k=v;k=v".  Returns false on a
// malformed string exactly as the reference; a load failure also returns
// false (the reference would FATAL inside QueryProxy::Init).
bool InitQueryProxy(const char* conf) {
  if (!conf) return false;
  std::map<std::string, std::string> cfg;
  std::stringstream ss(conf);
  std::string item;
  bool any = false;
  while (std::getline(ss, item, ';')) {
    if (item.empty()) continue;
    const size_t eq = item.find('=');
    if (eq == std::string::npos || eq == 0 || eq + 1 >= item.size() ||
        item.find('=', eq + 1) != std::string::npos) {
      SetError("InitQueryProxy: malformed item '" + item + "'");
      return false;
    }
    cfg[item.substr(0, eq)] = item.substr(eq + 1);
    any = true;
  }
  if (!any) return false;
  const std::string mode = cfg.count("mode") ? cfg["mode"] : "local";
  if (mode != "local") {
    SetError("InitQueryProxy: only mode=local is served by the GPU backend "
             "(remote mode = gRPC shard servers is replaced by in-process "
             "multi-GPU sharding)");
    return false;
  }
  if (!cfg.count("data_path")) {
    SetError("InitQueryProxy: data_path missing");
    return false;
  }
  const int device = cfg.count("device") ? atoi(cfg["device"].c_str()) : 0;
  const int shard_idx = cfg.count("shard_idx") ? atoi(cfg["shard_idx"].c_str()) : 0;
  const int shard_num = cfg.count("shard_num") ? atoi(cfg["shard_num"].c_str()) : 1;
  // verify_edges=1: refuse a dataset whose Edge records are not exactly the
  // (src, dst, type) entries of its node rows (SparseGetAdj answers EdgeExist
  // from the rows; see euler_gpu_dat_verify_edges)
  if (cfg.count("verify_edges") && atoi(cfg["verify_edges"].c_str()) != 0) {
    int64_t records = 0, missing = 0, triples = 0;
    if (euler_gpu_dat_verify_edges(cfg["data_path"].c_str(), shard_idx, shard_num, &records,
                                   &missing, &triples) != EULER_GPU_OK)
      return false;
    if (missing != 0 || records != triples) {
      SetError("InitQueryProxy: Edge records and node rows disagree (" +
               std::to_string(records) + " records, " + std::to_string(missing) +
               " not in the rows, " + std::to_string(triples) + " row entries)");
      return false;
    }
  }
  euler_gpu_graph* g = nullptr;
  if (euler_gpu_graph_load(cfg["data_path"].c_str(), device, shard_idx, shard_num,
                           &g) != EULER_GPU_OK)
    return false;
  std::lock_guard<std::mutex> lk(g_default_mu);
  if (g_default_graph) DestroyGraph(g_default_graph);
  g_default_graph = g;
  return true;
}

euler_gpu_graph* euler_gpu_default_graph(void) {
  std::lock_guard<std::mutex> lk(g_default_mu);
  return g_default_graph;
}

}  // extern "C"
