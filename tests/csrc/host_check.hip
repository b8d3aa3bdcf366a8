// TEST INFRASTRUCTURE (not product code): runs the per-item functions of
// euler_amd/csrc/layer_fns.h - the very source the HIP kernels of
// layer_kernels.hip call one item per lane - in plain host loops over host
// arrays, so that `pytest -m "not gpu"` can compare their LOGIC with the
// oracle where no GPU exists.  What this cannot cover is the kernels' lane /
// wave mapping (that is the -m gpu tests' job).  Nothing in euler_amd/ links or
// loads this file; it is compiled on demand by tests/test_host_check.py with
// `hipcc -ffp-contract=off` (host pass only is used).
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "layer_fns.h"
#include "local_layer_host.h"
#include "wb_index.h"

using namespace euler_gpu;

namespace {

struct HostGraph {
  GraphView v{};
  std::vector<uint8_t> meta;
  std::vector<uint64_t> slots;
  std::vector<WbRec> wrec;
  std::vector<EdgeBlock> wb;
  int64_t wb_overflows = 0;
};

}  // namespace

extern "C" {

// CSR in the oracle's layout (row_ptr [n+1], type_end / type_prefix [n*T]).
void* hc_graph_create(int64_t n, int32_t T, const uint64_t* row_id,
                      const int64_t* row_ptr, const int32_t* type_end,
                      const uint64_t* nbr, const float* prefix_w,
                      const float* type_prefix, int32_t force_hash) {
  HostGraph* g = new HostGraph();
  GraphView& v = g->v;
  v.n_rows = n; v.T = T; v.meta_stride = 8 + 8 * T;
  v.n_edges = n ? row_ptr[n] : 0;
  g->meta.resize((size_t)n * v.meta_stride);
  for (int64_t i = 0; i < n; ++i) {
    uint8_t* rec = g->meta.data() + (size_t)i * v.meta_stride;
    std::memcpy(rec, &row_ptr[i], 8);
    std::memcpy(rec + 8, type_end + i * T, 4 * T);
    std::memcpy(rec + 8 + 4 * T, type_prefix + i * T, 4 * T);
  }
  v.row_meta = g->meta.data();
  v.nbr = nbr;
  v.prefix_w = prefix_w;
  v.row_id = row_id;
  bool identity = n > 0 && !force_hash;
  uint64_t base = n > 0 ? row_id[0] : 0, stride = 1;
  if (n > 1) {
    stride = row_id[1] - row_id[0];
    if (row_id[1] <= row_id[0]) identity = false;
  }
  for (int64_t i = 0; identity && i < n; ++i)
    identity = row_id[i] == base + (uint64_t)i * stride;
  if (identity) {
    v.map_mode = 0; v.id_base = base; v.id_stride = stride;
  } else {
    uint64_t cap = 16;
    while (cap < (uint64_t)n * 2 + 1) cap <<= 1;
    g->slots.assign(2 * cap, 0);
    for (uint64_t i = 0; i < cap; ++i) g->slots[2 * i + 1] = ~0ULL;
    for (int64_t i = 0; i < n; ++i) {
      uint64_t h = Mix64(row_id[i]) & (cap - 1);
      while ((int64_t)g->slots[2 * h + 1] >= 0 && g->slots[2 * h] != row_id[i])
        h = (h + 1) & (cap - 1);
      g->slots[2 * h] = row_id[i];
      g->slots[2 * h + 1] = (uint64_t)i;
    }
    v.map_mode = 1; v.hash_mask = cap - 1; v.id_base = 0; v.id_stride = 1;
    v.hash_slots = g->slots.data();
  }
  return g;
}

void hc_graph_destroy(void* h) { delete static_cast<HostGraph*>(h); }

// The weight-bucket index of a single-type graph, built by the per-block function the
// device builder runs one block per lane (wb_index.h: WbBuildBlock); returns the block count.
int64_t hc_wb_build(void* h) {
  HostGraph* g = static_cast<HostGraph*>(h);
  const GraphView& v = g->v;
  if (v.T != 1) return -1;
  g->wrec.assign((size_t)v.n_rows, WbRec{0, 0, 0, 0.f});
  uint64_t blocks = 0;
  for (int64_t r = 0; r < v.n_rows; ++r) {
    const RowMeta m = LoadRowMeta(v, r);
    const uint32_t deg = (uint32_t)m.type_end[0];
    WbRec rec{(uint32_t)blocks, (uint32_t)m.row_ptr, deg, deg ? v.prefix_w[m.row_ptr + deg - 1] : 0.f};
    g->wrec[(size_t)r] = rec;
    blocks += WbBuckets(deg);
  }
  g->wb.resize((size_t)blocks);
  for (int64_t r = 0; r < v.n_rows; ++r) {
    const WbRec& rec = g->wrec[(size_t)r];
    const uint32_t nbk = WbBuckets(rec.deg);
    for (uint32_t j = 0; j < nbk; ++j)
      if (WbBuildBlock(v.prefix_w, v.nbr, rec.lo, rec.deg, rec.total, j, &g->wb[rec.wb_lo + j])) ++g->wb_overflows;
  }
  return (int64_t)blocks;
}

// buckets whose block cannot hold every answer (what the device builder counts to decide
// whether the lean kernels may use the index)
int64_t hc_wb_overflows(void* h) { return static_cast<HostGraph*>(h)->wb_overflows; }

// One draw u on row `rows[i]`: the hot path (WbSampleHot) and, when it declines, the
// reference's search - exactly what the kernels do per lane.  cold_out[i] = 1 for a draw
// that took the cold way; m_out = row-relative edge, -1 for an empty row.
void hc_wb_sample(void* h, const int64_t* rows, const double* us, int64_t n, int64_t* m_out,
                  float* w_out, uint64_t* id_out, int32_t* cold_out) {
  HostGraph* g = static_cast<HostGraph*>(h);
  const GraphView& v = g->v;
  for (int64_t i = 0; i < n; ++i) {
    const WbRec& rec = g->wrec[(size_t)rows[i]];
    m_out[i] = -1; w_out[i] = 0.f; id_out[i] = 0; cold_out[i] = 0;
    if (rec.deg == 0) continue;
    uint64_t id = 0; float w = 0.f; uint32_t m = 0;
    if (WbSampleHot(g->wb.data(), rec, us[i], &id, &w, &m)) {
      m_out[i] = (int64_t)m - (int64_t)rec.lo; w_out[i] = w; id_out[i] = id;
      continue;
    }
    cold_out[i] = 1;
    const float* nw = v.prefix_w + rec.lo;
    const uint32_t mid = (uint32_t)RandomSelect(nw, 0, (uint64_t)(rec.deg - 1), us[i]);
    m_out[i] = mid; id_out[i] = v.nbr[rec.lo + mid];
    w_out[i] = nw[mid] - (mid == 0u ? 0.f : nw[mid - 1]);   // host pass only (-ffp-contract=off)
  }
}

void hc_edge_sum_weight(void* h, const uint64_t* ids, int64_t n, const int32_t* et,
                        int32_t k, float* out) {
  const GraphView& g = static_cast<HostGraph*>(h)->v;
  for (int64_t i = 0; i < n; ++i) out[i] = EdgeSumWeight(g, ids[i], et, k);
}

// The two kernels of euler_gpu_sample_root with the same interleaved scratch
// layout (slot j of batch row b at j * batch + b).
void hc_sample_root(uint64_t seed, uint32_t call_id, const uint64_t* roots,
                    const float* weights, int64_t batch, int32_t n, int32_t m,
                    int64_t default_node, uint64_t* out) {
  const int64_t cells = batch * n;
  std::vector<float> wn(cells), prob(cells), sum(batch);
  std::vector<int32_t> alias(cells), stack(cells);
  for (int64_t b = 0; b < batch; ++b)
    sum[b] = AliasBuildRow(weights + b * n, n, batch, wn.data() + b, prob.data() + b,
                           alias.data() + b, stack.data() + b);
  for (int64_t i = 0; i < batch * m; ++i) {
    const int64_t b = i / m;
    const int32_t j = (int32_t)(i - b * m);
    if (sum[b] == 0.f) {
      out[i] = (uint64_t)default_node;
    } else {
      const int32_t slot = SampleRootSlot(seed, call_id, b, j, n, batch,
                                          prob.data() + b, alias.data() + b);
      out[i] = roots[b * n + slot];
    }
  }
}

void hc_sample_layer(void* h, uint64_t seed, uint32_t call_id, const uint64_t* roots,
                     int64_t n, const int32_t* et, int32_t k, int64_t default_node,
                     uint64_t* out_id, float* out_w, int32_t* out_t) {
  const GraphView& g = static_cast<HostGraph*>(h)->v;
  int32_t types[kMaxListedTypes] = {0};
  for (int32_t i = 0; i < k && i < kMaxListedTypes; ++i) types[i] = et[i];
  for (int64_t i = 0; i < n; ++i)
    SampleLayerAt(g, seed, call_id, i, roots[i], types, k, default_node, out_id + i,
                  out_w + i, out_t + i);
}

// euler_gpu_local_sample_layer with its kernel replaced by a host loop over the
// same LocalLayerPick.
int hc_local_sample_layer(uint64_t seed, uint32_t call_id, const int32_t* idx,
                          const uint64_t* ids, const float* w, const int32_t* t,
                          int64_t total, int64_t batch, int32_t n, int32_t m,
                          const char* weight_func, int64_t default_node, uint64_t* out_id,
                          float* out_w, int32_t* out_t) {
  LocalLayerTables tb;
  if (!BuildLocalLayerTables(idx, ids, w, t, total, batch, n,
                             std::string(weight_func) == "sqrt", &tb))
    return -1;
  const uint64_t fill = 0x0101010101010101ULL * (uint64_t)(uint8_t)default_node;
  for (int64_t x = 0; x < batch * m; ++x) {
    const int64_t b = x / m;
    const int64_t s0 = tb.seg[b];
    const int64_t mid = LocalLayerPick(seed, call_id, b, (int32_t)(x - b * m),
                                       tb.sum_w.data() + s0, tb.seg[b + 1] - s0);
    if (mid < 0) { out_id[x] = fill; out_w[x] = 0.f; out_t[x] = 0; continue; }
    out_id[x] = tb.u_id[s0 + mid];
    out_w[x] = tb.u_w[s0 + mid];
    out_t[x] = tb.u_t[s0 + mid];
  }
  return 0;
}

// The per-sample logic of the generic K1 kernel (k1_variants.hip:
// SampleNeighborKernel = InitRowSampler + SampleAt per (root, slot)) with the
// API_SAMPLE_NB fill for empty rows (count x (0, 0.0, 0)).
void hc_sample_neighbor_core(void* h, uint64_t seed, uint32_t call_id, const uint64_t* ids,
                             int64_t n, const int32_t* et, int32_t k, int32_t count,
                             uint64_t* out_id, float* out_w, int32_t* out_t) {
  const GraphView& g = static_cast<HostGraph*>(h)->v;
  int32_t types[kMaxListedTypes] = {0};
  for (int32_t i = 0; i < k && i < kMaxListedTypes; ++i) types[i] = et[i];
  for (int64_t i = 0; i < n; ++i) {
    RowSampler rs;
    InitRowSampler(rs, g, FindRow(g, ids[i]), types, k);
    for (int32_t j = 0; j < count; ++j) {
      const int64_t o = i * count + j;
      if (!rs.valid) { out_id[o] = 0; out_w[o] = 0.f; out_t[o] = 0; continue; }
      SampleAt(rs, seed, call_id, ids[i], j, out_id + o, out_w + o, out_t + o);
    }
  }
}

// mask[r * m + j] = EdgeExistAny(roots[r], l_nb[(r / n) * m + j])
void hc_edge_exist_mask(void* h, const uint64_t* roots, const uint64_t* l_nb,
                        int64_t batch, int32_t n, int32_t m, const int32_t* et,
                        int32_t k, uint8_t* mask) {
  const GraphView& g = static_cast<HostGraph*>(h)->v;
  for (int64_t r = 0; r < batch * n; ++r) {
    const int64_t row = FindRow(g, roots[r]);
    for (int32_t j = 0; j < m; ++j)
      mask[r * m + j] = EdgeExistAny(g, row, l_nb[(r / n) * m + j], et, k) ? 1 : 0;
  }
}

}  // extern "C"
