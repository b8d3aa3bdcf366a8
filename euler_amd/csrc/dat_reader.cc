// Reader for the on-disk graph the reference loads in Graph::Init: the
// `euler.meta` header (core/graph/graph_builder.cc:230-307) and the
// `Node/*_<partition>.dat` record files (graph_builder.cc:310-320; record =
// Node::DeSerialize, core/graph/node.cc:414-526; primitives =
// common/bytes_io.h:28-81, common/file_io.h:113-135).  Records go straight into
// the flat CSR arrays that are uploaded to HBM - no per-node heap objects.
#include <dirent.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <fstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace euler_gpu {

namespace {

struct Cursor {
  const char* p;
  size_t n;
  size_t i = 0;
  template <typename T>
  bool Get(T* v) {
    if (i + sizeof(T) > n) return false;
    memcpy(v, p + i, sizeof(T));
    i += sizeof(T);
    return true;
  }
  template <typename T>
  bool GetVec(std::vector<T>* v) {
    uint32_t num = 0;
    if (!Get(&num)) return false;
    if (i + (size_t)num * sizeof(T) > n) return false;
    v->resize(num);
    if (num) memcpy(v->data(), p + i, (size_t)num * sizeof(T));
    i += (size_t)num * sizeof(T);
    return true;
  }
  bool GetString(std::string* s) {
    uint32_t len = 0;
    if (!Get(&len)) return false;
    if (i + len > n) return false;
    s->assign(p + i, len);
    i += len;
    return true;
  }
};

bool ReadFile(const std::string& path, std::string* out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  f.seekg(0, std::ios::end);
  const std::streamoff sz = f.tellg();
  f.seekg(0);
  out->resize((size_t)sz);
  if (sz > 0) f.read(&(*out)[0], sz);
  return (bool)f;
}

// Graph::Init's file filter (core/graph/graph.cc:90-98): "<x>_<idx>.dat" with
// idx % shard_number == shard_index.
bool KeepFile(const std::string& name, int32_t shard_index, int32_t shards) {
  std::vector<std::string> parts;
  std::string cur;
  for (char ch : name) {
    if (ch == '_' || ch == '.') { if (!cur.empty()) parts.push_back(cur); cur.clear(); }
    else cur.push_back(ch);
  }
  if (!cur.empty()) parts.push_back(cur);
  return parts.size() == 3 && parts[2] == "dat" &&
         atoi(parts[1].c_str()) % shards == shard_index;
}

// One partition file parsed on its own (row offsets relative to the file).
struct FilePart {
  std::vector<uint64_t> row_id, nbr, ufeat_val;
  std::vector<int32_t> type_end, node_type;
  std::vector<float> prefix_w, type_prefix, node_weight, feat_val;
  std::vector<int64_t> row_end, feat_end, ufeat_end;     // cumulative, per row
  std::vector<std::vector<int32_t>> f_idx_rows, u_idx_rows;
  std::string error;
};

void ParseNodeFile(const std::string& path, const std::string& fn, int32_t T, FilePart* out) {
  std::string blob;
  if (!ReadFile(path, &blob)) { out->error = "graph_load: cannot read " + fn; return; }
  std::vector<int32_t> gids, gidx, in_gids, in_gidx, u64_idx, f32_idx;
  std::vector<float> gw, nw, in_gw, in_nw, f32_val;
  std::vector<uint64_t> nb, in_nb, u64_val;
  Cursor f{blob.data(), blob.size()};
  while (f.i < f.n) {
    uint32_t len = 0;
    if (!f.Get(&len) || f.i + len > f.n) { out->error = "graph_load: truncated record in " + fn; return; }
    Cursor r{blob.data() + f.i, len};
    f.i += len;
    uint64_t id; int32_t type; float weight;
    if (!(r.Get(&id) && r.Get(&type) && r.Get(&weight) && r.GetVec(&gids) &&
          r.GetVec(&gw) && r.GetVec(&gidx) && r.GetVec(&nb) && r.GetVec(&nw))) {
      out->error = "graph_load: malformed node record in " + fn; return;
    }
    if ((int32_t)gids.size() > T || gids.size() != gw.size() ||
        gids.size() != gidx.size() || nb.size() != nw.size()) {
      out->error = "graph_load: inconsistent node record in " + fn; return;
    }
    out->row_id.push_back(id);
    out->node_type.push_back(type);
    out->node_weight.push_back(weight);
    // CompactWeightedCollection::Init(ids, weights): running f32 sums
    float acc = 0.f;
    int32_t last_end = 0;
    for (int32_t t = 0; t < T; ++t) {
      if (t < (int32_t)gids.size()) {
        if (gids[t] != t) { out->error = "graph_load: edge group ids must be 0..T-1"; return; }
        acc += gw[t];
        last_end = gidx[t];
      }
      out->type_end.push_back(last_end);
      out->type_prefix.push_back(acc);
    }
    if (last_end != (int32_t)nb.size()) { out->error = "graph_load: group index does not cover row"; return; }
    out->nbr.insert(out->nbr.end(), nb.begin(), nb.end());
    out->prefix_w.insert(out->prefix_w.end(), nw.begin(), nw.end());
    out->row_end.push_back((int64_t)out->nbr.size());
    // in-neighbour block (same five vectors), then uint64 / float / binary
    // features (node.cc:462-523); the uint64 and float features are kept
    if (!(r.GetVec(&in_gids) && r.GetVec(&in_gw) && r.GetVec(&in_gidx) &&
          r.GetVec(&in_nb) && r.GetVec(&in_nw) && r.GetVec(&u64_idx) &&
          r.GetVec(&u64_val) && r.GetVec(&f32_idx) && r.GetVec(&f32_val))) {
      out->error = "graph_load: malformed feature block in " + fn; return;
    }
    if (!f32_idx.empty() && f32_idx.back() != (int32_t)f32_val.size()) {
      out->error = "graph_load: float feature index does not cover values"; return;
    }
    if (!u64_idx.empty() && u64_idx.back() != (int32_t)u64_val.size()) {
      out->error = "graph_load: uint64 feature index does not cover values"; return;
    }
    out->f_idx_rows.push_back(f32_idx);
    out->feat_val.insert(out->feat_val.end(), f32_val.begin(), f32_val.end());
    out->feat_end.push_back((int64_t)out->feat_val.size());
    out->u_idx_rows.push_back(u64_idx);
    out->ufeat_val.insert(out->ufeat_val.end(), u64_val.begin(), u64_val.end());
    out->ufeat_end.push_back((int64_t)out->ufeat_val.size());
  }
}

}  // namespace

void DatGraph::Describe(euler_gpu_host_csr* c) const {
  *c = euler_gpu_host_csr{};
  c->n_rows = (int64_t)row_id.size();
  c->n_edge_types = n_edge_types; c->n_node_types = n_node_types;
  c->row_id = row_id.data(); c->row_ptr = row_ptr.data();
  c->type_end = type_end.data(); c->nbr = nbr.data();
  c->prefix_w = prefix_w.data(); c->type_prefix = type_prefix.data();
  c->node_type = node_type.data(); c->node_weight = node_weight.data();
  c->n_float_features = n_float;
  if (n_float > 0) {
    c->feat_ptr = feat_ptr.data(); c->feat_idx = feat_idx.data();
    c->feat_val = feat_val.data();
  }
  c->n_u64_features = n_u64;
  if (n_u64 > 0) {
    c->ufeat_ptr = ufeat_ptr.data(); c->ufeat_idx = ufeat_idx.data();
    c->ufeat_val = ufeat_val.data();
  }
}

int LoadDatDirectory(const char* data_path, int32_t shard_index, int32_t shards,
                     DatGraph* out) {
  std::vector<uint64_t>* row_id = &out->row_id;
  std::vector<int64_t>* row_ptr = &out->row_ptr;
  std::vector<int32_t>* type_end = &out->type_end;
  std::vector<uint64_t>* nbr = &out->nbr;
  std::vector<float>* prefix_w = &out->prefix_w;
  std::vector<float>* type_prefix = &out->type_prefix;
  std::vector<int32_t>* node_type = &out->node_type;
  std::vector<float>* node_weight = &out->node_weight;
  int32_t* n_edge_types = &out->n_edge_types;
  int32_t* n_node_types = &out->n_node_types;
  int32_t* partitions = &out->partitions;
  if (shards <= 0 || shard_index < 0 || shard_index >= shards)
    return Fail(EULER_GPU_EINVAL, "graph_load: bad shard arguments");
  const std::string root(data_path);
  // ---- euler.meta
  std::string meta;
  if (!ReadFile(root + "/euler.meta", &meta))
    return Fail(EULER_GPU_EIO, "graph_load: cannot read " + root + "/euler.meta");
  Cursor m{meta.data(), meta.size()};
  std::string name, version;
  uint64_t node_count = 0, edge_count = 0;
  int32_t parts = 0;
  bool ok = m.GetString(&name) && m.GetString(&version) && m.Get(&node_count) &&
            m.Get(&edge_count) && m.Get(&parts);
  for (int pass = 0; ok && pass < 2; ++pass) {   // node features, edge features
    uint32_t cnt = 0;
    ok = m.Get(&cnt);
    for (uint32_t i = 0; ok && i < cnt; ++i) {
      std::string fname; int32_t ftype, idx; int64_t dim;
      ok = m.GetString(&fname) && m.Get(&ftype) && m.Get(&idx) && m.Get(&dim);
    }
  }
  uint32_t nt = 0, et = 0;
  ok = ok && m.Get(&nt);
  for (uint32_t i = 0; ok && i < nt; ++i) {
    std::string s; uint32_t idx;
    ok = m.GetString(&s) && m.Get(&idx);
  }
  ok = ok && m.Get(&et);
  if (!ok) return Fail(EULER_GPU_EIO, "graph_load: malformed euler.meta");
  if (parts <= 0) return Fail(EULER_GPU_EIO, "graph_load: partitions_num must be > 0");
  *n_node_types = (int32_t)nt;
  *n_edge_types = (int32_t)et;
  *partitions = parts;
  if (et == 0 || et > (uint32_t)kMaxListedTypes)
    return Fail(EULER_GPU_EIO, "graph_load: need 1..32 edge types");
  // ---- Node/*.dat
  const std::string node_dir = root + "/Node";
  DIR* d = opendir(node_dir.c_str());
  if (!d) return Fail(EULER_GPU_EIO, "graph_load: no directory " + node_dir);
  std::vector<std::string> files;
  while (dirent* ent = readdir(d)) {
    const std::string fn(ent->d_name);
    if (KeepFile(fn, shard_index, shards)) files.push_back(fn);
  }
  closedir(d);
  std::sort(files.begin(), files.end());
  // The partition files are independent: up to 8 host threads parse them into
  // per-file parts, which are then appended in file order (the order a single
  // reader would produce).
  const int32_t T = (int32_t)et;
  std::vector<FilePart> parts_(files.size());
  {
    const size_t hw = std::max<size_t>(1, std::thread::hardware_concurrency());
    const size_t n_thr = std::min<size_t>(std::min<size_t>(8, hw), files.size());
    auto work = [&](size_t first) {
      for (size_t f = first; f < files.size(); f += std::max<size_t>(n_thr, 1))
        ParseNodeFile(node_dir + "/" + files[f], files[f], T, &parts_[f]);
    };
    if (n_thr <= 1) {
      if (!files.empty()) work(0);
    } else {
      std::vector<std::thread> pool;
      for (size_t t = 0; t < n_thr; ++t) pool.emplace_back(work, t);
      for (auto& th : pool) th.join();
    }
  }
  for (const FilePart& fp : parts_)
    if (!fp.error.empty()) return Fail(EULER_GPU_EIO, fp.error);
  row_id->clear(); row_ptr->assign(1, 0); type_end->clear(); nbr->clear();
  prefix_w->clear(); type_prefix->clear(); node_type->clear(); node_weight->clear();
  std::vector<std::vector<int32_t>> f_idx_rows, u_idx_rows;
  out->feat_ptr.assign(1, 0);
  out->feat_val.clear();
  out->ufeat_ptr.assign(1, 0);
  out->ufeat_val.clear();
  for (FilePart& fp : parts_) {
    const int64_t e0 = (int64_t)nbr->size();
    const int64_t f0 = (int64_t)out->feat_val.size(), u0 = (int64_t)out->ufeat_val.size();
    row_id->insert(row_id->end(), fp.row_id.begin(), fp.row_id.end());
    node_type->insert(node_type->end(), fp.node_type.begin(), fp.node_type.end());
    node_weight->insert(node_weight->end(), fp.node_weight.begin(), fp.node_weight.end());
    type_end->insert(type_end->end(), fp.type_end.begin(), fp.type_end.end());
    type_prefix->insert(type_prefix->end(), fp.type_prefix.begin(), fp.type_prefix.end());
    nbr->insert(nbr->end(), fp.nbr.begin(), fp.nbr.end());
    prefix_w->insert(prefix_w->end(), fp.prefix_w.begin(), fp.prefix_w.end());
    out->feat_val.insert(out->feat_val.end(), fp.feat_val.begin(), fp.feat_val.end());
    out->ufeat_val.insert(out->ufeat_val.end(), fp.ufeat_val.begin(), fp.ufeat_val.end());
    for (size_t r = 0; r < fp.row_id.size(); ++r) {
      row_ptr->push_back(e0 + fp.row_end[r]);
      out->feat_ptr.push_back(f0 + fp.feat_end[r]);
      out->ufeat_ptr.push_back(u0 + fp.ufeat_end[r]);
    }
    for (auto& v : fp.f_idx_rows) f_idx_rows.push_back(std::move(v));
    for (auto& v : fp.u_idx_rows) u_idx_rows.push_back(std::move(v));
    fp = FilePart();                       // release the part's memory as we go
  }
  int32_t F = 0;
  for (const auto& v : f_idx_rows) F = std::max(F, (int32_t)v.size());
  out->n_float = F;
  out->feat_idx.assign((size_t)F * f_idx_rows.size(), 0);
  for (size_t i = 0; i < f_idx_rows.size(); ++i) {
    int32_t last = 0;
    for (int32_t f = 0; f < F; ++f) {
      if (f < (int32_t)f_idx_rows[i].size()) last = f_idx_rows[i][f];
      out->feat_idx[i * F + f] = last;       // a missing slot has length 0
    }
  }
  int32_t U = 0;
  for (const auto& v : u_idx_rows) U = std::max(U, (int32_t)v.size());
  out->n_u64 = U;
  out->ufeat_idx.assign((size_t)U * u_idx_rows.size(), 0);
  for (size_t i = 0; i < u_idx_rows.size(); ++i) {
    int32_t last = 0;
    for (int32_t f = 0; f < U; ++f) {
      if (f < (int32_t)u_idx_rows[i].size()) last = u_idx_rows[i][f];
      out->ufeat_idx[i * U + f] = last;
    }
  }
  return EULER_GPU_OK;
}

// Edge/*.dat against the node rows: the reference answers EdgeExist from its Edge
// records (core/api/api.cc:46-48), this backend from the adjacency rows; the two
// agree exactly when every Edge record (src, dst, type - the first 20 bytes of a
// record, core/graph/edge.cc:136-153) is an entry of src's row and the rows hold
// no further (src, dst, type) triples.
int VerifyEdgeFiles(const char* data_path, int32_t shard_index, int32_t shards,
                    const DatGraph& g, int64_t* edge_records, int64_t* not_in_rows,
                    int64_t* row_triples) {
  const std::string edge_dir = std::string(data_path) + "/Edge";
  DIR* d = opendir(edge_dir.c_str());
  if (!d) return Fail(EULER_GPU_EIO, "dat_verify_edges: no directory " + edge_dir);
  std::vector<std::string> files;
  while (dirent* ent = readdir(d)) {
    const std::string fn(ent->d_name);
    if (KeepFile(fn, shard_index, shards)) files.push_back(fn);
  }
  closedir(d);
  std::sort(files.begin(), files.end());
  const int32_t T = g.n_edge_types;
  const int64_t n = (int64_t)g.row_id.size();
  std::unordered_map<uint64_t, int64_t> row_of;
  row_of.reserve((size_t)n * 2);
  for (int64_t r = 0; r < n; ++r) row_of[g.row_id[r]] = r;
  // distinct (dst) per (row, type) segment = the triples the rows hold
  int64_t triples = 0;
  for (int64_t r = 0; r < n; ++r) {
    for (int32_t t = 0; t < T; ++t) {
      const int64_t b = g.row_ptr[r] + (t == 0 ? 0 : g.type_end[r * T + t - 1]);
      const int64_t e = g.row_ptr[r] + g.type_end[r * T + t];
      std::vector<uint64_t> seg(g.nbr.begin() + b, g.nbr.begin() + e);
      std::sort(seg.begin(), seg.end());
      triples += std::unique(seg.begin(), seg.end()) - seg.begin();
    }
  }
  int64_t records = 0, missing = 0;
  for (const auto& fn : files) {
    std::string blob;
    if (!ReadFile(edge_dir + "/" + fn, &blob))
      return Fail(EULER_GPU_EIO, "dat_verify_edges: cannot read " + fn);
    Cursor f{blob.data(), blob.size()};
    while (f.i < f.n) {
      uint32_t len = 0;
      if (!f.Get(&len) || f.i + len > f.n || len < 20)
        return Fail(EULER_GPU_EIO, "dat_verify_edges: truncated record in " + fn);
      Cursor r{blob.data() + f.i, len};
      f.i += len;
      uint64_t src, dst; int32_t type;
      r.Get(&src); r.Get(&dst); r.Get(&type);
      ++records;
      bool found = false;
      auto it = row_of.find(src);
      if (it != row_of.end() && type >= 0 && type < T) {
        const int64_t row = it->second;
        const int64_t b = g.row_ptr[row] + (type == 0 ? 0 : g.type_end[row * T + type - 1]);
        const int64_t e = g.row_ptr[row] + g.type_end[row * T + type];
        for (int64_t p = b; p < e && !found; ++p) found = g.nbr[p] == dst;
      }
      if (!found) ++missing;
    }
  }
  if (edge_records) *edge_records = records;
  if (not_in_rows) *not_in_rows = missing;
  if (row_triples) *row_triples = triples;
  return EULER_GPU_OK;
}

}  // namespace euler_gpu
