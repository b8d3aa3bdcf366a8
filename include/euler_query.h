// Client-side shim: euler::Query / euler::QueryProxy with the signatures the
// reference's TF kernels call (euler/client/query.h:33-68,
// euler/client/query_proxy.h:46-59), routed to the MI355X backend.
//
// A TF kernel of the hot path builds a GQL string, fills named input tensors
// and calls QueryProxy::GetInstance()->RunAsyncGremlin(query, callback).  This
// shim recognises the query shapes those kernels generate - there is no GQL
// parser (euler/parser needs flex / bison and is out of scope):
//
//   tf_euler/kernels/sample_neighbor_op.cc:36-40
//       v(nodes).sampleNB(edge_types, nb_count,<D>).as(nb)
//   tf_euler/kernels/sample_fanout_op.cc:37-42
//       v(nodes).sampleNB(et_0,nb_count_0,<D>).as(nb_0).sampleNB(et_1,...).as(nb_1)...
//   tf_euler/kernels/random_walk_op.cc:181-185          (p = q = 1 walk)
//       v(nodes).sampleNB(et_0, nb_count_, <D>).as(nb_0).sampleNB(et_1, ...)...
//   tf_euler/kernels/random_walk_op.cc:73               (node2vec step)
//       v(nodes).outV(edge_types).as(nb)
//   tf_euler/kernels/sample_node_op.cc:63,72
//       sampleN(node_type, count).as(id)
//
// i.e. `v(<ids>)` followed by one or more `.sampleNB(<types>, <count>, <D>).as(<alias>)`
// steps, `v(<ids>).outV(<types>).as(<alias>)`, and `sampleN(<type>, <count>).as(<alias>)`;
// argument names are the caller's input tensor names.  Each step runs the plugin
// op the reference's translator would emit (API_SAMPLE_NB / API_GET_NB_NODE /
// API_SAMPLE_NODE, include/euler_op_framework.h) with the alias as node name, so
// the results are named "<alias>:<i>" in the FillNeighbor layout
// (core/kernels/common.cc:275-334) exactly as the TF kernels read them.  A
// `.has(...)` condition (attribute indexes) or any other shape is logged and
// produces no results, the reference's error behaviour on the compute path.
//
// A maintainer points the TF kernels at this header instead of
// euler/client/query_proxy.h + query.h and links libeuler_gpu.so; the kernel
// bodies stay as they are.
#pragma once

#include <functional>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "euler_op_framework.h"

namespace euler {
inline namespace gpu_abi {

class Query {
 public:
  explicit Query(const std::string& gremlin);
  ~Query();
  Query(const Query&) = delete;
  Query& operator=(const Query&) = delete;

  // The tensor is owned by the query (freed with it), uninitialised like the
  // reference's (core/framework/tensor.cc:23-27).
  Tensor* AllocInput(const std::string& name, const TensorShape& shape,
                     const DataType& type);
  // Missing names map to nullptr (the reference leaves them out of the map).
  std::unordered_map<std::string, Tensor*> GetResult(
      const std::vector<std::string>& result_names);
  Tensor* GetResult(const std::string& result_name);
  bool SingleOpQuery() { return false; }
  const std::string& gremlin() const { return gremlin_; }

 private:
  std::shared_ptr<OpKernelContext> ctx_;
  std::string gremlin_;
  friend class QueryProxy;
};

class QueryProxy {
 public:
  typedef std::function<void()> DoneCallback;
  QueryProxy(const QueryProxy&) = delete;
  void operator=(const QueryProxy&) = delete;

  // The singleton exists once a graph is installed - InitQueryProxy("k=v;...")
  // of include/euler_gpu.h, the reference's own C entry - or after Init(graph).
  // nullptr (and an error line) otherwise, as in the reference.
  static QueryProxy* GetInstance();
  static bool Init(euler_gpu_graph* graph);

  std::unordered_map<std::string, Tensor*> RunGremlin(
      Query* query, const std::vector<std::string>& result_names);
  // Runs on one of the proxy's 8 query threads (the reference's client pool,
  // client/query_proxy.cc:205-210) and calls `callback` from that thread.
  void RunAsyncGremlin(Query* query, DoneCallback callback);

  int32_t GetShardNum() { return 1; }

  // Reproducible sampling (not in the reference, whose RNG cannot be seeded):
  // fixes the seed and restarts the process-wide call-id sequence.  Without
  // it the seed comes from std::random_device once per process and every op
  // invocation draws a fresh call id.
  static void SetSeed(uint64_t seed);

 private:
  QueryProxy() {}
  bool Execute(Query* query);
};

}  // namespace gpu_abi
}  // namespace euler

extern "C" {
// C view for tests and non-C++ hosts: runs `gremlin` with the named inputs
// (n_inputs tensors: name, dtype as euler::DataType, element count, data) on the
// default graph through Query / QueryProxy::RunAsyncGremlin, waits, and copies
// result `result_name` into out (capacity bytes).  Returns the result's byte
// size, -1 when the query produced no such result, -2 when out is too small.
int64_t euler_query_run(const char* gremlin, int32_t n_inputs, const char* const* names,
                        const int32_t* dtypes, const int64_t* counts,
                        const void* const* data, const char* result_name, void* out,
                        int64_t capacity);
void euler_query_set_seed(uint64_t seed);
/* QueryProxy::Init(graph): the proxy serves this (borrowed) graph instead of the
 * process default; NULL returns to the default graph. */
void euler_query_set_graph(euler_gpu_graph* graph);
}
