// Host-side mirror of the reference's plugin API for the ops on the hot path
// (SURVEY.md §8b): euler::Tensor / OpKernel / OpKernelContext /
// REGISTER_OP_KERNEL with the shapes of euler/core/framework/op_kernel.h:38-130
// and tensor.h:57-118, minus protobuf: a node definition is reduced to
// {name, op, inputs[]} (the only DAGNodeProto fields the hot-path kernels read:
// core/kernels/sample_neighbor_op.cc:37-60, common.cc OutputName).
//
// The GPU kernels register under the reference's own op names
// (API_SAMPLE_NB, API_SAMPLE_NODE, ID_UNIQUE, IDX_GATHER, DATA_GATHER,
// API_GET_NB_NODE and the layerwise chain API_GET_EDGE_SUM_WEIGHT,
// API_SAMPLE_ROOT, API_SAMPLE_L, API_SPARSE_GEN_ADJ, API_SPARSE_GET_ADJ,
// API_GATHER_RESULT), so a build of the reference that links this
// library INSTEAD of the corresponding core/kernels/*.cc files dispatches the
// same DAG nodes to the MI355X.  Tensors are host buffers (malloc, uninitialised
// like the reference's, op_kernel.cc:92-105); device memory stays behind the
// C ABI.
//
// Coexistence with the real euler:: classes: everything here lives in the INLINE
// namespace euler::gpu_abi.  Source written against the reference's names
// (`euler::OpKernel`, `euler::Tensor`, REGISTER_OP_KERNEL) compiles unchanged
// against this header, while the symbols are `euler::gpu_abi::...` - a binary may
// link libeuler_gpu.so next to the reference's libeuler_core without an ODR
// clash, and a translation unit that includes BOTH headers fails to compile
// (ambiguous names) instead of silently mixing the two class layouts.
#pragma once

#include <stdint.h>

#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "euler_gpu.h"

namespace euler {
inline namespace gpu_abi {

enum DataType : int32_t {   // euler/core/framework/types.h:26-39
  kInt8 = 0, kInt16, kInt32, kInt64, kUInt8, kUInt16, kUInt32, kUInt64,
  kFloat, kDouble, kBool, kString
};
size_t SizeOfType(DataType t);

class TensorShape {
 public:
  TensorShape() {}
  TensorShape(std::initializer_list<size_t> dims) : dims_(dims) {}
  explicit TensorShape(const std::vector<size_t>& dims) : dims_(dims) {}
  const std::vector<size_t>& Dims() const { return dims_; }
  size_t Size() const { return dims_.size(); }
  size_t NumElements() const {
    size_t n = 1;
    for (auto d : dims_) n *= d;
    return n;
  }
 private:
  std::vector<size_t> dims_;
};

// Host tensor (core/framework/tensor.h).  The memory is the tensor's own, released by its destructor; tensors of
// 16 KB or more are PINNED host blocks from a process-wide cache (the results of a GPU op arrive by PCIe: pageable
// memory is staged by the runtime at a tenth of the link's rate).  EULER_GPU_PINNED_POOL_MB (default 4096) caps the
// blocks the cache keeps; 0 = plain malloc like the reference (op_kernel.cc:92-105).
class Tensor {
 public:
  Tensor(const TensorShape& shape, DataType type);
  ~Tensor();
  Tensor(const Tensor&) = delete;
  Tensor& operator=(const Tensor&) = delete;
  template <typename T> T* Raw() const { return reinterpret_cast<T*>(data_); }
  const TensorShape& Shape() const { return shape_; }
  int NumElements() const { return (int)shape_.NumElements(); }
  DataType Type() const { return type_; }
  size_t TotalBytes() const { return shape_.NumElements() * SizeOfType(type_); }
 private:
  TensorShape shape_;
  DataType type_;
  void* data_;
};

// What the reference's Status (euler/common/status.h) is to an op body: `auto s = ctx->tensor(..);
// if (!s.ok()) ...`.  Converts to and from the int codes this library returns (0 = ok), so both
// spellings compile: `if (ctx->Allocate(..) != 0)` and `if (!ctx->Allocate(..).ok())`.
class OpStatus {
 public:
  OpStatus(int code = 0) : code_(code) {}          // NOLINT: implicit on purpose
  bool ok() const { return code_ == 0; }
  int code() const { return code_; }
  operator int() const { return code_; }            // NOLINT
  static OpStatus OK() { return OpStatus(0); }
 private:
  int code_;
};

// The fields of DAGNodeProto the hot-path kernels use, readable BOTH ways: as the protobuf
// accessors the reference's kernel sources call (`node_def.name()`, `node_def.inputs(0)`,
// `node_def.inputs_size()`, `node_def.post_process(i)`, `node_def.dnf_size()` - core/kernels/
// id_unique_op.cc:39, sample_neighbor_op.cc:39-86) and as plain members (`nd.inputs.push_back(..)`,
// `nd.name = "x"`): a member is a std::string / std::vector<std::string> that can also be called.
struct NodeStr : std::string {
  NodeStr() {}
  NodeStr(const std::string& s) : std::string(s) {}       // NOLINT
  NodeStr(const char* s) : std::string(s) {}              // NOLINT
  using std::string::operator=;
  const std::string& operator()() const { return *this; }
};
struct NodeStrList : std::vector<std::string> {
  using std::vector<std::string>::vector;
  NodeStrList() {}
  NodeStrList(const std::vector<std::string>& v) : std::vector<std::string>(v) {}    // NOLINT
  const std::string& operator()(int i) const { return (*this)[(size_t)i]; }
};
struct NodeDef {
  NodeStr name;                     // outputs are "<name>:<i>"
  NodeStr op;
  NodeStrList inputs;               // tensor names looked up in the context
  // DAGNodeProto.post_process of API_GET_NB_NODE: "order_by id|weight [desc]",
  // "limit k" (core/kernels/get_neighbor_op.cc:117-168)
  NodeStrList post_process;
  int inputs_size() const { return (int)inputs.size(); }
  int post_process_size() const { return (int)post_process.size(); }
  int dnf_size() const { return 0; }               // conditions (`has` / index lookups): out of scope
  void set_name(const std::string& v) { name = v; }
  void set_op(const std::string& v) { op = v; }
  void add_inputs(const std::string& v) { inputs.push_back(v); }
  void add_post_process(const std::string& v) { post_process.push_back(v); }
};
// The reference's kernels spell the node definition `DAGNodeProto` (core/framework/op_kernel.h:35).
typedef NodeDef DAGNodeProto;
std::string OutputName(const NodeDef& node_def, int i);
std::string OutputName(const std::string& name, int i);

class OpKernelContext {
 public:
  ~OpKernelContext();
  // 0 / ok() on success (the reference returns Status, op_kernel.h:84-99).
  OpStatus Allocate(const std::string& name, const TensorShape& shape, DataType type,
                    Tensor** tensor);
  OpStatus tensor(const std::string& name, Tensor** tensor);
  OpStatus Deallocate(const std::string& name);
  // A second name for an existing tensor (op_kernel.cc AddAlias; the context
  // frees every distinct tensor once, op_kernel.cc:80-90).
  OpStatus AddAlias(const std::string& name, Tensor* tensor);
  OpStatus RemoveAlias(const std::string& name);
  // Sampling randomness (not in the reference, whose RNG is time(0)-seeded and
  // cannot be fixed).  A context the host did not touch - what a DAG executor
  // creates per query - samples with the PROCESS seed (std::random_device once
  // per process, or SetProcessSeed) and takes every op invocation's call id from
  // a process-wide atomic sequence, so two queries never repeat each other's
  // draws.  SetSeed / SetCallId pin this context for reproducible runs: its ops
  // then use seed, call_id, call_id + 1, ...
  void SetSeed(uint64_t seed) { seed_ = seed; seed_set_ = true; }
  void SetCallId(uint32_t first) { call_id_ = first; call_id_set_ = true; }
  uint64_t seed() const;
  uint32_t NextCallId();
  static void SetProcessSeed(uint64_t seed);   // also restarts the call-id sequence
  void SetGraph(euler_gpu_graph* g) { graph_ = g; }
  euler_gpu_graph* graph() const;
 private:
  std::mutex mu_;
  std::unordered_map<std::string, Tensor*> tensor_map_;
  uint64_t seed_ = 0;
  uint32_t call_id_ = 0;
  bool seed_set_ = false, call_id_set_ = false;
  euler_gpu_graph* graph_ = nullptr;
};

class OpKernel {
 public:
  explicit OpKernel(const std::string& name) : name_(name) {}
  virtual ~OpKernel() {}
  // stateless + thread-safe, results persisted in ctx (op_kernel.h:40-49)
  virtual void Compute(const NodeDef& node_def, OpKernelContext* ctx) = 0;
  const std::string& name() const { return name_; }
 private:
  std::string name_;
};

class AsyncOpKernel : public OpKernel {
 public:
  using OpKernel::OpKernel;
  typedef std::function<void()> DoneCallback;
  void Compute(const NodeDef& node_def, OpKernelContext* ctx) override;
  virtual void AsyncCompute(const NodeDef& node_def, OpKernelContext* ctx,
                            DoneCallback callback) = 0;
};

class OpKernelRegistrar {
 public:
  typedef OpKernel* (*Factory)(const std::string& name);
  OpKernelRegistrar(const std::string& name, Factory factory);
};

// 0 / ok() = found / created (cached singleton per op name, op_kernel.cc:217-231).
OpStatus LookupOpKernel(const std::string& name);
OpStatus CreateOpKernel(const std::string& name, OpKernel** kernel);

#define REGISTER_OP_KERNEL(name, cls) \
  REGISTER_OP_KERNEL_UNIQ_HELPER(__COUNTER__, name, cls)
#define REGISTER_OP_KERNEL_UNIQ_HELPER(counter, name, cls) \
  REGISTER_OP_KERNEL_UNIQ(counter, name, cls)
#define REGISTER_OP_KERNEL_UNIQ(counter, name, cls)                      \
  static ::euler::OpKernelRegistrar registrar__##counter##__obj(         \
      name, [](const std::string& op) -> ::euler::OpKernel* {            \
        return new cls(op);                                              \
      });

}  // namespace gpu_abi
}  // namespace euler

// C view of the registry, used by tests and by non-C++ hosts.
extern "C" {
int euler_op_registered(const char* op_name);
// Runs one op on host buffers: builds a context with the given named int/uint64
// input tensors, executes, and copies output `out_index` into `out` (up to
// out_capacity bytes).  Returns the output's byte size or a negative code.
int64_t euler_op_run_sample_nb(euler_gpu_graph* g, uint64_t seed,
                               const uint64_t* node_ids, int64_t n,
                               const int32_t* edge_types, int32_t k,
                               int32_t count, int32_t* idx_out, uint64_t* id_out,
                               float* w_out, int32_t* t_out);
// The same op with DAGNodeProto.post_process strings (';'-separated: "order_by id|weight
// [desc]", "limit k"; core/kernels/sample_neighbor_op.cc:86-132) and an explicit call
// id; rows then have different lengths: outputs have room for n * count entries,
// idx_out [n, 2] delimits them.  Returns the number of entries or < 0.
int64_t euler_op_run_sample_nb_post(euler_gpu_graph* g, uint64_t seed, uint32_t call_id,
                                    const uint64_t* node_ids, int64_t n,
                                    const int32_t* edge_types, int32_t k, int32_t count,
                                    const char* post_process, int32_t* idx_out,
                                    uint64_t* id_out, float* w_out, int32_t* t_out);
// API_GET_NB_NODE with ';'-separated post-process strings ("order_by weight
// desc;limit 4"); outputs have room for `capacity` neighbours.
int64_t euler_op_run_get_nb(euler_gpu_graph* g, const uint64_t* node_ids, int64_t n,
                            const int32_t* edge_types, int32_t k,
                            const char* post_process, int64_t capacity,
                            int32_t* idx_out, uint64_t* id_out, float* w_out,
                            int32_t* t_out);
// The six-node DAG of `v(nodes).sampleLNB(edge_types, n, m, default_node)`
// (euler/parser/translator.cc:338-386,489-527; with a non-empty weight_func the
// API_GET_NB_NODE -> API_LOCAL_SAMPLE_L form, :388-441) executed node by node through
// the registry: adjacency idx [batch*n, 2], adjacency ids (room for `capacity`),
// sampled layer [batch*m].  Returns the number of adjacency ids or < 0.
int64_t euler_op_run_sample_lnb(euler_gpu_graph* g, uint64_t seed, uint32_t first_call_id,
                                const uint64_t* node_ids, int64_t batch, int32_t n,
                                const int32_t* edge_types, int32_t k, int32_t m,
                                const char* weight_func, int64_t default_node,
                                int64_t capacity,
                                int32_t* adj_idx_out, uint64_t* adj_id_out,
                                uint64_t* l_nb_out);
}
