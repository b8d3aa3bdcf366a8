// TEST INFRASTRUCTURE (never part of the product): a C entry per op that runs ONE op kernel of
// the plugin-API mirror's registry (include/euler_op_framework.h) on host buffers.  op_name
// "ID_UNIQUE" runs the GPU kernel libeuler_gpu.so registered; "REF:ID_UNIQUE" the REFERENCE's
// own core/kernels/id_unique_op.cc, compiled unmodified into this library against the mirror
// (oracle/Makefile: ref_kernels; shim headers under oracle/shim/).  tests/ compare the two.
#include <string.h>

#include <string>

#include "euler_op_framework.h"

using euler::DAGNodeProto;
using euler::OpKernel;
using euler::OpKernelContext;
using euler::Tensor;
using euler::TensorShape;

namespace {
Tensor* Put(OpKernelContext* ctx, const std::string& name, const TensorShape& shape, euler::DataType t,
            const void* src) {
  Tensor* x = nullptr;
  if (!ctx->Allocate(name, shape, t, &x).ok()) return nullptr;
  if (x->TotalBytes() > 0) memcpy(x->Raw<char>(), src, x->TotalBytes());
  return x;
}
}  // namespace

extern "C" {

// ids [n] (node ids) -> unique ids (first-occurrence order) + gather_idx [n]; returns the number
// of unique ids or < 0 (core/kernels/id_unique_op.cc:35-64)
int64_t refk_id_unique(const char* op_name, const uint64_t* ids, int64_t n, uint64_t* uniq_out,
                       int32_t* gather_idx_out) {
  OpKernelContext ctx;
  if (!Put(&ctx, "in:0", TensorShape({(size_t)n}), euler::kUInt64, ids)) return -1;
  DAGNodeProto nd;
  nd.set_name("u"); nd.set_op(op_name); nd.add_inputs("in:0");
  OpKernel* k = nullptr;
  if (!euler::CreateOpKernel(op_name, &k).ok()) return -2;
  k->Compute(nd, &ctx);
  Tensor *u = nullptr, *gi = nullptr;
  if (!ctx.tensor("u:0", &u).ok() || !ctx.tensor("u:1", &gi).ok()) return -3;
  memcpy(uniq_out, u->Raw<char>(), u->TotalBytes());
  memcpy(gather_idx_out, gi->Raw<char>(), gi->TotalBytes());
  return u->NumElements();
}

// idx [m, 2], gather_idx [n] -> idx_out [n, 2] (core/kernels/idx_gather_op.cc:33-55)
int64_t refk_idx_gather(const char* op_name, const int32_t* idx, int64_t m, const int32_t* gather_idx,
                        int64_t n, int32_t* idx_out) {
  OpKernelContext ctx;
  if (!Put(&ctx, "idx:0", TensorShape({(size_t)m, 2}), euler::kInt32, idx) ||
      !Put(&ctx, "g:0", TensorShape({(size_t)n}), euler::kInt32, gather_idx)) return -1;
  DAGNodeProto nd;
  nd.set_name("o"); nd.set_op(op_name); nd.add_inputs("idx:0"); nd.add_inputs("g:0");
  OpKernel* k = nullptr;
  if (!euler::CreateOpKernel(op_name, &k).ok()) return -2;
  k->Compute(nd, &ctx);
  Tensor* o = nullptr;
  if (!ctx.tensor("o:0", &o).ok()) return -3;
  memcpy(idx_out, o->Raw<char>(), o->TotalBytes());
  return o->NumElements() / 2;
}

// data [total] of dtype (euler::DataType: 2 kInt32, 7 kUInt64, 8 kFloat), idx [m, 2],
// gather_idx [n] -> data_out (room for out_capacity elements); returns the element count
// (core/kernels/data_gather_op.cc:33-80)
int64_t refk_data_gather(const char* op_name, const void* data, int64_t total, int32_t dtype,
                         const int32_t* idx, int64_t m, const int32_t* gather_idx, int64_t n,
                         void* data_out, int64_t out_capacity) {
  OpKernelContext ctx;
  if (!Put(&ctx, "d:0", TensorShape({(size_t)total}), (euler::DataType)dtype, data) ||
      !Put(&ctx, "idx:0", TensorShape({(size_t)m, 2}), euler::kInt32, idx) ||
      !Put(&ctx, "g:0", TensorShape({(size_t)n}), euler::kInt32, gather_idx)) return -1;
  DAGNodeProto nd;
  nd.set_name("o"); nd.set_op(op_name);
  nd.add_inputs("d:0"); nd.add_inputs("idx:0"); nd.add_inputs("g:0");
  OpKernel* k = nullptr;
  if (!euler::CreateOpKernel(op_name, &k).ok()) return -2;
  k->Compute(nd, &ctx);
  Tensor* o = nullptr;
  if (!ctx.tensor("o:0", &o).ok()) return -3;
  if (o->NumElements() > out_capacity) return -4;
  memcpy(data_out, o->Raw<char>(), o->TotalBytes());
  return o->NumElements();
}

}  // extern "C"
