// TEST INFRASTRUCTURE: forwarding header at the reference's include path
// (euler/core/framework/op_kernel.h) so that the reference's own kernel sources -
// core/kernels/id_unique_op.cc, idx_gather_op.cc, data_gather_op.cc - compile UNMODIFIED against
// the plugin-API mirror include/euler_op_framework.h (oracle/Makefile: ref_kernels).  The
// kernels then register in the mirror's registry under "REF:" + their op name: the GPU kernels
// of libeuler_gpu.so hold the plain names, and a duplicate name is fatal as in the reference
// (core/framework/op_kernel.cc:203-207).
#ifndef ORACLE_SHIM_EULER_CORE_FRAMEWORK_OP_KERNEL_H_
#define ORACLE_SHIM_EULER_CORE_FRAMEWORK_OP_KERNEL_H_

#include "euler_op_framework.h"

#undef REGISTER_OP_KERNEL_UNIQ
#define REGISTER_OP_KERNEL_UNIQ(counter, name, cls)                      \
  static ::euler::OpKernelRegistrar registrar__##counter##__obj(         \
      std::string("REF:") + name, [](const std::string& op) -> ::euler::OpKernel* { \
        return new cls(op);                                              \
      });

#endif  // ORACLE_SHIM_EULER_CORE_FRAMEWORK_OP_KERNEL_H_
