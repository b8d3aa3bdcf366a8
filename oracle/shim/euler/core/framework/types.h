// TEST INFRASTRUCTURE: forwarding header (see op_kernel.h beside it) - the mirror
// include/euler_op_framework.h declares Tensor / TensorShape / DataType / DAGNodeProto.
#include "euler/core/framework/op_kernel.h"
