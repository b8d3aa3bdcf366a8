"""euler_amd - MI355X-native backend for Euler's minibatch-construction hot
path (SampleNode / SampleNeighbor / SampleFanout / RandomWalk + the
scatter/gather message-passing ops).  Operator names and semantics follow
tf_euler (`euler_amd.euler_ops` mirrors `tf_euler.python.euler_ops`); tensors
are torch tensors in HBM; the work is done by hand-written HIP kernels behind
the C ABI of include/euler_gpu.h.  There is no CPU fallback."""
from .graph import Graph, synth_params            # noqa: F401
from . import ops                                 # noqa: F401
from . import dataflow                            # noqa: F401
from .euler_ops import *                          # noqa: F401,F403
