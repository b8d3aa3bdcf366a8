"""tf_euler/python/euler_ops/base.py: graph initialisation.  The reference
loads libtf_euler.so and calls the C entry `InitQueryProxy("k=v;k=v")`
(tf_euler/utils/init_query_proxy.cc:19-37); the same entry of libeuler_gpu.so
loads the graph into HBM and installs it as the process-wide default graph."""
import ctypes as C

from .. import _lib
from ..graph import Graph

__all__ = ["initialize_graph", "initialize_embedded_graph",
           "initialize_shared_graph", "set_default_graph", "get_default_graph",
           "set_seed"]

_default = None


def set_default_graph(graph):
    """Install an already built euler_amd.Graph as the graph the module-level
    operators use (the reference has exactly one graph per process)."""
    global _default
    _default = graph
    return graph


def get_default_graph():
    if _default is None:
        raise RuntimeError("euler graph is not initialised: call "
                           "initialize_embedded_graph(...) or set_default_graph()")
    return _default


def set_seed(seed, call_id=0):
    """Make sampling reproducible (the reference's RNG cannot be seeded)."""
    get_default_graph().set_seed(seed, call_id)


def initialize_graph(config):
    """config: str "k=v;k=v" or dict (mode, data_path, sampler_type, data_type
    [, device, shard_idx, shard_num]).  Returns True on success."""
    if isinstance(config, dict):
        config = ';'.join('{}={}'.format(k, v) for k, v in config.items())
    if not isinstance(config, (str, bytes)):
        raise TypeError('Expect str or dict for graph config, '
                        'got {}.'.format(type(config).__name__))
    if not isinstance(config, bytes):
        config = config.encode()
    ok = bool(_lib.lib().InitQueryProxy(config))
    if ok:
        h = C.c_void_p(_lib.lib().euler_gpu_default_graph())
        g = Graph(h, _lib.lib().euler_gpu_graph_device(h))
        g.close = lambda: None      # owned by the library's default slot
        set_default_graph(g)
    return ok


def initialize_embedded_graph(data_dir, sampler_type='all', data_type='all',
                              device=0):
    return initialize_graph({'mode': 'local', 'data_path': data_dir,
                             'data_type': data_type,
                             'sampler_type': sampler_type, 'device': device})


def initialize_shared_graph(data_dir, zk_addr, zk_path, shard_num):
    """Remote mode (ZooKeeper + gRPC shard servers) is replaced by in-process
    multi-GPU sharding: see euler_amd.distributed.ShardedSampler."""
    raise NotImplementedError(
        "mode=remote is not served by the GPU backend; use "
        "euler_amd.distributed.ShardedSampler (one process per GPU over RCCL)")
