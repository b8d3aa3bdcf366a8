"""CPU-only checks of the product's host side: the C-ABI library loads and
exports every symbol include/euler_gpu.h declares, the .dat reader parses the
reference tool's files, host-only entry points behave, and the ops fail
loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from euler_amd import _lib
    return _lib.lib()


def test_library_exports_every_declared_symbol(L):
    from euler_amd import _lib
    hdr = "".join(open(os.path.join(ROOT, "include", f)).read()
                  for f in ("euler_gpu.h", "euler_gpu_measure.h", "euler_op_framework.h",
                            "euler_query.h"))
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b((?:euler_(?:gpu|shm|op|query)_\w+)|InitQueryProxy)\s*\(", hdr))
    declared -= {"euler_gpu_graph", "euler_gpu_host_csr", "euler_gpu_synth_params"}
    assert len(declared) >= 55 and "euler_shm_alltoall_i64" in declared
    # the measurement surface stays out of the product header
    product = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "euler_gpu.h")).read(), flags=re.S)
    for name in ("euler_gpu_set_tuning", "euler_gpu_set_debug_buffer", "euler_gpu_time_sample_fanout",
                 "euler_gpu_sample_neighbor_algo_bytes"):
        assert name not in product, name + " belongs to include/euler_gpu_measure.h"
    for name in sorted(declared):
        assert hasattr(L, name), "libeuler_gpu.so does not export " + name
        assert name in _lib.SIGNATURES, "python binding lacks " + name


def test_version_and_registry(L):
    assert b"gfx950" in L.euler_gpu_version()
    for op in (b"API_SAMPLE_NB", b"API_SAMPLE_NODE", b"ID_UNIQUE", b"IDX_GATHER",
               b"DATA_GATHER", b"API_GET_NB_NODE", b"API_GET_EDGE_SUM_WEIGHT",
               b"API_SAMPLE_ROOT", b"API_SAMPLE_L", b"API_LOCAL_SAMPLE_L", b"API_SPARSE_GEN_ADJ",
               b"API_SPARSE_GET_ADJ", b"API_GATHER_RESULT"):
        assert L.euler_op_registered(op) == 1
    assert L.euler_op_registered(b"API_NOT_THERE") == 0


def read_dat(path, shard_index=0, shards=1):
    from euler_amd import _lib
    L = _lib.lib()
    csr = _lib.HostCSR()
    parts = C.c_int32(0)
    owner = C.c_void_p()
    rc = L.euler_gpu_dat_open(str(path).encode(), shard_index, shards, C.byref(csr),
                              C.byref(parts), C.byref(owner))
    if rc != 0:
        raise RuntimeError(L.euler_gpu_last_error().decode())
    n, T = csr.n_rows, csr.n_edge_types
    out = dict(
        row_id=np.ctypeslib.as_array(csr.row_id, (n,)).copy(),
        row_ptr=np.ctypeslib.as_array(csr.row_ptr, (n + 1,)).copy(),
        type_end=np.ctypeslib.as_array(csr.type_end, (n * T,)).copy(),
        type_prefix=np.ctypeslib.as_array(csr.type_prefix, (n * T,)).copy(),
        node_type=np.ctypeslib.as_array(csr.node_type, (n,)).copy(),
        node_weight=np.ctypeslib.as_array(csr.node_weight, (n,)).copy(),
        n_types=T, n_node_types=csr.n_node_types, partitions=parts.value)
    E = int(out["row_ptr"][-1])
    out["nbr"] = np.ctypeslib.as_array(csr.nbr, (max(E, 1),))[:E].copy()
    out["prefix_w"] = np.ctypeslib.as_array(csr.prefix_w, (max(E, 1),))[:E].copy()
    F = csr.n_float_features
    out["n_float"] = F
    if F > 0:
        out["feat_ptr"] = np.ctypeslib.as_array(csr.feat_ptr, (n + 1,)).copy()
        out["feat_idx"] = np.ctypeslib.as_array(csr.feat_idx, (n * F,)).copy()
        tot = int(out["feat_ptr"][-1])
        out["feat_val"] = np.ctypeslib.as_array(csr.feat_val, (max(tot, 1),))[:tot].copy()
    U = csr.n_u64_features
    out["n_u64"] = U
    if U > 0:
        out["ufeat_ptr"] = np.ctypeslib.as_array(csr.ufeat_ptr, (n + 1,)).copy()
        out["ufeat_idx"] = np.ctypeslib.as_array(csr.ufeat_idx, (n * U,)).copy()
        tot = int(out["ufeat_ptr"][-1])
        out["ufeat_val"] = np.ctypeslib.as_array(csr.ufeat_val, (max(tot, 1),))[:tot].copy()
    L.euler_gpu_dat_close(owner)
    return out


def test_dat_reader_on_reference_tool_output(fixture_csr):
    """tests/golden/fixture_dat was written by the reference's euler/tools and
    fixture_graph.npz by the reference's own loader from the same files."""
    d = read_dat(os.path.join(ROOT, "tests", "golden", "fixture_dat"))
    assert d["partitions"] == 2 and d["n_types"] == 2 and d["n_node_types"] == 2
    order = np.argsort(d["row_id"])
    assert np.array_equal(d["row_id"][order], fixture_csr.row_id)
    assert np.array_equal(d["node_type"][order], fixture_csr.node_type)
    assert np.array_equal(d["node_weight"][order], fixture_csr.node_weight)
    T = 2
    for new, old in zip(order, range(6)):
        b, e = d["row_ptr"][new], d["row_ptr"][new + 1]
        fb, fe = fixture_csr.row_ptr[old], fixture_csr.row_ptr[old + 1]
        assert np.array_equal(d["nbr"][b:e], fixture_csr.nbr[fb:fe])
        assert np.array_equal(d["prefix_w"][b:e], fixture_csr.prefix_w[fb:fe])
        assert np.array_equal(d["type_end"][new * T:new * T + T],
                              fixture_csr.type_end[old * T:old * T + T])
        assert np.array_equal(d["type_prefix"][new * T:new * T + T],
                              fixture_csr.type_prefix[old * T:old * T + T])
    # float features of the records == what the reference's DeSerialize holds
    fg = np.load(os.path.join(ROOT, "tests", "golden", "features.npz"))
    F = int(fg["fx_n_float"])
    assert d["n_float"] == F
    for new, old in zip(order, range(6)):
        assert np.array_equal(d["feat_idx"][new * F:new * F + F],
                              fg["fx_feat_idx"][old * F:old * F + F])
        assert np.array_equal(d["feat_val"][d["feat_ptr"][new]:d["feat_ptr"][new + 1]],
                              fg["fx_feat_val"][fg["fx_feat_ptr"][old]:fg["fx_feat_ptr"][old + 1]])
    # ... and the uint64 ("sparse") features
    sg = np.load(os.path.join(ROOT, "tests", "golden", "sparse_features.npz"))
    U = int(sg["fx_n_u64"])
    assert d["n_u64"] == U and U > 0
    for new, old in zip(order, range(6)):
        assert np.array_equal(d["ufeat_idx"][new * U:new * U + U],
                              sg["fx_feat_idx"][old * U:old * U + U])
        assert np.array_equal(d["ufeat_val"][d["ufeat_ptr"][new]:d["ufeat_ptr"][new + 1]],
                              sg["fx_feat_val"][sg["fx_feat_ptr"][old]:sg["fx_feat_ptr"][old + 1]])
    # shard filter of Graph::Init: file idx % shards == shard_index
    s0 = read_dat(os.path.join(ROOT, "tests", "golden", "fixture_dat"), 0, 2)
    s1 = read_dat(os.path.join(ROOT, "tests", "golden", "fixture_dat"), 1, 2)
    assert sorted(s0["row_id"].tolist() + s1["row_id"].tolist()) == [1, 2, 3, 4, 5, 6]
    assert all(i % 2 == 0 for i in s0["row_id"]) and all(i % 2 == 1 for i in s1["row_id"])


def _vec(fmt, values):
    return struct.pack("<I", len(values)) + b"".join(struct.pack("<" + fmt, v) for v in values)


def _str(s):
    b = s.encode()
    return struct.pack("<I", len(b)) + b


def write_dat_dir(path, csr, partitions=2):
    """Write a CSR in the reference's on-disk layout (euler.meta:
    graph_builder.cc:230-307; node record: node.cc:414-526) - test helper."""
    path = str(path)
    os.makedirs(os.path.join(path, "Node"), exist_ok=True)
    T = csr.n_types
    n_nt = int(csr.node_type.max()) + 1
    meta = _str("g") + _str("1") + struct.pack("<QQi", csr.n_rows, len(csr.nbr), partitions)
    meta += struct.pack("<I", 0) + struct.pack("<I", 0)
    meta += struct.pack("<I", n_nt) + b"".join(_str(str(i)) + struct.pack("<I", i) for i in range(n_nt))
    meta += struct.pack("<I", T) + b"".join(_str(str(i)) + struct.pack("<I", i) for i in range(T))
    open(os.path.join(path, "euler.meta"), "wb").write(meta)
    files = [b"" for _ in range(partitions)]
    for r in range(csr.n_rows):
        b, e = csr.row_ptr[r], csr.row_ptr[r + 1]
        te = csr.type_end[r * T:(r + 1) * T]
        tp = csr.type_prefix[r * T:(r + 1) * T]
        tw = np.diff(np.concatenate([[0], tp])).astype(np.float32)
        rec = struct.pack("<Qif", int(csr.row_id[r]), int(csr.node_type[r]),
                          float(csr.node_weight[r]))
        rec += _vec("i", range(T)) + _vec("f", tw.tolist()) + _vec("i", te.tolist())
        rec += _vec("Q", csr.nbr[b:e].tolist()) + _vec("f", csr.prefix_w[b:e].tolist())
        rec += _vec("i", []) + _vec("f", []) + _vec("i", []) + _vec("Q", []) + _vec("f", [])
        rec += _vec("i", []) + _vec("Q", []) + _vec("i", []) + _vec("f", []) + _vec("i", []) + _vec("b", [])
        files[int(csr.row_id[r]) % partitions] += struct.pack("<I", len(rec)) + rec
    for p in range(partitions):
        open(os.path.join(path, "Node", "data_%d.dat" % p), "wb").write(files[p])


def test_dat_writer_helper_roundtrip_and_reference_loader(O, random_csr, tmp_path):
    """Integer-weight rows survive the type-weight diff exactly; the reference's
    own loader (oracle/_ref) reads the helper's files identically."""
    rng = np.random.default_rng(0)
    n, T = 50, 3
    ids = np.arange(10, 10 + n).astype(np.uint64)
    deg = rng.integers(0, 6, (n, T))
    seg = np.zeros(n * T + 1, np.int64)
    seg[1:] = np.cumsum(deg.reshape(-1))
    nbr = rng.choice(ids, int(seg[-1])).astype(np.uint64)
    w = rng.integers(1, 9, int(seg[-1])).astype(np.float32)
    csr = O.csr_from_raw(ids, seg, nbr, w, T, rng.integers(0, 2, n), np.ones(n))
    write_dat_dir(tmp_path, csr, partitions=4)
    d = read_dat(tmp_path)
    order = np.argsort(d["row_id"])
    assert np.array_equal(d["row_id"][order], ids)
    assert int(d["row_ptr"][-1]) == len(nbr)
    if O.have_ref():
        R = O.RefGraph.load(str(tmp_path), T)
        rc = R.export_csr(ids)
        assert np.array_equal(rc.nbr, csr.nbr)
        assert np.array_equal(rc.prefix_w, csr.prefix_w)
        assert np.array_equal(rc.type_prefix, csr.type_prefix)


def test_dat_reader_errors(tmp_path):
    with pytest.raises(RuntimeError, match="euler.meta"):
        read_dat(tmp_path / "missing")
    os.makedirs(tmp_path / "bad" / "Node")
    open(tmp_path / "bad" / "euler.meta", "wb").write(b"\x05\x00\x00\x00ab")
    with pytest.raises(RuntimeError, match="malformed"):
        read_dat(tmp_path / "bad")


def test_init_query_proxy_string_rules(L):
    # tf_euler/utils/init_query_proxy.cc:22-33: empty / malformed items -> false
    assert L.InitQueryProxy(b"") is False
    assert L.InitQueryProxy(b"mode") is False
    assert L.InitQueryProxy(b"mode=local;=x") is False
    assert L.InitQueryProxy(b"mode=local;data_path=") is False
    assert L.InitQueryProxy(b"mode=remote;zk_server=a:1;zk_path=/e") is False
    assert b"mode=local" in L.euler_gpu_last_error()
    assert L.InitQueryProxy(b"mode=local;data_path=/nonexistent/dir") is False


def test_synth_table_matches_oracle_fill(O):
    import euler_amd
    for n_nodes, n_edges in ((5000, 50000), (100000, 1000000), (1 << 14, 200000)):
        p = euler_amd.synth_params(1, n_nodes, n_edges)
        po = O.synth_params(1, n_nodes, n_edges)
        assert p.scale == po.scale
        a = np.array(list(p.deg_table))
        b = np.array(list(po.deg_table))
        assert np.allclose(a, b, rtol=1e-12, atol=0)


def test_host_only_entry_points(L):
    from euler_amd import ops
    assert L.euler_gpu_gen_pair_count(9, 2, 2) == 30
    counts = (C.c_int32 * 2)(25, 10)
    ws = L.euler_gpu_sample_fanout_workspace(1024, counts, 2)
    assert ws >= 1024 + 25600
    split = ops.sample_node_split(20240521, 1, 10, [1.0, 2.0, 0.0, 3.0])
    assert split.sum() == 10 and split[2] == 0


def test_sample_node_split_matches_oracle(O):
    from euler_amd import ops
    for call, sw in enumerate(([1.0, 2.0, 0.0, 3.0], [5, 1, 1, 1, 8], [0.3, 0.3, 0.6])):
        for count in (1, 10, 33):
            a = ops.sample_node_split(7, call, count, sw)
            b = O.sample_node_split(7, call, count, sw)
            assert np.array_equal(a, b)


def test_no_cpu_fallback():
    """Without a GPU every operator must raise, never compute on the host."""
    import torch
    import euler_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from euler_amd._lib import EulerGpuError
    with pytest.raises(RuntimeError):
        euler_amd.ops.gather(torch.zeros(3, 2), torch.zeros(2, dtype=torch.int32))
    with pytest.raises(EulerGpuError):
        euler_amd.Graph.synthetic(euler_amd.synth_params(1, 100, 1000))
    with pytest.raises(RuntimeError):
        euler_amd.sample_node(4, -1)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under euler_amd/ may name it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "euler_amd")):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".cc")) or fn == "Makefile":
                text = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, re.M), fn
                assert "libeuler_oracle" not in text and "_ref/" not in text, fn


def test_cpp_host_example_builds():
    """examples/cpp/sage_minibatch.cc - a C++ host over include/euler_gpu.h and the
    HIP runtime only - compiles and links against the in-tree library (running it
    needs a GPU: tests/test_layerwise_gpu.py::test_cpp_host_example)."""
    import shutil
    import subprocess
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    d = os.path.join(ROOT, "examples", "cpp")
    subprocess.check_call(["make", "-s", "-C", d])
    exe = os.path.join(d, "sage_minibatch")
    assert os.path.exists(exe)
    deps = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libeuler_gpu.so" in deps and "torch" not in deps and "python" not in deps.lower()


def test_makefile_lists_every_kernel_header():
    """Every header of euler_amd/csrc (and of include/) is a dependency of every object in the
    library's Makefile: a header left out of HDRS lets objects compiled against two versions of a
    record struct be linked into one library (WbRec in wb_index.h did exactly that once - a
    memory fault on the GPU, nothing at build time)."""
    import glob
    import re
    csrc = os.path.join(ROOT, "euler_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    hdrs = re.search(r"^HDRS :=(.*?)\n\n", mk, re.S | re.M).group(1)
    for h in glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")):
        assert os.path.basename(h) in hdrs, os.path.basename(h)

def test_dat_reader_many_partitions_keeps_file_order(O, tmp_path):
    """The partition files are parsed by several host threads; rows must still
    come out in file order (names sorted), every row intact, and a corrupt file
    must be reported whichever thread meets it."""
    rng = np.random.default_rng(5)
    n, T, parts = 3000, 2, 24
    ids = np.arange(1, n + 1).astype(np.uint64) * 5
    deg = rng.integers(0, 9, (n, T))
    seg = np.zeros(n * T + 1, np.int64)
    seg[1:] = np.cumsum(deg.reshape(-1))
    nbr = rng.choice(ids, int(seg[-1])).astype(np.uint64)
    w = rng.integers(1, 9, int(seg[-1])).astype(np.float32)
    csr = O.csr_from_raw(ids, seg, nbr, w, T, rng.integers(0, 2, n), np.ones(n))
    write_dat_dir(tmp_path, csr, partitions=parts)
    d = read_dat(tmp_path)
    want = []
    for name in sorted("data_%d.dat" % p for p in range(parts)):
        p = int(name.split("_")[1].split(".")[0])
        want.extend(int(v) for v in ids if int(v) % parts == p)
    assert d["row_id"].tolist() == want
    pos = {int(v): i for i, v in enumerate(csr.row_id)}
    for new in rng.choice(n, 200, replace=False):
        old = pos[int(d["row_id"][new])]
        b, e = d["row_ptr"][new], d["row_ptr"][new + 1]
        assert np.array_equal(d["nbr"][b:e], csr.nbr[csr.row_ptr[old]:csr.row_ptr[old + 1]])
        assert np.array_equal(d["prefix_w"][b:e],
                              csr.prefix_w[csr.row_ptr[old]:csr.row_ptr[old + 1]])
        assert np.array_equal(d["type_end"][new * T:new * T + T], csr.type_end[old * T:old * T + T])
    # shards still select whole files
    s1 = read_dat(tmp_path, 1, 3)
    assert all(int(v) % parts % 3 == 1 for v in s1["row_id"])
    with open(os.path.join(str(tmp_path), "Node", "data_17.dat"), "r+b") as f:
        f.truncate(os.path.getsize(f.name) - 3)
    with pytest.raises(RuntimeError, match="data_17"):
        read_dat(tmp_path)


def test_edge_records_match_node_rows(L, tmp_path):
    """euler_gpu_dat_verify_edges on the reference tool's own output: every Edge
    record is an entry of its source's row and the rows hold nothing else - the
    condition under which EdgeExist-from-rows equals the reference's EdgeExist."""
    import shutil
    dat = os.path.join(ROOT, "tests", "golden", "fixture_dat")
    rec, miss, trip = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    assert L.euler_gpu_dat_verify_edges(dat.encode(), 0, 1, C.byref(rec), C.byref(miss),
                                        C.byref(trip)) == 0
    assert rec.value > 0 and miss.value == 0 and rec.value == trip.value
    for shard in (0, 1):          # per shard the same holds (files split by partition)
        assert L.euler_gpu_dat_verify_edges(dat.encode(), shard, 2, C.byref(rec), C.byref(miss),
                                            C.byref(trip)) == 0
        assert miss.value == 0 and rec.value == trip.value
    # an inconsistent dataset is reported: drop one Edge partition
    bad = str(tmp_path / "bad")
    shutil.copytree(dat, bad)
    open(os.path.join(bad, "Edge", "data_1.dat"), "wb").close()
    assert L.euler_gpu_dat_verify_edges(bad.encode(), 0, 1, C.byref(rec), C.byref(miss),
                                        C.byref(trip)) == 0
    assert rec.value < trip.value
    # ... and InitQueryProxy(verify_edges=1) refuses it before touching the GPU
    assert L.InitQueryProxy(("mode=local;data_path=%s;verify_edges=1" % bad).encode()) is False
    assert b"disagree" in L.euler_gpu_last_error()
    shutil.rmtree(os.path.join(bad, "Edge"))
    assert L.euler_gpu_dat_verify_edges(bad.encode(), 0, 1, C.byref(rec), C.byref(miss),
                                        C.byref(trip)) != 0


def test_neighbor_dataflow_block_arithmetic():
    """NeighborDataFlow.produce_subgraph (tf_euler/python/dataflow/
    neighbor_dataflow.py:45-75: no dedup between hops) is index arithmetic only:
    checked on the CPU against the reference's formulas written out by hand."""
    import torch
    from euler_amd.dataflow import NeighborDataFlow

    class Fixed(NeighborDataFlow):
        def get_neighbors(self, n_id):
            return ([torch.tensor([7, 8, 9, 7]), torch.tensor([1, 2, 3, 4, 5, 6, 7, 8, 9])],
                    [torch.tensor([0, 0, 1, 1]), torch.tensor([0, 1, 2, 3, 0, 1, 2, 3, 4])])

    roots = torch.tensor([10, 11])
    for loops in (True, False):
        df = Fixed(2, add_self_loops=loops)(roots)
        n_id = roots.numpy()
        last_idx = np.arange(2)
        nbrs = [np.array([7, 8, 9, 7]), np.array([1, 2, 3, 4, 5, 6, 7, 8, 9])]
        srcs = [np.array([0, 0, 1, 1]), np.array([0, 1, 2, 3, 0, 1, 2, 3, 4])]
        for i, blk in enumerate(df.blocks):
            new_n_id = np.concatenate([nbrs[i], n_id])
            new_inv = np.arange(len(new_n_id))
            res = new_inv[-len(n_id):]
            src = srcs[i]
            if loops:
                src = np.concatenate([src, last_idx])
                last_idx = new_inv
            else:
                new_inv = new_inv[:-len(n_id)]
                last_idx = new_inv
            assert np.array_equal(blk.n_id.numpy(), new_n_id)
            assert np.array_equal(blk.res_n_id.numpy(), res)
            assert np.array_equal(blk.edge_index.numpy(), np.stack([src, new_inv]))
            assert blk.size == [len(n_id), len(new_n_id)]
            n_id = new_n_id
        assert len(df) == 2 and [b.size for b in df] == [b.size for b in df.blocks[::-1]]


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` with no launcher around it starts its own N ranks
    (torch.distributed.run); with fewer GPUs visible than ranks it must refuse loudly
    instead of running one rank and printing n_gpus: 1 (here: no GPU at all)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 2, (out.returncode, out.stderr[-500:])
    assert "--gpus 2 but only" in out.stderr and "{" not in out.stdout


def test_python_surface_argument_caches():
    """Graph methods convert their small list arguments (edge types, counts) once: the same
    lists come back every minibatch.  Cached entries are value-keyed, ragged / array inputs
    bypass the cache, and the tuning keys retired in round 3 answer EINVAL."""
    from euler_amd import graph as g, _lib
    a = g._i32_array([1, 4, 6])
    assert a is g._i32_array([1, 4, 6]) and a is g._i32_array((1, 4, 6))
    assert a[2] == 3 and a[0].dtype == np.int32 and a[0].tolist() == [1, 4, 6]
    assert g._i32_array([1, 4, 7])[0].tolist() == [1, 4, 7]
    assert g._i32_array([])[2] == 0
    arr = np.array([[0, 1], [2, 3]], np.int64)
    assert g._i32_array(arr)[0].tolist() == [0, 1, 2, 3]          # arrays: converted, not cached
    assert g._i32_array([True, 2])[0].tolist() == [1, 2]           # (bools are not cache keys)
    p = g._fanout_plan([[0], [0]], [25, 10])
    assert p is g._fanout_plan([[0], [0]], [25, 10]) and p is g._fanout_plan(((0,), (0,)), (25, 10))
    assert p[2] == 1 and p[0].shape == (2, 1) and p[3].tolist() == [25, 10]
    q = g._fanout_plan([[0, 1], [2, 3]], [3, 4])
    assert q[2] == 2 and q[0].tolist() == [[0, 1], [2, 3]]
    f = g._fanout_plan([0, 0], [25, 10])                           # flat list of single types
    assert f[2] == 1 and f[0].tolist() == [[0], [0]]
    L = _lib.lib()
    for key in (1, 6, 19, 21, 22, 26):
        assert L.euler_gpu_set_tuning(key, 0) != 0
    for key, v in ((0, 6), (27, 1), (38, 262144), (43, 12), (44, 1)):
        assert L.euler_gpu_set_tuning(key, v) == 0


def test_block_construction_workspace_sizes():
    """euler_gpu_sage_blocks[_multi]_workspace are host arithmetic (no GPU): the multi form's
    workspace holds the single call's (its fallback - hops with type draws - runs the separate
    calls in it), grows with the number of minibatches, and is M copies of a hop's arrays."""
    import ctypes as C
    from euler_amd import _lib
    L = _lib.lib()
    for n, fanouts in ((1024, [25, 10]), (300, [5, 4, 3]), (1, [1])):
        fan = (C.c_int32 * len(fanouts))(*fanouts)
        one = L.euler_gpu_sage_blocks_workspace(n, fan, len(fanouts))
        sizes = [L.euler_gpu_sage_blocks_multi_workspace(m, n, fan, len(fanouts)) for m in (1, 2, 8, 64)]
        assert one > 0 and sizes[0] >= one
        assert all(b >= a for a, b in zip(sizes, sizes[1:]))      # (tiny flows: the single call's size rules)
        if n >= 300:
            assert sizes[3] > sizes[2] > sizes[0]
            assert sizes[3] <= 9 * sizes[2]      # ~linear in M

