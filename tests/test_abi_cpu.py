"""CPU checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/tike_amd.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import tike_amd._lib as L


def test_library_loads_and_exports_every_declared_symbol():
    names = L.declared_symbols()
    assert len(names) >= 10
    lib = ctypes.CDLL(L.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_every_declared_symbol_has_a_python_prototype():
    assert sorted(L._PROTOTYPES) == L.declared_symbols()


def test_prototype_arity_matches_header():
    text = open(L.HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, args in L._PROTOTYPES.items():
        m = re.search(rf"int\s+{name}\s*\(([^)]*)\)", text)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else len(params.split(","))
        assert n == len(args), (name, n, len(args))


def test_no_product_module_imports_the_oracle():
    root = os.path.join(os.path.dirname(L.__file__))
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src,
                                     flags=re.M), os.path.join(dirpath, f)


def test_every_entry_point_is_documented():
    """INTEGRATION.md names every extern "C" entry of include/tike_amd.h (the
    table that maps each one to the reference lines it replaces)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "tike_amd.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    names = sorted(set(re.findall(r"\bint (tike_[a-z0-9_]+)\(", header)))
    assert names, "no entry points found"
    missing = [n for n in names if n not in doc]
    assert not missing, missing


def test_comm_entries_reject_bad_arguments_without_a_gpu():
    """Argument validation of the collective entries happens before RCCL is
    bound, so it can be checked on a CPU-only host."""
    comm = ctypes.c_void_p()
    ident = ctypes.create_string_buffer(L.COMM_ID_BYTES)
    lib = L.lib
    assert lib.tike_comm_unique_id(None) == L.ERR_ARG
    assert lib.tike_comm_create(None, 1, 0, ctypes.byref(comm)) == L.ERR_ARG
    assert lib.tike_comm_create(ident, 2, 2, ctypes.byref(comm)) == L.ERR_ARG
    assert lib.tike_comm_create(ident, 0, 0, ctypes.byref(comm)) == L.ERR_ARG
    assert lib.tike_comm_allreduce_sum(None, None, 4, 0, None) == L.ERR_ARG
    assert lib.tike_comm_broadcast(None, None, 4, 0, None) == L.ERR_ARG
    assert lib.tike_comm_destroy(None) == 0


def test_abi_version_of_header_library_and_binding_agree():
    """`tike_abi_version()` lets a binding detect a library built from another
    header before it passes 31 positional arguments to the wrong entry."""
    text = open(L.HEADER_PATH).read()
    m = re.search(r"^#define\s+TIKE_ABI_VERSION\s+(\d+)", text, flags=re.M)
    assert m, "include/tike_amd.h must define TIKE_ABI_VERSION"
    lib = ctypes.CDLL(L.LIB_PATH)
    lib.tike_abi_version.restype = ctypes.c_int
    assert lib.tike_abi_version() == int(m.group(1)) == L.ABI_VERSION


def test_binding_refuses_a_library_of_another_abi_version(monkeypatch):
    monkeypatch.setattr(L, "ABI_VERSION", L.ABI_VERSION + 1)
    import pytest
    with pytest.raises(ImportError, match="ABI version"):
        L._check_abi_version()


import pytest as _pytest


@_pytest.mark.parametrize("harness", ["test_fft_radix", "test_fft_mixed"])
def test_fft_radix_host_unit_test_builds_and_passes(tmp_path, harness):
    """csrc/fft_radix.h and csrc/fft_mixed.h say "host-unit-tested": this is
    the harness that runs tests/csrc/test_fft_radix.cpp (every butterfly,
    radices 2..32 incl. 3, 5, 7, 11, 13) and tests/csrc/test_fft_mixed.cpp (the
    planner and the stage arithmetic of the shape-general engine for 35 sizes
    up to 4096) with plain g++, no GPU."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        import pytest
        pytest.skip("no g++")
    here = os.path.dirname(os.path.abspath(__file__))
    src = os.path.join(here, "csrc", harness + ".cpp")
    exe = str(tmp_path / harness)
    cmd = ["g++", "-O2", "-std=c++17", "-I",
           os.path.join(os.path.dirname(here), "tike_amd", "csrc"), src, "-o",
           exe]
    subprocess.run(cmd, check=True, capture_output=True, timeout=300)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr


def test_cluster_entries_on_the_host():
    """tike_cluster_farthest_fill / tike_cluster_swap_sweep are host code: a
    known small case by hand, and TIKE_ERR_ARG for unusable arguments."""
    import numpy as np
    lib = L.lib
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    # five points on a line, clusters 0 and 1 seeded at points 2 and 3:
    # cluster 0 (mean 2) takes the farthest free point -- 0 and 4 tie, the
    # first wins --, cluster 1 (mean 3) then takes 1 (2 away; 4 is 1 away),
    # cluster 0 (mean 1) takes what is left
    pts = np.array([[0, 0], [1, 0], [2, 0], [3, 0], [4, 0]], np.float32)
    owner = np.array([-1, -1, 0, 1, -1], np.int64)
    assert lib.tike_cluster_farthest_fill(ptr(pts), 5, ptr(owner), 2, 3) == 0
    assert owner.tolist() == [0, 1, 0, 1, 0]
    assert lib.tike_cluster_farthest_fill(ptr(pts), 5, ptr(owner), 2,
                                          1) == 1000001  # nothing free
    assert lib.tike_cluster_farthest_fill(None, 5, ptr(owner), 2, 0) == 1000001
    # two clusters on a line whose labels are crossed: one swap repairs them
    dist = np.array([[0., 3.], [1., 2.], [2., 1.], [3., 0.]])  # to centroids
    label = np.array([0, 1, 0, 1], np.int64)
    best = dist.argmin(axis=1).astype(np.int64)
    regret = dist[np.arange(4), best] - dist[np.arange(4), label]
    order = np.argsort(regret).astype(np.int64)
    moved = ctypes.c_int(0)
    assert lib.tike_cluster_swap_sweep(ptr(dist), 4, 2, ptr(label), ptr(best),
                                       ptr(order), ptr(regret),
                                       ctypes.byref(moved)) == 0
    assert moved.value == 1 and label.tolist() == [0, 0, 1, 1]
    assert np.all(regret == 0)
    assert lib.tike_cluster_swap_sweep(ptr(dist), 0, 2, ptr(label), ptr(best),
                                       ptr(order), ptr(regret),
                                       ctypes.byref(moved)) == 1000001


def test_build_id_is_the_hash_of_the_sources_in_the_tree():
    """`tike_build_id()` = sha256 over the sources in the Makefile's order: the
    loaded library was built from THIS tree (profiles/*_pmc_traffic_*.json
    carry the id; bench.py quotes a traffic figure only for the same build)."""
    import hashlib
    import re
    csrc = os.path.join(os.path.dirname(L.LIB_PATH))
    mk = open(os.path.join(csrc, "Makefile")).read()
    srcs = re.search(r"^SRCS = (.*)$", mk, flags=re.M).group(1).split()
    hdrs = re.search(r"^HDRS = (.*)$", mk, flags=re.M).group(1).split()
    h = hashlib.sha256()
    for name in srcs + hdrs + ["cluster_host.cpp"]:
        h.update(open(os.path.join(csrc, name), "rb").read())
    assert L.build_id() == h.hexdigest()
