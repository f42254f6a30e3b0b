"""CPU: the shared-memory collective stand-in (tests/shim/shmccl.cpp, test infrastructure for tests/test_gpu_multirank.py) builds and exports
every entry point libdeepmod_hip looks up in its collective library (deepmod_amd/csrc/deepmod_hip.hip load_rccl)."""
import ctypes
import os
import re

from conftest import ROOT
from shim import build as shim_build


def test_shim_exports_what_the_product_binds():
    src = open(os.path.join(ROOT, "deepmod_amd", "csrc", "deepmod_hip.hip")).read()
    wanted = set(re.findall(r'bind\("(nccl[A-Za-z]+)"', src))
    assert {"ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclReduce", "ncclReduceScatter", "ncclCommDestroy", "ncclGetVersion"} <= wanted
    lib = ctypes.CDLL(shim_build.library())
    for name in wanted:
        assert hasattr(lib, name), name
    ids = [ctypes.create_string_buffer(128) for _ in range(2)]
    assert all(lib.ncclGetUniqueId(i) == 0 for i in ids)
    assert ids[0].raw != ids[1].raw and ids[0].raw.startswith(b"/deepmod_shmccl_")
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("deepmod_shmccl_")]


def test_collective_library_is_bound_through_its_own_header():
    """Round 6: no hand-declared RCCL ABI.  The product and the stand-in both include <rccl/rccl.h>; every function pointer of the
    product is decltype(&ncclX) of the header's prototype, the data types / operations it passes are the header's enumerators, and the stand-in DEFINES
    the header's functions (a drifted signature is a compile error there: build.library() above compiles it).  The constants a CUDA-era copy would have
    hard-coded are checked against the header text itself."""
    src = open(os.path.join(ROOT, "deepmod_amd", "csrc", "deepmod_hip.hip")).read()
    shim = open(os.path.join(ROOT, "tests", "shim", "shmccl.cpp")).read()
    assert "#include <rccl/rccl.h>" in src and "#include <rccl/rccl.h>" in shim
    assert "struct NcclId" not in src and "NCCL_INT32" not in src and "typedef int (*fn_" not in src
    for fn in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclReduce", "ncclReduceScatter", "ncclCommDestroy", "ncclGetErrorString", "ncclGetVersion"):
        assert "decltype(&%s)" % fn in src, fn
        assert re.search(r"^(ncclResult_t|const char\*) %s\(" % fn, shim, flags=re.M), fn
    assert "static_assert(sizeof(ncclUniqueId) == 128" in src
    assert "ncclInt32, ncclSum" in src and "ncclFloat64, ncclMax" in src
    header = open("/opt/rocm/include/rccl/rccl.h").read()
    assert re.search(r"#define NCCL_UNIQUE_ID_BYTES 128", header)
    assert re.search(r"ncclInt32\s*=\s*2", header) and re.search(r"ncclFloat64\s*=\s*8", header) and re.search(r"ncclSum\s*=\s*0", header) and re.search(r"ncclMax\s*=\s*2", header)


def test_bound_collective_library_is_reported():
    """dm_rccl_info: the file the loader mapped and ncclGetVersion's code - librccl needs no GPU to load.  With DEEPMOD_RCCL_LIBRARY the stand-in is what is
    reported (checked in a child process: the library binds once per process)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from deepmod_amd import comm; p, v = comm.rccl_info(); print(p); print(v)" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout.split()
    assert "librccl" in out[0] and int(out[1]) >= 20000
    env = dict(os.environ, DEEPMOD_RCCL_LIBRARY=shim_build.library())
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True, env=env).stdout.split()
    assert out[0].endswith("libshmccl.so") and int(out[1]) >= 20000
