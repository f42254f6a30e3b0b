"""CPU: the shared-memory collective stand-in (tests/shim/shmccl.cpp, test infrastructure for tests/test_gpu_multirank.py) builds and exports
every entry point libdeepmod_hip looks up in its collective library (deepmod_amd/csrc/deepmod_hip.hip load_rccl)."""
import ctypes
import os
import re

from conftest import ROOT
from shim import build as shim_build


def test_shim_exports_what_the_product_binds():
    src = open(os.path.join(ROOT, "deepmod_amd", "csrc", "deepmod_hip.hip")).read()
    wanted = set(re.findall(r'dlsym\(h, "(nccl[A-Za-z]+)"\)', src))
    assert {"ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclReduce", "ncclReduceScatter", "ncclCommDestroy"} <= wanted
    lib = ctypes.CDLL(shim_build.library())
    for name in wanted:
        assert hasattr(lib, name), name
    ids = [ctypes.create_string_buffer(128) for _ in range(2)]
    assert all(lib.ncclGetUniqueId(i) == 0 for i in ids)
    assert ids[0].raw != ids[1].raw and ids[0].raw.startswith(b"/deepmod_shmccl_")
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("deepmod_shmccl_")]
