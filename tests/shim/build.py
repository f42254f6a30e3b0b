"""Builds tests/shim/shmccl.cpp (the shared-memory stand-in for librccl on one-GPU boxes: test infrastructure) in-tree."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "shmccl.cpp")
OUT = os.path.join(HERE, "_build", "libshmccl.so")


def library(force: bool = False) -> str:
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        tmp = OUT + ".tmp.%d" % os.getpid()
        subprocess.check_call([hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "--offload-arch=gfx950", "-o", tmp, SRC, "-lrt"])
        os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    print(library(force=True))
