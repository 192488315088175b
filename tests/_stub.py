"""Builds tests/stub_rccl/rccl_stub.cpp (a shared-memory stand-in for the RCCL entry points libalpine_hip.so calls; see
that file's header) into tests/stub_rccl/_build/ with g++.  Test infrastructure only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "stub_rccl", "rccl_stub.cpp")
OUT_DIR = os.path.join(HERE, "stub_rccl", "_build")
LIB = os.path.join(OUT_DIR, "librccl_stub.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
CXXFLAGS = ["-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", f"-I{ROCM}/include"]


def build_rccl_stub() -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        tmp = LIB + f".{os.getpid()}.tmp"
        subprocess.run(["g++", *CXXFLAGS, "-shared", "-fPIC", SRC, "-o", tmp, "-ldl", "-lrt", "-pthread"], check=True)
        os.replace(tmp, LIB)
    return LIB


def build_selftest() -> str:
    """The stand-in's own check: R forked processes all-reduce host buffers through it (memcpy in place of the HIP copies)."""
    lib = build_rccl_stub()
    exe = os.path.join(OUT_DIR, "stub_selftest")
    src = os.path.join(HERE, "stub_rccl", "selftest.cpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(lib)):
        subprocess.run(["g++", *CXXFLAGS, src, "-o", exe, "-rdynamic", f"-L{OUT_DIR}", "-lrccl_stub", f"-Wl,-rpath,{OUT_DIR}", "-ldl"], check=True)
    return exe
