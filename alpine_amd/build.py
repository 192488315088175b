"""Build libalpine_hip.so (gfx950 only) in-tree with hipcc.  Used by __graft_entry__.build()."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "alpine_hip.hip")
DEPS = [os.path.join(HERE, "csrc", f) for f in sorted(os.listdir(os.path.join(HERE, "csrc")))] + \
       [os.path.join(os.path.dirname(HERE), "include", "alpine_hip.h")]
LIB = os.path.join(HERE, "libalpine_hip.so")
# diagnostics build (-DALPINE_DIAGNOSTICS): the timing-only ablations of tools/ (wrong results by design) exist only here
LIB_DIAG = os.path.join(HERE, "libalpine_hip_diag.so")
# throwaway build with in-kernel time stamps (-DALPINE_STAMPS) for tools/stamps.py
LIB_STAMPS = os.path.join(HERE, "libalpine_hip_stamps.so")
ROCM_LIB = "/opt/rocm/lib"


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libalpine_hip.so cannot be built")


def is_stale(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_library(force: bool = False, verbose: bool = False, extra_flags=(), diagnostics: bool = False, stamps: bool = False) -> str:
    """hipcc -> libalpine_hip.so (links libamdhip64 and librccl; in a torch process both resolve to the copies torch has
    already loaded, same sonames).  diagnostics=True builds libalpine_hip_diag.so with the ablation knobs compiled in."""
    lib = LIB_STAMPS if stamps else (LIB_DIAG if diagnostics else LIB)
    if stamps:
        extra_flags = tuple(extra_flags) + ("-DALPINE_STAMPS",)
    if not force and not is_stale(lib):
        return lib
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function",
           # no SLP vectorisation: v_pk_add_f32 / v_pk_fma_f32 formed from adjacent scalar float ops are slower than the
           # scalar forms beside MFMAs (MI355X_MICROARCH.md, filler prices) and need register pairs (v_mov copies)
           "-fno-slp-vectorize", *(("-DALPINE_DIAGNOSTICS",) if diagnostics else ()), *extra_flags,
           "-o", lib, SRC, f"-L{ROCM_LIB}", "-lrccl", f"-Wl,-rpath,{ROCM_LIB}"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    import sys
    print(build_library(force="--force" in sys.argv, verbose=True, diagnostics="--diag" in sys.argv, stamps="--stamps" in sys.argv))
