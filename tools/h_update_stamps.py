"""Where does a block of h_update_mfma_kernel spend its time?  In-kernel stamps (s_memrealtime, 10 ns ticks) in a THROWAWAY
build of the library: `build` patches a copy of alpine_amd/csrc under /tmp with a `g_stamps` device array and STAMP(i) marks
at the phase boundaries of the H update and compiles tools/libalpine_stamp.so (needs hipcc, no GPU); `run` (on the GPU box)
loads that library through ALPINE_HIP_LIBRARY, runs a few iterations of the cfg3 workload at --cells and prints the median
duration of every phase over the blocks of the last H update.  The product library carries no stamps.

    python tools/h_update_stamps.py build            # build container
    gpurun -- python tools/h_update_stamps.py run 25000
    gpurun -- python tools/h_update_stamps.py sweepclock [x_scale]      # in-kernel shader clock of the x3 / x3w sweeps

`sweepclock`: the same throwaway build also stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) at the start of
every sweep workgroup and after its last flush; after >= 2 s of back-to-back iterations the in-kernel clock is
delta(s_memtime) / delta(s_memrealtime) x 100 MHz, median over the workgroups of the last launch (MI355X_MICROARCH.md, DVFS
give-back item 6).  x_scale = 1 (counts, 32x32x16 form) or 0.3712345 (full significands, x3w).

Round 2, 25 000 cells (one of 8 shards of cfg3), per block of 128 cells: fills + barrier 2.8 us | pieces (11 per tile) + H tile
10.9 | 2W^TW.H on the MFMA 2.8 | guided terms + update 9.3 -> 8.1 (group-uniform loops) | store 1.0 | barrier + H H^T partial 3.6 |
statistics 8.2 -> 6.6 (DPP sums) | total 40 -> 36 us; the MFMA phase shows the clock: 64 MFMAs x 64 cycles in 2.7 us = 1.5 GHz."""
import ctypes as C
import os
import shutil
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "tools", "libalpine_stamp.so")
MARKS = [   # (text to find inside h_update_mfma_kernel, replacement)
    ("    const bool y_lds = tail.ybuf_rows > 0;", "    const bool y_lds = tail.ybuf_rows > 0;\n    STAMP(0);"),
    ("    for (int idx = tid; idx < nB; idx += 256) Bl[idx] = B[idx];\n    __syncthreads();\n",
     "    for (int idx = tid; idx < nB; idx += 256) Bl[idx] = B[idx];\n    __syncthreads();\n    STAMP(1);\n"),
    ("                tile_raw_to_cd<KT>(hraw, tr, lane, hreg);\n            }\n",
     "                tile_raw_to_cd<KT>(hraw, tr, lane, hreg);\n            }\n            STAMP(2);\n"),
    ("            // numerator = 2 W^TX (+ guided)", "            asm volatile(\"s_nop 0\" :: \"v\"(acc[0][0]));\n            STAMP(3);\n            // numerator = 2 W^TX (+ guided)"),
    ("            (void)K;", "            (void)K;\n            STAMP(4);"),
    ("        __syncthreads();                                                       // the four tiles of this group are in LDS\n",
     "        STAMP(5);\n        __syncthreads();                                                       // the four tiles of this group are in LDS\n"),
    ("        if (meta.n_cov > 0) {\n            if (y_lds) {", "        STAMP(6);\n        if (meta.n_cov > 0) {\n            if (y_lds) {"),
    ("                             hs_smem, tid >> 7, grp);\n            }\n        }\n    }\n}",
     "                             hs_smem, tid >> 7, grp);\n            }\n        }\n        STAMP(7);\n    }\n}"),
]
SWEEP_MARKS = [   # (kernel name, start anchor, end anchor) in kernels_x3.hpp
    ("void stream_gemm_x3_kernel(", "    const int first_tile = (int)(pos / g.R);\n",
     "                sg_flush_tile<KT>(flush_tr[wave], d, out + (int64_t)(128 * hf + tt) * KP, 4 * KP, lane);\n            }\n        }\n"),
    ("void stream_gemm_x3w_kernel(", "    const int first_tile = (int)(pos / g.R);\n",
     "                sg_flush_tile16<KT>(flush_tr[wave], d, out + (int64_t)(64 * cg + tt) * KP, 4 * KP, lane);\n            }\n        }\n"),
]
PHASES = ["Y / 2W^TW / B fills + barrier", "H issue + pieces + H to C/D", "2W^TW.H on the MFMA", "guided terms + update", "tile store",
          "barrier + H H^T partial", "covariate statistics"]


def build(wxor=0):
    """wxor != 0: the sweep workgroups take the work range of blockIdx ^ wxor (same results; moves every range to another XCD)."""
    global LIB
    if wxor:
        LIB = LIB.replace(".so", f"_x{wxor}.so")
    work = "/tmp/alpine_stamp_build"
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(os.path.join(work, "alpine_amd"))
    shutil.copytree(os.path.join(REPO, "alpine_amd", "csrc"), os.path.join(work, "alpine_amd", "csrc"))
    shutil.copytree(os.path.join(REPO, "include"), os.path.join(work, "include"))
    kp = os.path.join(work, "alpine_amd", "csrc", "kernels.hpp")
    s = open(kp).read()
    s = s.replace("namespace alpine {\n\ntypedef float f32x4",
                  "namespace alpine {\n__device__ unsigned long long g_stamps[16 * 8192];\n#define STAMP(i) do { if (threadIdx.x == 0) "
                  "g_stamps[16 * blockIdx.x + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)\n\ntypedef float f32x4", 1)
    k = s.index("void h_update_mfma_kernel(")
    body = s[k:]
    for a, b in MARKS:
        assert a in body, f"source changed, mark not found: {a[:50]!r}"
        body = body.replace(a, b, 1)
    open(kp, "w").write(s[:k] + body)
    xp = os.path.join(work, "alpine_amd", "csrc", "kernels_x3.hpp")
    x = open(xp).read()
    x = x.replace("namespace alpine {\n", "namespace alpine {\n__device__ unsigned long long g_clk[4 * 1024];\n", 1)
    for name, a0, a1 in SWEEP_MARKS:
        k = x.index(name)
        body = x[k:]
        assert a0 in body and a1 in body, f"source changed: {name}"
        if wxor:
            assert "    const int w = blockIdx.x;\n" in body
            body = body.replace("    const int w = blockIdx.x;\n", f"    const int w = blockIdx.x ^ {wxor};\n", 1)
        body = body.replace(a0, a0 + "    const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();\n", 1)
        body = body.replace(a1, a1 + "        if (tid == 0) { g_clk[4 * (blockIdx.x & 1023)] = __builtin_amdgcn_s_memtime() - clk_t0; "
                                     "const unsigned long long clk_r1 = __builtin_amdgcn_s_memrealtime(); "
                                     "g_clk[4 * (blockIdx.x & 1023) + 1] = clk_r1 - clk_r0; g_clk[4 * (blockIdx.x & 1023) + 2] = clk_r0; "
                                     "unsigned xcc; asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\" : \"=s\"(xcc)); "
                                     "g_clk[4 * (blockIdx.x & 1023) + 3] = (unsigned long long)(xcc & 15u); }\n", 1)
        x = x[:k] + body
    open(xp, "w").write(x)
    hp = os.path.join(work, "alpine_amd", "csrc", "alpine_hip.hip")
    open(hp, "a").write('\nextern "C" int alpine_debug_read_stamps(unsigned long long* host, int n)\n{\n    return (int)hipMemcpyFromSymbol('
                        'host, HIP_SYMBOL(alpine::g_stamps), sizeof(unsigned long long) * n);\n}\n'
                        'extern "C" int alpine_debug_read_clk(unsigned long long* host, int n)\n{\n    return (int)hipMemcpyFromSymbol('
                        'host, HIP_SYMBOL(alpine::g_clk), sizeof(unsigned long long) * n);\n}\n')
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function",
                    "-fno-slp-vectorize", "-o", LIB, hp, "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    print(LIB)


def run(cells):
    os.environ["ALPINE_HIP_LIBRARY"] = LIB
    sys.path.insert(0, REPO)
    import torch
    import bench
    from alpine_amd import _native
    from alpine_amd.datasets import synth_counts_device_chunks
    from alpine_amd.model import draw_initial_factors
    wl = dict(bench.WORKLOADS["cfg3"])
    G, N, ku, kcov = wl["genes"], cells, wl["ku"], wl["kcov"]
    dev = torch.device("cuda", 0)
    W0, H0, B0 = draw_initial_factors(42, 1e-6, G, N, kcov + [ku], [2, 2])
    eng = _native.NativeShard(n_genes=G, n_cells=N, n_components=ku, cov_components=kcov, cov_levels=[2, 2], lam=[1e3, 1e3],
                              orth_W=wl["orth_W"], alpha_W=wl["alpha_W"], l1_ratio_W=wl["l1_ratio_W"], x_dtype="x3")
    for off, chunk in synth_counts_device_chunks(N, G, rank=ku, seed=0, device=dev, chunk_cells=8192):
        torch.cuda.synchronize()
        eng.upload_X_device(chunk.data_ptr(), chunk.stride(0), chunk.shape[0], _native.X_CELLS_BY_GENES, off)
        eng.synchronize()
    eng.finalize_X()
    for i in range(2):
        eng.upload_Y(i, bench.labels_onehot(N, seed=1 + i))
    eng.set_factors(W0, H0, B0)
    eng.run(10, with_loss=True)
    eng.synchronize()
    lib = _native.load()
    nb = (N + 127) // 128
    buf = (C.c_ulonglong * (16 * nb))()
    lib.alpine_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
    assert lib.alpine_debug_read_stamps(buf, 16 * nb) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 16)[:, :8].astype(np.int64)
    d = np.diff(a, axis=1) / 100.0
    print(f"cells {N}: {nb} blocks, kernel span (first block start -> last block end) {(a[:, 7].max() - a[:, 0].min()) / 100.0:.1f} us")
    for i, n in enumerate(PHASES):
        print(f"  {n:32s} median {np.median(d[:, i]):6.2f} us   p90 {np.percentile(d[:, i], 90):6.2f}")
    print(f"  block total: median {np.median((a[:, 7] - a[:, 0]) / 100.0):.2f} us; start skew p90 - p10 {(np.percentile(a[:, 0], 90) - np.percentile(a[:, 0], 10)) / 100.0:.2f} us")
    eng.close()


def sweepclock(x_scale, wxor=0):
    import time
    os.environ["ALPINE_HIP_LIBRARY"] = LIB.replace(".so", f"_x{wxor}.so") if wxor else LIB
    sys.path.insert(0, REPO)
    import torch
    import bench
    from alpine_amd import _native
    from alpine_amd.datasets import synth_counts_device_chunks
    from alpine_amd.model import draw_initial_factors
    wl = dict(bench.WORKLOADS["cfg3"])
    G, N, ku, kcov = wl["genes"], wl["cells"], wl["ku"], wl["kcov"]
    dev = torch.device("cuda", 0)
    W0, H0, B0 = draw_initial_factors(42, 1e-6, G, N, kcov + [ku], [2, 2])
    eng = _native.NativeShard(n_genes=G, n_cells=N, n_components=ku, cov_components=kcov, cov_levels=[2, 2], lam=[1e3, 1e3],
                              orth_W=wl["orth_W"], alpha_W=wl["alpha_W"], l1_ratio_W=wl["l1_ratio_W"], x_dtype="x3")
    for off, chunk in synth_counts_device_chunks(N, G, rank=ku, seed=0, device=dev, chunk_cells=8192):
        if x_scale != 1.0:
            chunk = (chunk * x_scale).contiguous()
        torch.cuda.synchronize()
        eng.upload_X_device(chunk.data_ptr(), chunk.stride(0), chunk.shape[0], _native.X_CELLS_BY_GENES, off)
        eng.synchronize()
    eng.finalize_X()
    for i in range(2):
        eng.upload_Y(i, bench.labels_onehot(N, seed=1 + i))
    eng.set_factors(W0, H0, B0)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 3.0:              # >= 2 s of back-to-back launches before the reading
        eng.run(50, with_loss=True)
        eng.synchronize()
        n += 50
    info = eng.info()
    lib = _native.load()
    nwg = info.grid_b
    buf = (C.c_ulonglong * (4 * 1024))()
    lib.alpine_debug_read_clk.argtypes = [C.c_void_p, C.c_int]
    assert lib.alpine_debug_read_clk(buf, 4 * 1024) == 0
    raw = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 4)[:nwg]
    a = raw[:, :2].astype(np.float64)
    ghz = a[:, 0] / a[:, 1] * 0.1
    # spread of the workgroups of that launch on the shared 100 MHz counter: who starts late, who ends late, by XCD
    start = (raw[:, 2] - raw[:, 2].min()).astype(np.float64) / 100.0
    life = a[:, 1] / 100.0
    end = start + life
    xcc = raw[:, 3].astype(np.int64)
    pc = lambda v: " / ".join(f"{np.percentile(v, q):.1f}" for q in (0, 10, 50, 90, 100))
    print(f"  start offsets us (min/p10/p50/p90/max): {pc(start)}")
    print(f"  lifetimes us: {pc(life)}")
    print(f"  end offsets us: {pc(end)}   -> span first start .. last end {end.max():.1f} us; work-balanced ideal {life.mean() + start.min():.1f} us")
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        print(f"  XCD {x}: {int(m.sum())} workgroups, lifetime median {np.median(life[m]):.1f} us (min {life[m].min():.1f}, max {life[m].max():.1f}), start median {np.median(start[m]):.1f}")
    order = np.argsort(life)
    print("  slowest 8 workgroups (blockIdx, XCD, lifetime us):", [(int(i), int(xcc[i]), round(float(life[i]), 1)) for i in order[-8:]])
    print("  fastest 8 workgroups:", [(int(i), int(xcc[i]), round(float(life[i]), 1)) for i in order[:8]])
    print(f"x_scale {x_scale}: x3_wide={info.x3_wide} multi-plane fraction {info.x_multi_plane_fraction:.3f}; {n} iterations in "
          f"{time.perf_counter() - t0:.1f} s; last W^TX sweep: {nwg} workgroups, in-kernel clock median {np.median(ghz):.3f} GHz "
          f"(p10 {np.percentile(ghz, 10):.3f}, p90 {np.percentile(ghz, 90):.3f}); workgroup lifetime median {np.median(a[:, 1]) / 100.0:.1f} us")
    eng.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sweepclock":
        sweepclock(float(sys.argv[2]) if len(sys.argv) > 2 else 1.0, int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    elif len(sys.argv) > 1 and sys.argv[1] == "build":
        build(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    elif len(sys.argv) > 1 and sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 25000)
    else:
        sys.exit(__doc__)
