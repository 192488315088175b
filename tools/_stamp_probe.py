# throwaway: per-block phase times of h_update_mfma_kernel from a stamped build (tools/libalpine_stamp.so)
import ctypes as C, os, sys, numpy as np
os.environ["ALPINE_HIP_LIBRARY"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libalpine_stamp.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from alpine_amd import _native
from alpine_amd.datasets import synth_counts_device_chunks
from alpine_amd.model import draw_initial_factors
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
wl = dict(bench.WORKLOADS["cfg3"]); wl["cells"] = cells
G, N, ku, kcov = wl["genes"], wl["cells"], wl["ku"], wl["kcov"]
dev = torch.device("cuda", 0)
W0, H0, B0 = draw_initial_factors(42, 1e-6, G, N, kcov + [ku], [2, 2])
eng = _native.NativeShard(n_genes=G, n_cells=N, n_components=ku, cov_components=kcov, cov_levels=[2, 2], lam=[1e3, 1e3],
                          orth_W=wl["orth_W"], alpha_W=wl["alpha_W"], l1_ratio_W=wl["l1_ratio_W"], x_dtype="x3")
for off, chunk in synth_counts_device_chunks(N, G, rank=ku, seed=0, device=dev, chunk_cells=8192):
    torch.cuda.synchronize(); eng.upload_X_device(chunk.data_ptr(), chunk.stride(0), chunk.shape[0], _native.X_CELLS_BY_GENES, off); eng.synchronize()
eng.finalize_X()
for i in range(2): eng.upload_Y(i, bench.labels_onehot(N, seed=1 + i))
eng.set_factors(W0, H0, B0)
eng.run(10, with_loss=True); eng.synchronize()
lib = _native.load()
nb = (N + 127) // 128
buf = (C.c_ulonglong * (16 * nb))()
lib.alpine_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.alpine_debug_read_stamps(buf, 16 * nb) == 0
full = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 16).astype(np.int64)
a = full[:, :8]
if full[:, 8].any():
    t3, t4 = full[:, 3], full[:, 4]
    for nm, x0, x1 in (("  mfma end -> cov0 meta loaded", t3, full[:, 8]), ("  cov0 class0 first loop", full[:, 8], full[:, 9]), ("  shfl_xor", full[:, 9], full[:, 10]),
                       ("  rest of guided terms", full[:, 10], full[:, 11]), ("  final divide", full[:, 11], t4)):
        print(f"{nm:36s} median {np.median((x1 - x0) / 100.0):7.2f} us")
d = np.diff(a, axis=1) / 100.0          # s_memrealtime ticks are 10 ns -> us
names = ["ybuf/M2l/Bl fill + barrier", "H issue + pieces + H to C/D", "MFMA den", "guided + update", "tile store", "barrier + gram", "stats"]
print("blocks", nb, "kernel span us (first start -> last end)", (a[:, 7].max() - a[:, 0].min()) / 100.0)
for i, n in enumerate(names):
    print(f"{n:32s} median {np.median(d[:, i]):7.2f} us   p90 {np.percentile(d[:, i], 90):7.2f}")
print("block total median", np.median((a[:, 7] - a[:, 0]) / 100.0), "start skew p90-p10", (np.percentile(a[:, 0], 90) - np.percentile(a[:, 0], 10)) / 100.0)
eng.close()
