#!/usr/bin/env python3
"""bench.py -- NMF update iterations / second of the ALPINE MU fit loop on MI355X.

A "step" is one full-batch multiplicative-update iteration (W, every B_i, H; alpine/main.py:589-663)
INCLUDING its loss row (main.py:666) over one synthetic gene x cell matrix that is already resident
in HBM when the timed region starts.  Workload at every GPU count = BASELINE.json's metric shape,
"cfg3" of SURVEY.md 8d: 20 000 genes x 200 000 cells, K = 50 + [5, 5], 2 two-level covariates,
lam = [1e3, 1e3], alpha_W = 1.0, orth_W = 0.1, l1_ratio_W = 0.5, KL loss, float32 X / W / H / B in HBM.
With N GPUs the cell axis is sharded (one process per GPU, one RCCL all-reduce per iteration): strong scaling.

Sweep modes (--dtype; all but "bf16" give float32-grade results):
  x3 (default)  X float32; products formed from the exact bf16 planes of both factors (6 bf16 MFMAs per fragment
                pair, float32 accumulate): no precondition on X, HBM-bound (4 B per element of X per sweep)
  f32           the float32 MFMA on the same storage: MFMA-bound (reported under other_modes of the default run)
  split         X pre-split into 1-2 exact bf16 planes at ingest (integer counts): HBM-bound at 2 B per element
  bf16          operands ROUNDED to bf16 (BASELINE config 5), tolerance reported by the tests

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: starts its own N worker processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU; the per-iteration all-reduce of the reduce block is RCCL over xGMI, enqueued by the library's
own C loop (alpine_comm_init_rank / alpine_run; --comm torch keeps torch.distributed.all_reduce on the nccl backend
instead).  `python bench.py --gpus N` without a launcher spawns the N workers itself (fresh child processes, started
before this process touches a GPU), relays rank 0's JSON line and fails if any worker fails or the watchdog expires.

Rank 0 prints ONE JSON line (see the repo's driver contract) with two extra objects:
  roofline     -- the dominant kernel (the streaming sweep, 2 launches/iteration): algorithmic bytes (or, for
                  --dtype f32, flops) per launch / average launch time measured live with hipEvents on the
                  kernel's stream, against the 8 TB/s HBM peak (f32: the 157.3 TF float32 MFMA peak)
  cpu_baseline -- the oracle's faithful torch-CPU restatement of the reference loop timed on this host's cores: the
                  whole workload directly when the host has > 250 GiB free (1 warm-up + 2 timed iterations, under a
                  wall budget), else a 50 000-cell sample extrapolated linearly in cells; the branch taken is recorded
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
HBM_PEAK_GBPS = 8000.0
HBM_READ_CEILING_GBPS = 7100.0      # compute-free streaming read of the sweeps' tiles measured on MI355X (tools/xcd_balance.hip: 6.96-7.1 TB/s)
FULLSIG_SCALE = 0.3712345          # multiplies the synthetic counts for the full-significand legs (every float32 then needs all three bf16 planes)

WORKLOADS = {
    # name: genes, cells, K_u, k_i, alpha, orth, l1
    "cfg2": dict(genes=20000, cells=50000, ku=50, kcov=[5], alpha_W=0.0, orth_W=0.0, l1_ratio_W=0.0),
    "cfg3": dict(genes=20000, cells=200000, ku=50, kcov=[5, 5], alpha_W=1.0, orth_W=0.1, l1_ratio_W=0.5),
    "cfg4": dict(genes=20000, cells=1000000, ku=100, kcov=[5], alpha_W=0.0, orth_W=0.0, l1_ratio_W=0.0),
    "tiny": dict(genes=2000, cells=5000, ku=20, kcov=[2], alpha_W=0.0, orth_W=0.0, l1_ratio_W=0.0),
    # cfg3's matrix with K = 22 + [5, 5] = 32 (one MFMA tile of components): the shape at which the x3 kernels fit two waves per SIMD
    "cfg3_k32": dict(genes=20000, cells=200000, ku=22, kcov=[5, 5], alpha_W=1.0, orth_W=0.1, l1_ratio_W=0.5),
    # cfg3's matrix with K = 140 + [5, 5] = 150 and K = 246 + [5, 5] = 256: the blocked two-half path for 128 < K <= 256 (kernels_wide.hpp)
    "cfg3_k150": dict(genes=20000, cells=200000, ku=140, kcov=[5, 5], alpha_W=1.0, orth_W=0.1, l1_ratio_W=0.5),
    "cfg3_k216": dict(genes=20000, cells=200000, ku=206, kcov=[5, 5], alpha_W=1.0, orth_W=0.1, l1_ratio_W=0.5),      # 14 of 16 component tiles: the widest one-pass sweep
    "cfg3_k256": dict(genes=20000, cells=200000, ku=246, kcov=[5, 5], alpha_W=1.0, orth_W=0.1, l1_ratio_W=0.5),
    "cfg3_k512": dict(genes=20000, cells=200000, ku=502, kcov=[5, 5], alpha_W=1.0, orth_W=0.1, l1_ratio_W=0.5),      # four column blocks of 128 (round 4)
    "cfg3_k1024": dict(genes=20000, cells=200000, ku=1014, kcov=[5, 5], alpha_W=1.0, orth_W=0.1, l1_ratio_W=0.5),   # the largest model this build takes
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # (0.5 s of device time at the headline size: past the first ~50 ms, in which the clock still settles)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--cells", type=int, default=None, help="override the number of cells")
    ap.add_argument("--genes", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-modes", action="store_true", help="skip the extra (non-headline) split / bf16 measurements at N=1")
    ap.add_argument("--no-loss", action="store_true", help="updates only (secondary number)")
    ap.add_argument("--cpu-sample-cells", type=int, default=0,
                    help="cells of the CPU-baseline sample; 0 = the whole workload timed directly when MemAvailable > 250 GiB "
                         "(~100 GB RSS at 20k x 200k), else 50000 (BASELINE config 2's size, ~25 GB RSS) or 12000")
    ap.add_argument("--cpu-warmup-budget", type=float, default=45.0,
                    help="seconds the warm-up iteration of the direct CPU leg may take before it falls back to the 50000-cell sample")
    ap.add_argument("--comm", default="auto", choices=["auto", "native", "torch"],
                    help="N > 1: carrier of the all-reduce -- native = RCCL inside libalpine_hip (default), torch = torch.distributed")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="N > 1 self-launch: watchdog over the workers, seconds")
    ap.add_argument("--dtype", default="x3", choices=["f32", "bf16", "split", "x3"],
                    help="x3: float32 X, products from exact bf16 planes in the sweep (float32-grade, any X); f32: float32 MFMA; "
                         "split: X pre-split into exact bf16 planes (float32-grade; needs bf16-exact X such as counts); "
                         "bf16: operands rounded to bf16")
    ap.add_argument("--x-scale", type=float, default=1.0, help="multiply the synthetic counts (e.g. 300 -> values need two bf16 planes)")
    ap.add_argument("--split-a", type=int, default=0)
    ap.add_argument("--split-b", type=int, default=0)
    return ap.parse_args()


def labels_onehot(n_cells: int, seed: int) -> np.ndarray:
    """2 x N float32 one-hot of an i.i.d. p=0.5 two-level covariate."""
    rng = np.random.default_rng(seed)
    lab = rng.integers(0, 2, size=n_cells)
    Y = np.zeros((2, n_cells), dtype=np.float32)
    Y[lab, np.arange(n_cells)] = 1.0
    return Y


def cpu_baseline(wl: dict, sample_cells: int, full_cells: int, dev=None, warmup_budget_s: float = 45.0) -> dict:
    """Oracle (faithful restatement of main.py:500-667 incl. randperm gather and per-iteration loss) on torch-CPU, all host
    cores, on the same synthetic matrix the GPU legs ran on (generated on the device, copied to the host).

    SURVEY.md 8d: time the metric's workload DIRECTLY when the host has the RAM (the reference needs ~6.2 x the bytes of X as
    RSS, ~100 GB at 20k x 200k): `sample_cells` <= 0 and MemAvailable > 250 GiB -> all `full_cells` cells, 1 warm-up + 2 timed
    iterations, guarded by a wall budget (warm-up iteration > `warmup_budget_s` -> fall back to the 50 000-cell sample,
    extrapolated linearly in cells, as does a host with less memory)."""
    import resource
    import torch
    from oracle import alpine_oracle as orc
    G, ku, kcov = wl["genes"], wl["ku"], wl["kcov"]
    t_leg = time.perf_counter()
    mem_avail_gb = None
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable"):
                    mem_avail_gb = int(line.split()[1]) / 2**20
    except OSError:
        pass
    mem_txt = f"MemAvailable {mem_avail_gb:.0f} GiB" if mem_avail_gb is not None else "MemAvailable unknown"
    if sample_cells > 0:
        plan = [min(sample_cells, full_cells)]
        branch = f"--cpu-sample-cells {sample_cells}"
    elif mem_avail_gb is not None and mem_avail_gb > 250.0:
        plan = [full_cells] + ([50000] if full_cells > 50000 else [])        # direct; the sample is the budget fall-back
        branch = f"auto: {mem_txt} > 250 GiB -> all {full_cells} cells timed directly"
    else:
        # ~6.5 x the bytes of X as RSS (24 GB at 20k x 50k): BASELINE config 2's size when the host clearly has that, else small
        need_gb = 6.5 * 4.0 * G * 50000 / 2**30 + 8.0
        big = mem_avail_gb is not None and mem_avail_gb > 2.0 * need_gb
        plan = [min(full_cells, 50000 if big else 12000)]
        branch = f"auto: {mem_txt} {'>' if big else '<='} 2 x {need_gb:.0f} GiB (and <= 250 GiB) -> {plan[0]} cells"

    # genes x cells, the oracle's layout, straight from the device generator (a numpy Poisson draw of 4e9 elements would take
    # longer than the timed iterations); without a device (tests of this function alone) the host generator
    n_max = plan[0]
    if dev is not None:
        from alpine_amd.datasets import synth_counts_device_chunks
        X_gn = np.empty((G, n_max), dtype=np.float32)
        for off, chunk in synth_counts_device_chunks(n_max, G, rank=ku, seed=0, device=dev, chunk_cells=8192):
            X_gn[:, off:off + chunk.shape[0]] = chunk.t().contiguous().cpu().numpy()
            del chunk
        torch.cuda.empty_cache()
    else:
        from alpine_amd.datasets import synth_counts_host
        X_gn = np.ascontiguousarray(synth_counts_host(n_max, G, rank=ku, seed=0).T)
    t_gen = time.perf_counter() - t_leg
    p = orc.OracleParams(n_components=ku, n_covariate_components=list(kcov), lam=[1e3] * len(kcov),
                         orth_W=wl["orth_W"], alpha_W=wl["alpha_W"], l1_ratio_W=wl["l1_ratio_W"])

    fallback = None
    for attempt, n_s in enumerate(plan):
        Xs = X_gn if n_s == X_gn.shape[1] else np.ascontiguousarray(X_gn[:, :n_s])
        Ys = [labels_onehot(n_s, seed=1 + i).T for i in range(len(kcov))]       # N x C
        s = orc.init_factors(p, Xs, Ys)
        t0 = time.perf_counter()
        orc.fit_faithful(p, s, 1)                       # warm-up
        t_warm = time.perf_counter() - t0
        if attempt + 1 < len(plan) and t_warm > warmup_budget_s:
            fallback = (f"the warm-up iteration at {n_s} cells took {t_warm:.1f} s > the {warmup_budget_s:.0f} s budget: "
                        f"fell back to {plan[attempt + 1]} cells, scaled linearly")
            del s
            continue
        n_it = 2 if n_s == full_cells and n_s > 50000 else 3
        t0 = time.perf_counter()
        orc.fit_faithful(p, s, n_it)
        dt = time.perf_counter() - t0
        break
    del s
    it_s_sample = n_it / dt
    # the same mathematics in the re-associated minimal-op form (no G x N temporaries, trace-form loss) on the same cores:
    # separates the algorithmic part of the speed-up from the hardware part
    s2 = orc.init_factors(p, Xs, Ys)
    orc.fit_fused(p, s2, 1, with_loss=True)
    t0 = time.perf_counter()
    orc.fit_fused(p, s2, n_it, with_loss=True)
    it_s_fused = n_it / (time.perf_counter() - t0)
    del s2
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    direct = n_s == full_cells
    how = ("timed directly at the metric's size, no extrapolation" if direct else
           f"measured {it_s_sample:.4f} it/s on the sample, scaled linearly in cells to {full_cells} (conservative: the reference "
           f"scales super-linearly, BASELINE.md section 2)")
    return {
        "value": it_s_sample * n_s / full_cells,
        "unit": "iterations/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": (f"oracle.fit_faithful (torch-CPU fp32 restatement of alpine/main.py:500-667 incl. randperm gather and "
                   f"per-iteration loss), {G} genes x {n_s} cells of the same synthetic matrix, {n_it} timed "
                   f"iterations after 1 warm-up ({t_warm:.1f} s) = {dt:.1f} s; {how}"),
        "measured_sample_it_per_s": it_s_sample, "sample_cells": n_s, "extrapolated": not direct, "sample_branch": branch,
        "budget_fallback": fallback, "warmup_iteration_s": t_warm, "timed_iterations": n_it, "timed_s": dt, "matrix_to_host_s": t_gen,
        "peak_rss_GiB": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20, 1), "mem_available_GiB_before": mem_avail_gb,
        "leg_seconds": time.perf_counter() - t_leg,
        "fused_port_value": it_s_fused * n_s / full_cells,      # oracle.fit_fused, same sample and scaling
        "fused_port_measured_sample_it_per_s": it_s_fused,
        "host_cpu": cpu_model, "os_cpu_count": os.cpu_count(), "torch": torch.__version__,
    }


def pmc_traffic(workload: str, world: int, kp: int, dtype: str = "f32", fullsig: bool = False, kernel_name: str = ""):
    """HBM bytes per launch of the sweep kernel from the committed rocprofv3 PMC passes (collected in separate
    runs, FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md); None when no matching profile exists.
    `fullsig`: the leg ran on full-significand data -> the `_fullsig` profile of that mode (another kernel, other traffic);
    the profile must be of the kernel that ran (`kernel_name`) and of the same k tiling."""
    import glob
    import re
    if world != 1:
        return None
    best = None
    tag = {"f32": "", "bf16": "_bf16", "split": "_split", "x3": "_x3"}[dtype] + ("_fullsig" if fullsig else "")
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*", f"{workload}{tag}_stream_gemm_pmc_summary.json"))):
        try:
            d = json.load(open(f))
            m = re.search(r"<(\d+)", d.get("kernel", ""))
            same_kernel = not kernel_name or d.get("kernel", "").split("<")[0].strip().split("::")[-1] == kernel_name
            if m and int(m.group(1)) == kp // 32 and same_kernel:
                best = {"bytes_per_launch": d["traffic_bytes_per_launch"], "over_algorithmic": d["traffic_over_algorithmic"],
                        "kernel": d.get("kernel"), "source": os.path.relpath(f, REPO)}
        except (OSError, ValueError, KeyError):
            pass
    return best


def visible_gpu_count() -> int:
    """GPUs this process tree may use, counted WITHOUT starting a GPU runtime in the caller (the launcher parent stays
    GPU-free for its whole life; its workers are fresh processes): a short-lived child asks torch, which honours the
    *_VISIBLE_DEVICES variables and the container's device permissions (sysfs would list the host's GPUs)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True,
                           timeout=600)
        return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else 0
    except (OSError, ValueError, subprocess.TimeoutExpired):
        return 0


def launch_workers(args) -> int:
    """`python bench.py --gpus N` outside a launcher: start N fresh worker processes (one per GPU) BEFORE this process
    makes any GPU call, relay rank 0's JSON line, fail loudly if a worker fails or the watchdog expires.  No exec of
    this process, no kill-by-pattern: only the PIDs started here are ever signalled."""
    import collections
    import socket
    import subprocess
    import threading
    n = args.gpus
    rehearsal = os.environ.get("ALPINE_BENCH_REHEARSAL_ONE_GPU") == "1"
    have = visible_gpu_count()                  # asked in a short-lived child: this parent never loads a GPU runtime
    if have < n and not rehearsal:
        print(f"bench.py --gpus {n}: only {have} GPU(s) visible (ALPINE_BENCH_REHEARSAL_ONE_GPU=1 rehearses N ranks on one "
              f"GPU over gloo)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    base = dict(os.environ)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    procs, lines = [], []
    tails = [collections.deque(maxlen=60) for _ in range(n)]       # last stderr lines of every rank, replayed on failure

    def pump(stream, keep):
        for line in stream:
            if keep and line.lstrip().startswith("{"):
                lines.append(line)
            else:
                sys.stderr.write(line)
        stream.close()

    def pump_err(stream, r):
        for line in stream:
            tails[r].append(line)
            sys.stderr.write(f"[rank {r}] {line}")
        stream.close()

    threads = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True)
        procs.append(p)
        for t in (threading.Thread(target=pump, args=(p.stdout, r == 0), daemon=True),
                  threading.Thread(target=pump_err, args=(p.stderr, r), daemon=True)):
            t.start()
            threads.append(t)
    deadline = time.monotonic() + args.launch_timeout
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            rc = bad[0][1] if bad[0][1] > 0 else 1
            print(f"bench.py: worker rank {bad[0][0]} exited with {bad[0][1]}; stopping the others", file=sys.stderr)
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            print(f"bench.py: watchdog: workers still running after {args.launch_timeout:.0f} s; stopping them", file=sys.stderr)
            rc = 124
            break
        time.sleep(0.2)
    if rc:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
    for t in threads:
        t.join(timeout=5)
    if rc:
        # one block per rank, so that the first failure is readable even when the ranks' output interleaved above
        for r, p in enumerate(procs):
            print(f"---- bench.py: rank {r} exit code {p.poll()}; last {len(tails[r])} stderr lines ----", file=sys.stderr)
            sys.stderr.write("".join(tails[r]))
    if rc == 0 and len(lines) != 1:
        print(f"bench.py: expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        rc = 1
    if rc == 0:
        sys.stdout.write(lines[0])
        sys.stdout.flush()
    return rc


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_workers(args))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # before the HIP runtime starts: the host driver only supports dmabuf IPC
    import torch
    import torch.distributed as dist
    from alpine_amd import _native
    from alpine_amd.datasets import synth_counts_device_chunks
    from alpine_amd.model import draw_initial_factors
    from alpine_amd.sharded import NativeComm, ShardedLoop, TorchDistComm, attach_native_comm, shard_bounds

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world                  # under a launcher the environment decides
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X; there is no CPU fallback for the product path")
    # rehearsal knob for a one-GPU box: all ranks share device 0 and gloo carries the (device) reduce block;
    # the driver's multi-GPU runs never set it and use RCCL with one device per rank
    rehearsal = os.environ.get("ALPINE_BENCH_REHEARSAL_ONE_GPU") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    wl = dict(WORKLOADS[args.workload])
    if args.cells:
        wl["cells"] = args.cells
    if args.genes:
        wl["genes"] = args.genes
    G, N, ku, kcov = wl["genes"], wl["cells"], wl["ku"], wl["kcov"]
    K = ku + sum(kcov)
    c0, c1 = shard_bounds(N, world, rank)
    n_loc = c1 - c0
    lam = [1e3] * len(kcov)
    levels = [2] * len(kcov)

    with_loss = not args.no_loss
    stream = None
    if world > 1:
        # engine kernels and the RCCL all-reduce share ONE explicit stream (the default stream's handle is 0, which the
        # C ABI reads as "create a private stream" -- that would leave the collective unordered with the kernels)
        stream = torch.cuda.Stream(dev)
        torch.cuda.set_stream(stream)
        assert stream.cuda_stream != 0
    W0, H0, B0 = draw_initial_factors(42, 1e-6, G, N, kcov + [ku], levels)
    Ys = [np.ascontiguousarray(labels_onehot(N, seed=1 + i)[:, c0:c1]) for i in range(len(kcov))]

    comm_state = {"carrier": None, "native": False, "note": None, "rccl_ranks": None, "rccl_rank": None}

    def measure(dtype: str, x_scale: float = None) -> dict:
        """Build the resident state for one storage mode, run `warmup` untimed and `steps` timed iterations."""
        x_scale = args.x_scale if x_scale is None else x_scale
        kw = dict(n_genes=G, n_cells=n_loc, n_components=ku, cov_components=kcov, cov_levels=levels, lam=lam,
                  orth_W=wl["orth_W"], alpha_W=wl["alpha_W"], l1_ratio_W=wl["l1_ratio_W"], eps=1e-6,
                  loss_type="kl-divergence", device_id=local_rank, split_a=args.split_a, split_b=args.split_b, x_dtype=dtype)
        block = None
        if world > 1:
            nfl = _native.reduce_block_floats(G, n_loc, ku, kcov, levels)
            block = torch.zeros(nfl, dtype=torch.float32, device=dev)
            stream.synchronize()
            kw.update(stream=stream.cuda_stream, reduce_block=block.data_ptr())
        eng = _native.NativeShard(**kw)
        try:
            # synthetic input, generated on the device in cell chunks (never on the host)
            t_gen = time.perf_counter()
            for off, chunk in synth_counts_device_chunks(n_loc, G, rank=ku, seed=0, device=dev, chunk_cells=8192, cell_offset=c0):
                if x_scale != 1.0:
                    chunk = (chunk * x_scale).contiguous()
                torch.cuda.synchronize()
                eng.upload_X_device(chunk.data_ptr(), chunk.stride(0), chunk.shape[0], _native.X_CELLS_BY_GENES, off)
                eng.synchronize()
                del chunk
            eng.finalize_X()
            torch.cuda.empty_cache()
            for i in range(len(kcov)):
                eng.upload_Y(i, Ys[i])
            eng.set_factors(W0, H0, B0, h_col0=c0)
            t_gen = time.perf_counter() - t_gen
            info = eng.info()
            comm = None
            if world > 1:
                # carrier of the per-iteration all-reduce: the library's own RCCL communicator (default) or torch.distributed
                want = args.comm
                if want == "auto":
                    want = "torch" if rehearsal else "native"       # RCCL refuses several ranks on one GPU
                if want == "native":
                    try:
                        attach_native_comm(eng, dist)
                        comm = NativeComm(eng)
                        comm_state["carrier"] = "native: ncclAllReduce enqueued by libalpine_hip (alpine_run) on the ctx stream"
                        comm_state["native"] = True
                        # what the communicator ITSELF reports (ncclCommCount / ncclCommUserRank), not what it was told
                        comm_state["rccl_ranks"], comm_state["rccl_rank"] = eng.comm_count()
                    except Exception as e:          # noqa: BLE001 -- every rank raised together (all_ranks_ok): fall back together
                        if args.comm == "native":
                            raise
                        comm_state["note"] = f"native communicator failed ({type(e).__name__}: {e}); fell back to torch.distributed"
                if comm is None:
                    class TimedComm(TorchDistComm):
                        """all-reduce bracketed by events on the engine's stream: the time an iteration spends between the
                        end of its phase-1 kernels and the arrival of the reduced block (transfer + waiting for the slowest rank)."""
                        def __init__(self, blk):
                            super().__init__(blk)
                            self.pairs, self.on = [], False

                        def all_reduce(self):
                            if not self.on:
                                return super().all_reduce()
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record()
                            super().all_reduce()
                            e1.record()
                            self.pairs.append((e0, e1))
                    comm = TimedComm(block)
                    comm_state["carrier"] = f"torch.distributed.all_reduce ({dist.get_backend()}) on the ctx stream"
            loop = ShardedLoop(eng, comm) if world > 1 else None

            def probe_allreduce():
                """Before the timed loop (N > 1): one tiny all-reduce (pays the carrier's lazy connection set-up) and the
                reduce-block-sized one on its own, host-timed with a device sync on both sides -- so that a slow first
                iteration or a slow collective is visible in the JSON line instead of being folded into ms_per_step.
                The block holds zeros here (phase 1 rewrites all of it every iteration)."""
                def timed(n_floats, reps=1):
                    eng.synchronize()
                    torch.cuda.synchronize()
                    dist.barrier()
                    t = time.perf_counter()
                    for _ in range(reps):
                        comm.all_reduce_slice(0, n_floats)
                    eng.synchronize()
                    torch.cuda.synchronize()
                    return 1e3 * (time.perf_counter() - t) / reps
                nfl_ = int(info.reduce_block_floats)
                first = timed(min(1024, nfl_))
                first_full = timed(nfl_)
                standalone = timed(nfl_, reps=10)
                return {"first_call_ms": first, "first_full_block_ms": first_full, "standalone_ms": standalone}

            ar_probe = probe_allreduce() if world > 1 else None

            def run(n):
                if loop is not None:
                    loop.run(n, with_loss=with_loss)
                else:
                    eng.run(n, with_loss=with_loss)

            def fence():
                eng.synchronize()
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                    torch.cuda.synchronize()

            fence()
            t_w = time.perf_counter()
            run(args.warmup)
            fence()
            # hipEvents around the sweeps (and the all-reduce): every launch while an iteration is long, every 4th iteration
            # when it is short -- an event record between two kernels costs the stream ~2 us, 1.5 % of a 0.75 ms iteration
            est_ms = 1e3 * (time.perf_counter() - t_w) / max(1, args.warmup)
            stride = 1 if (est_ms >= 2.0 or args.warmup == 0) else 4
            if world > 1:
                sv = torch.tensor([stride], dtype=torch.int32, device=dev)
                dist.all_reduce(sv, op=dist.ReduceOp.MAX)
                stride = int(sv.item())
            eng.reset_losses()
            eng.set_profiling(stride)
            if comm is not None and hasattr(comm, "on"):
                comm.on = True
            t0 = time.perf_counter()
            run(args.steps)
            eng.synchronize()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            dt_local = dt
            fence()
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            ar_ms = None
            if comm is not None and hasattr(comm, "on"):
                comm.on = False
                ar_ms = float(np.mean([a.elapsed_time(b) for a, b in comm.pairs])) if comm.pairs else None
            elif comm is not None:
                ms_c, n_c = eng.kernel_time(_native.KERNEL_ALLREDUCE)       # hipEvents around ncclAllReduce inside the library
                ar_ms = ms_c / n_c if n_c else None
            ms_a, n_a = eng.kernel_time(_native.KERNEL_SWEEP_XHT)
            ms_b, n_b = eng.kernel_time(_native.KERNEL_SWEEP_WTX)
            per_rank = None
            if world > 1:
                # every rank's own numbers, so that the one JSON line shows a straggler, a rank on the wrong carrier or a communicator
                # that saw fewer ranks than the launch (the line's ms_per_step is the MAX over ranks)
                mine = {"rank": rank, "device": local_rank, "cells": int(n_loc), "ms_per_step": 1e3 * dt_local / args.steps,
                        "avg_ms_xht": ms_a / max(1, n_a), "avg_ms_wtx": ms_b / max(1, n_b), "allreduce_avg_ms": ar_ms,
                        "carrier": "native" if comm_state["native"] else "torch", "rccl_ranks": comm_state["rccl_ranks"], "rccl_rank": comm_state["rccl_rank"]}
                per_rank = [None] * world
                dist.all_gather_object(per_rank, mine)
            losses = eng.losses()
            eng.set_profiling(False)
            dt_noloss = None
            if world == 1 and with_loss:
                # secondary number: updates only.  The trace-form loss is a by-product of the next
                # iteration's phase 1, so this differs from the headline by one loss_finalize launch per iteration.
                eng.synchronize()
                t1 = time.perf_counter()
                eng.run(args.steps, with_loss=False)
                eng.synchronize()
                dt_noloss = time.perf_counter() - t1
        finally:
            eng.close()
            del block
            torch.cuda.empty_cache()
        return dict(dtype=dtype, dt=dt, ms_a=ms_a, n_a=n_a, ms_b=ms_b, n_b=n_b, losses=losses, info=info, t_gen=t_gen, dt_noloss=dt_noloss,
                    ar_ms=ar_ms, x_scale=x_scale, event_stride=stride, ar_probe=ar_probe, per_rank=per_rank)

    DTYPE_LABEL = {"f32": "f32", "bf16": "bf16 operands, f32 accumulate",
                   "split": "f32 via exact bf16-plane split (bf16 MFMA, f32 accumulate)",
                   "x3": "f32 X in HBM, products from exact bf16 planes (6 bf16 MFMAs per fragment pair), f32 accumulate"}

    def roofline(m: dict) -> dict:
        dtype, info = m["dtype"], m["info"]
        mf = dtype == "f32"
        launches = m["n_a"] + m["n_b"]
        avg_ms = (m["ms_a"] + m["ms_b"]) / max(1, launches)
        one_pass = K > 128 and dtype == "x3" and K <= 224           # the library's rule (alpine_create): stream_gemm_x3w2_kernel up to 224 components
        passes = -(-K // 128) if (K > 128 and not one_pass) else 1  # else each sweep of a wide model is one launch per column block of 128 components
        flops_per_launch = 2.0 * G * n_loc * K / passes           # algorithmic, unpadded K (4GNK per iteration / 2 sweeps)
        bytes_per_launch = (4.0 if dtype in ("f32", "x3") else 2.0) * G * n_loc   # X read once per sweep (counts: ONE bf16 plane in split mode)
        ach_tf = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        gbps = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # the kernel that actually ran: the x3 sweeps take the 16x16x32 form (x3w) on full-significand data and on wide
        # models with a padding tile -- alpine_finalize_X decides from a census of X and reports it in alpine_info.x3_wide
        if mf:
            kname, kdesc = "stream_gemm_kernel", "MFMA f32 32x32x2"
        elif one_pass:
            kname, kdesc = "stream_gemm_x3w2_kernel", "one pass over X for 128 < K <= 256: float32 X split into exact bf16 planes in registers, MFMA bf16 16x16x32"
        elif dtype == "x3" and info.sweep_waves_per_simd == 2:
            kname, kdesc = "stream_gemm_x3v_kernel", "64 < K <= 128, two waves per SIMD: float32 X split into exact bf16 planes in registers, MFMA bf16 16x16x32"
        elif dtype == "x3" and info.x3_wide:
            kname, kdesc = "stream_gemm_x3w_kernel", "float32 X split into exact bf16 planes in registers, MFMA bf16 16x16x32"
        elif dtype == "x3":
            kname, kdesc = "stream_gemm_x3_kernel", "float32 X split into exact bf16 planes in registers, MFMA bf16 32x32x16"
        else:
            kname, kdesc = "stream_gemm_bf16_kernel", "MFMA bf16 32x32x16, k-packed X" + (" with exact plane split" if dtype == "split" else "")
        fullsig = m.get("x_scale", 1.0) != 1.0
        tr = pmc_traffic(args.workload, world, info.k_padded, dtype, fullsig, kname) if not (args.cells or args.genes) else None
        return {
            "kernel": f"{kname} ({kdesc}; XH^T and W^TX sweeps)", "kernel_name": kname,
            "bound": "mfma" if mf else "hbm",
            "achieved": ach_tf if mf else gbps,
            "peak": FP32_MFMA_PEAK_TFLOPS if mf else HBM_PEAK_GBPS,
            "unit": "TFLOP/s" if mf else "GB/s",
            "frac": (ach_tf / FP32_MFMA_PEAK_TFLOPS) if mf else gbps / HBM_PEAK_GBPS,
            # second view of the same measurement: against the streaming-read ceiling MEASURED on this chip with a compute-free
            # kernel reading the same tiles (tools/xcd_balance.hip / read_bw_tiled.hip), not the 8 TB/s spec
            "frac_of_read_ceiling": None if mf else gbps / HBM_READ_CEILING_GBPS, "read_ceiling_GBps": None if mf else HBM_READ_CEILING_GBPS,
            "traffic": (tr or {}).get("bytes_per_launch"), "traffic_detail": tr,
            "avg_launch_ms": avg_ms, "launches": launches, "event_stride": m.get("event_stride", 1),     # launches = event-timed launches (every event_stride-th iteration)
            "avg_ms_xht": m["ms_a"] / max(1, m["n_a"]), "avg_ms_wtx": m["ms_b"] / max(1, m["n_b"]),
            "algorithmic_flops_per_launch": flops_per_launch, "algorithmic_bytes_per_launch": bytes_per_launch,
            "tflops": ach_tf, "hbm_achieved_GBps": gbps, "hbm_frac_of_8TBps": gbps / HBM_PEAK_GBPS,
            "sweep_launches_per_step": 2 * passes,
            "sweeps_share_of_step": (passes * (m["ms_a"] / max(1, m["n_a"]) + m["ms_b"] / max(1, m["n_b"])) / (1e3 * m["dt"] / args.steps)) if m["dt"] > 0 else 0.0,
            "x3_wide": int(info.x3_wide), "sweep_waves_per_simd": int(info.sweep_waves_per_simd), "x_multi_plane_fraction": float(info.x_multi_plane_fraction),
            "accumulation_span_rows": [int(info.span_rows_a), int(info.span_rows_b)],
        }

    main_m = measure(args.dtype)
    cells_by_rank, rccl_version = [n_loc], None
    if world > 1:
        cells_by_rank = [None] * world
        dist.all_gather_object(cells_by_rank, int(n_loc))
        try:
            rccl_version = {"libalpine_hip (librccl it is linked against)": _native.comm_version(),
                            "torch": ".".join(str(v) for v in torch.cuda.nccl.version())}
        except Exception as e:          # noqa: BLE001 -- informational only
            rccl_version = f"unavailable ({type(e).__name__}: {e})"
    others = {}
    if world == 1 and not args.no_other_modes and not args.no_loss:
        # the same workload in the other storage modes (not the headline): exact bf16-plane split (float32-grade
        # results, applies because the synthetic counts are bf16-exact) and rounded bf16 operands (BASELINE config 5)
        # ... and the headline's and the float32 MFMA's sweeps on FULL-SIGNIFICAND data (the same matrix times 0.3712345:
        # 24-bit significands, all three bf16 planes populated): throughput depends on the values on this chip
        # (x3 skips zero planes of count data; the clock the chip holds under load depends on operand toggling)
        legs = [(n, n, None) for n in ("x3", "split", "bf16", "f32") if n != args.dtype]
        if args.x_scale == 1.0:
            legs += [(f"{n}_fullsig", n, FULLSIG_SCALE) for n in ("x3", "f32")]
        for key, dt_name, xs in legs:
            try:
                om = measure(dt_name, xs)
                others[key] = {
                    "dtype": DTYPE_LABEL[dt_name], "x_scale": om["x_scale"], "value": args.steps / om["dt"], "unit": "iterations/s",
                    "ms_per_step": 1e3 * om["dt"] / args.steps, "roofline": roofline(om),
                    "final_loss_row": om["losses"][-1].tolist() if len(om["losses"]) else None,
                    "final_total_loss_rel_diff_vs_headline": (abs(om["losses"][-1][0] - main_m["losses"][-1][0]) / abs(main_m["losses"][-1][0])
                                                              if xs is None and len(om["losses"]) and len(main_m["losses"]) else None),
                    "device_GiB": round(om["info"].device_bytes / 2**30, 2),
                }
            except Exception as e:            # an optional leg must not take the headline down
                others[key] = {"error": f"{type(e).__name__}: {e}"}

    if world > 1 and rank == 0:
        # self-verification of the first real multi-rank runs: a native carrier must have been seen by RCCL as `world` ranks with the
        # launch's rank numbering on EVERY rank, and no rank may be on another carrier than rank 0 -- else the run fails instead of
        # printing a line that looks like a scaling result
        pr = main_m["per_rank"] or []
        bad = [r for r in pr if r["carrier"] != pr[0]["carrier"]
               or (r["carrier"] == "native" and (r["rccl_ranks"] != world or r["rccl_rank"] != r["rank"]))]
        if len(pr) != world or bad:
            print(f"bench.py: the communicator does not match the launch of {world} ranks: {bad or pr}", file=sys.stderr)
            dist.destroy_process_group()
            sys.exit(3)
    if rank == 0:
        dt, info, losses = main_m["dt"], main_m["info"], main_m["losses"]
        out = {
            "metric": "NMF update iterations/sec (20k genes x 200k cells, K=50)",
            "value": args.steps / dt, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": DTYPE_LABEL[args.dtype], "data": "synthetic",
            "config": {
                "workload": (f"{args.workload}: {G} genes x {N} cells, K={ku}+{kcov} (K={K}), {len(kcov)} two-level covariates, "
                             f"lam=1e3, alpha_W={wl['alpha_W']}, orth_W={wl['orth_W']}, l1_ratio_W={wl['l1_ratio_W']}, KL loss, "
                             f"full batch, loss row every iteration={with_loss}; X ~ Poisson(Gamma(0.3)xGamma(0.3)), mean~1"),
                "cells_per_gpu": n_loc, "cells_per_gpu_by_rank": cells_by_rank, "k_padded": info.k_padded, "split_xht": info.split_a, "split_wtx": info.split_b,
                "grid_xht": info.grid_a, "grid_wtx": info.grid_b, "device_GiB": round(info.device_bytes / 2**30, 2),
                "parallelism": f"cells/{world}",
                "x_scale": args.x_scale,      # 1.0 = the spec'd Poisson counts; throughput is value-dependent on this chip
            },
            "roofline": roofline(main_m),
            "final_loss_row": losses[-1].tolist() if len(losses) else None,
            "allreduce": ({"avg_ms_on_rank0": main_m["ar_ms"], "bytes": int(info.reduce_block_floats) * 4,
                           "carrier": comm_state["carrier"], "fallback_note": comm_state["note"],
                           "rccl_ranks": comm_state["rccl_ranks"],       # ncclCommCount of the library's communicator (None: torch carrier)
                           "per_rank": main_m["per_rank"],
                           "note": "avg_ms_on_rank0: hipEvents on the ctx stream around the all-reduce inside the timed loop (transfer + "
                                   "wait for the slowest rank); first_call_ms / standalone_ms: host-timed before the loop, see probe_allreduce",
                           **(main_m["ar_probe"] or {})}
                          if world > 1 else None),
            "rccl_version": rccl_version,
            "updates_only_iterations_per_s": (args.steps / main_m["dt_noloss"]) if main_m.get("dt_noloss") else None,
            "setup_s": main_m["t_gen"],
            "other_modes": others or None,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_sample_cells, N, dev=dev, warmup_budget_s=args.cpu_warmup_budget)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
