"""Reduce the raw rocprofv3 output of tools/profile_mode.sh to the small files kept under profiles/:
per-kernel mean of every counter, the kernel-stats table, and the HBM-traffic / MFMA-busy summary of the sweep kernel
that bench.py reads for `roofline.traffic` (FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md)."""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict


def counter_means(raw_dir):
    acc = defaultdict(lambda: [0.0, 0])
    per_dispatch = defaultdict(float)
    for f in glob.glob(os.path.join(raw_dir, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                per_dispatch[(row["Kernel_Name"], row["Counter_Name"], row["Dispatch_Id"])] += float(row["Counter_Value"])
    for (k, c, _), v in per_dispatch.items():
        a = acc[(k, c)]
        a[0] += v
        a[1] += 1
    return {kc: (s / n, n) for kc, (s, n) in acc.items()}


def write_means(path, means):
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "Counter_Name", "mean", "count"])
        for (k, c), (m, n) in sorted(means.items()):
            w.writerow([re.sub(r"^void ", "", k), c, m, n])


def main():
    raw, out, wl, mode = sys.argv[1:5]
    bench = json.loads(open(os.path.join(out, f"{wl}_{mode}_kernel_stats_bench.json")).read().strip().splitlines()[-1])
    stats = glob.glob(os.path.join(raw, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(out, f"{wl}_{mode}_kernel_stats.csv"))
    groups = {g: counter_means(os.path.join(raw, g)) for g in ("fetch", "write", "sq", "tcc")}
    for g, means in groups.items():
        if g == "tcc" and not means:
            continue
        write_means(os.path.join(out, f"{wl}_{mode}_pmc_{g}.csv"), means)
    sweep = [k for (k, c) in groups["fetch"] if "stream_gemm" in k]
    if not sweep:
        print("no sweep kernel in the counter output", file=sys.stderr)
        return 1
    k = max(sweep, key=lambda kk: groups["fetch"][(kk, "FETCH_SIZE")][0])
    fetch_kb = groups["fetch"][(k, "FETCH_SIZE")][0]
    write_kb = groups["write"][(k, "WRITE_SIZE")][0]
    busy = groups["sq"].get((k, "SQ_VALU_MFMA_BUSY_CYCLES"), (0.0, 0))[0]
    gui = groups["sq"].get((k, "GRBM_GUI_ACTIVE"), (0.0, 0))[0]
    sq = {n: groups["sq"].get((k, n), (None, 0))[0] for n in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_BUSY_CYCLES")}
    rd = 2.0 * fetch_kb * 1024.0                    # gfx950: FETCH_SIZE reports half of a 16 B/lane coalesced stream
    wr = write_kb * 1024.0
    alg = bench["roofline"]["algorithmic_bytes_per_launch"]
    summary = {
        "command": f"tools/profile_mode.sh {mode} {wl}  (rocprofv3 --pmc <one counter group per run> --kernel-trace -- python3 bench.py "
                   f"--workload {wl} --dtype {mode} --steps 3 --warmup 1 --no-cpu-baseline --no-other-modes)",
        "workload": bench["config"]["workload"],
        "kernel": re.sub(r"^void ", "", k).split("(")[0],
        "FETCH_SIZE_KB_mean": fetch_kb, "WRITE_SIZE_KB_mean": write_kb,
        "hbm_read_bytes_per_launch_corrected": rd, "hbm_write_bytes_per_launch": wr,
        "traffic_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (rd + wr) / alg,
        "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE_sum_over_8_XCD": gui,
        "mfma_busy_fraction": busy / (gui * 128.0) if gui else None,     # per-SIMD busy cycles / (per-XCD active cycles x 32 CUs x 4 SIMDs)
        "sq_counters_per_launch": sq,      # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles
        "avg_launch_ms_kernel_trace": None,
        "note": "FETCH_SIZE doubled (gfx950); the counter sees fabric reads (HBM + MALL hits), i.e. X plus the operand-panel re-reads",
    }
    if stats:
        with open(stats[0], newline="") as fh:
            for row in csv.DictReader(fh):
                if "stream_gemm" in row["Name"]:
                    summary["avg_launch_ms_kernel_trace"] = float(row["AverageNs"]) / 1e6
                    summary["kernel_trace_calls"] = int(row["Calls"])
                    break
    hit = groups["tcc"].get((k, "TCC_HIT_sum"), (None, 0))[0]
    miss = groups["tcc"].get((k, "TCC_MISS_sum"), (None, 0))[0]
    if hit is not None and miss is not None and hit + miss > 0:
        summary["TCC_HIT_sum"], summary["TCC_MISS_sum"], summary["l2_hit_rate"] = hit, miss, hit / (hit + miss)
    tag = "" if mode == "f32" else f"_{mode}"
    if stats and summary["avg_launch_ms_kernel_trace"] and gui:
        # DVFS note of MI355X_MICROARCH.md: effective shader clock ~ GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel time
        summary["effective_clock_GHz_from_GRBM"] = gui / 8.0 / (summary["avg_launch_ms_kernel_trace"] * 1e-3) / 1e9
    with open(os.path.join(out, f"{wl}{tag}_stream_gemm_pmc_summary.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    print(json.dumps(summary))
    return 0


if __name__ == "__main__":
    sys.exit(main())
