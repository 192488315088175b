"""Per-iteration timeline from a rocprofv3 --kernel-trace CSV: for the last full iteration of bench.py, every kernel
with its duration and the idle gap before it.  Usage: python tools/timeline.py <dir with *_kernel_trace.csv> [anchor] [out.json]
(anchor = a kernel that runs once per iteration, default loss_finalize; w_update_mfma for the fused MU iteration)"""
import csv
import glob
import json
import os
import sys


def main():
    d = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "loss_finalize"
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f, newline="")))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
    if len(idx) < 3:
        print("not enough iterations")
        return
    a, b = idx[-3], idx[-2]                 # one full period anchor -> anchor
    prev_end = int(rows[a - 1]["End_Timestamp"])
    tot_k = tot_g = 0
    rec = []
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("void ", "").replace("alpine::", "").split("(")[0][:48]
        print(f"{name:48s} gap {max(0, s - prev_end) / 1e3:7.1f} us   dur {(e - s) / 1e3:8.1f} us")
        rec.append({"kernel": name.strip(), "gap_us": max(0, s - prev_end) / 1e3, "dur_us": (e - s) / 1e3})
        tot_k += e - s
        tot_g += max(0, s - prev_end)
        prev_end = e
    print(f"period: kernels {tot_k / 1e3:.1f} us + gaps {tot_g / 1e3:.1f} us = {(tot_k + tot_g) / 1e3:.1f} us")
    if len(sys.argv) > 3:
        # medians over ALL periods of the trace as well (one period can be an outlier)
        per = {}
        for i0, i1 in zip(idx[1:-1], idx[2:]):
            for r in rows[i0:i1]:
                nm = r["Kernel_Name"].replace("void ", "").replace("alpine::", "").split("(")[0][:48].strip()
                per.setdefault(nm, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        n_per = max(1, len(idx) - 2)
        med = {k: sorted(v)[len(v) // 2] for k, v in per.items()}
        calls = {k: len(v) / n_per for k, v in per.items()}
        json.dump({"source": os.path.relpath(f), "anchor": anchor, "last_period": rec, "period_kernels_us": tot_k / 1e3, "period_gaps_us": tot_g / 1e3,
                   "median_dur_us_over_periods": med, "launches_per_period": calls,
                   "median_period_kernel_us": sum(med[k] * calls[k] for k in med), "periods": n_per}, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
