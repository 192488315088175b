"""Per-iteration timeline from a rocprofv3 --kernel-trace CSV: for the last full iteration of bench.py, every kernel
with its duration and the idle gap before it.  Usage: python tools/timeline.py <dir with *_kernel_trace.csv> [anchor]"""
import csv
import glob
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
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("void ", "").replace("alpine::", "").split("(")[0][:48]
        print(f"{name:48s} gap {max(0, s - prev_end) / 1e3:7.1f} us   dur {(e - s) / 1e3:8.1f} us")
        tot_k += e - s
        tot_g += max(0, s - prev_end)
        prev_end = e
    print(f"period: kernels {tot_k / 1e3:.1f} us + gaps {tot_g / 1e3:.1f} us = {(tot_k + tot_g) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
