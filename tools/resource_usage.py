"""Per-kernel register / scratch / LDS usage of libalpine_hip.so's device code, from hipcc's own remarks.

    python tools/resource_usage.py [--out profiles/rNN/kernel_resource_usage.txt] [--all]

Compiles alpine_amd/csrc/alpine_hip.hip with the library's flags plus -Rpass-analysis=kernel-resource-usage (into a
scratch file, not the shipped .so) and prints one line per kernel: VGPRs, AGPRs, spilled VGPRs, scratch bytes per lane,
occupancy, LDS.  Without --all only the sweeps, the update kernels and anything that uses scratch are listed."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occ",
        "VGPRs Spill": "vspill", "SGPRs Spill": "sspill", "LDS Size [bytes/block]": "lds", "TotalSGPRs": "sgpr"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--all", action="store_true")
    a = ap.parse_args()
    src = os.path.join(REPO, "alpine_amd", "csrc", "alpine_hip.hip")
    with tempfile.TemporaryDirectory() as td:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize",
               "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(td, "lib.so"), src, "-L/opt/rocm/lib", "-lrccl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-4000:])
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+):\s+(\d+)", line)
        if m and cur is not None and m.group(1).strip() in KEYS:
            cur[KEYS[m.group(1).strip()]] = int(m.group(2))
    dem = subprocess.run(["c++filt"] + [r_["name"] for r_ in rows], capture_output=True, text=True).stdout.splitlines()
    lines = [" ".join(cmd[1:9]) + " ...", f"{'kernel':74s} VGPR AGPR  spilled-VGPR scratch-B/lane waves/SIMD LDS-B"]
    for r_, d in zip(rows, dem):
        d = re.sub(r"^void ", "", re.sub(r"\(.*", "", d)).replace("alpine::", "")
        if a.all or "stream_gemm" in d or "update" in d or "wide" in d or "gram_cross" in d or r_.get("scratch", 0) > 0:
            lines.append(f"{d:74s} {r_.get('vgpr', -1):4d} {r_.get('agpr', -1):4d}  {r_.get('vspill', -1):12d} {r_.get('scratch', -1):14d} "
                         f"{r_.get('occ', -1):10d} {r_.get('lds', -1):5d}")
    text = "\n".join(lines) + "\n"
    sys.stdout.write(text)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(text)


if __name__ == "__main__":
    main()
