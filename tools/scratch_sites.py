"""Where do the kernels that use scratch touch it?  For every kernel of the library whose ISA contains scratch_* (spill)
instructions: the loop depth of each one, from the compiler's own basic-block annotations in the assembly
(`hipcc -S --cuda-device-only`).  Depth 0 = straight-line prologue / epilogue, 1 = the per-segment loop of a sweep,
2 = the stage loop (the hot loop).

    python tools/scratch_sites.py [--out profiles/rNN/scratch_sites.txt]"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "lib.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "--cuda-device-only", "-S",
               "-o", asm, os.path.join(REPO, "alpine_amd", "csrc", "alpine_hip.hip")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-4000:])
        text = open(asm).read().splitlines()
    out = [" ".join(cmd[1:8]) + " ...", ""]
    name, depth, sites, mfma_by_depth = None, 0, [], {}
    kernels = []
    for line in text:
        m = re.match(r"^(_ZN6alpine\w+):\s+; @", line)
        if m:
            name, depth, sites, mfma_by_depth = m.group(1), 0, [], {}
            continue
        if name is None:
            continue
        if re.match(r"^\.LBB\d+_\d+:", line):
            m = re.search(r"Depth=(\d+)", line)
            depth = int(m.group(1)) if m else 0
        elif "Loop Header: Depth=" in line or "Inner Loop Header: Depth=" in line:
            depth = int(re.search(r"Depth=(\d+)", line).group(1))
        ins = line.strip().split(" ")[0].split("\t")[0]
        if ins.startswith("scratch_"):
            sites.append((depth, line.strip()))
        if ins.startswith("v_mfma"):
            mfma_by_depth[depth] = mfma_by_depth.get(depth, 0) + 1
        if ins == "s_endpgm":
            if sites:
                kernels.append((name, sites, dict(mfma_by_depth)))
            name = None
    dem = subprocess.run(["c++filt"] + [k[0] for k in kernels], capture_output=True, text=True).stdout.splitlines()
    for (nm, sites, mf), d in zip(kernels, dem):
        d = re.sub(r"^void ", "", re.sub(r"\(.*", "", d)).replace("alpine::", "")
        hot = max(mf) if mf else 0
        out.append(f"{d}: MFMAs by loop depth {dict(sorted(mf.items()))}; hot loop = depth {hot}")
        for dep, s in sites:
            out.append(f"    depth {dep}{'  <-- IN THE HOT LOOP' if dep == hot and hot > 0 else ''}: {s}")
    text_out = "\n".join(out) + "\n"
    sys.stdout.write(text_out)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(text_out)


if __name__ == "__main__":
    main()
