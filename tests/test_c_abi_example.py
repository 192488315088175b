"""The C ABI driven from a plain-C host program (examples/fit_c.c): no Python, no torch types between the caller and
libalpine_hip.so.  CPU: it compiles against include/alpine_hip.h with gcc, links, and fails loudly without a GPU.
GPU: a golden case goes through it as a flat binary file and comes back equal to the reference's results."""
import os
import struct
import subprocess

import numpy as np
import pytest

from _golden import assert_loss_rows_close, load_case, rel_fro

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(REPO, "examples", "fit_c")


def build_example():
    from alpine_amd.build import build_library
    build_library()
    cmd = ["gcc", "-O2", "-Wall", "-Werror", "-D_DEFAULT_SOURCE", "-I" + os.path.join(REPO, "include"), os.path.join(REPO, "examples", "fit_c.c"), "-o", EXE,
           "-L" + os.path.join(REPO, "alpine_amd"), "-lalpine_hip", "-Wl,-rpath," + os.path.join(REPO, "alpine_amd"),
           "-Wl,-rpath-link,/opt/rocm/lib", "-lpthread"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return EXE


def write_problem(path, c, flags, scale=True):
    p = c.params
    n, g = c.X.shape
    ks, levs = p["n_covariate_components"], [y.shape[0] for y in c.Ys]
    with open(path, "wb") as f:
        f.write(struct.pack("<9i", 0x414C5031, g, n, p["n_components"], len(ks), 0 if p.get("loss_type", "kl-divergence") == "kl-divergence" else 1,
                            flags, c.T, int(scale)))
        for k, lev in zip(ks, levs):
            f.write(struct.pack("<2i", k, lev))
        f.write(struct.pack("<4d", p.get("orth_W", 0.0), p.get("alpha_W", 0.0), p.get("l1_ratio_W", 0.0), p.get("eps", 1e-6)))
        f.write(struct.pack(f"<{len(ks)}d", *p["lam"][:len(ks)]))
        for a in [c.X] + list(c.Ys) + [c.W0, c.H0] + list(c.B0):
            f.write(np.ascontiguousarray(a, dtype=np.float32).tobytes())


def read_result(path, c):
    n, g = c.X.shape
    K = c.W0.shape[1]
    raw = open(path, "rb").read()
    n_rows = struct.unpack_from("<i", raw, 0)[0]
    off = 4
    ncol = len(c.Ys) + 2
    losses = np.frombuffer(raw, dtype=np.float64, count=n_rows * ncol, offset=off).reshape(n_rows, ncol)
    off += 8 * n_rows * ncol
    W = np.frombuffer(raw, dtype=np.float32, count=g * K, offset=off).reshape(g, K)
    off += 4 * g * K
    H = np.frombuffer(raw, dtype=np.float32, count=K * n, offset=off).reshape(K, n)
    off += 4 * K * n
    Bs = []
    for b0 in c.B0:
        Bs.append(np.frombuffer(raw, dtype=np.float32, count=b0.size, offset=off).reshape(b0.shape))
        off += 4 * b0.size
    assert off == len(raw)
    return losses, W, H, Bs


def test_c_example_builds_links_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = build_example()
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    r = subprocess.run([exe, "--ranks", "0", "a", "b"], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    if torch.cuda.is_available():
        pytest.skip("GPU present: the no-GPU failure path is not reachable here")
    prob = tmp_path / "p.bin"
    write_problem(prob, load_case("kl_1cov"), flags=16)
    r = subprocess.run([exe, str(prob), str(tmp_path / "r.bin")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "alpine_create" in r.stderr          # no device: an error code and a message, no crash, no fallback
    assert not (tmp_path / "r.bin").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [16, 0], ids=["x3", "f32_mfma"])
@pytest.mark.parametrize("name", ["kl_2cov_nan", "fro_2cov_reg", "k74"])
def test_c_example_reproduces_reference(name, flags, tmp_path):
    exe = build_example() if not os.path.exists(EXE) else EXE
    c = load_case(name)
    prob, res = tmp_path / "p.bin", tmp_path / "r.bin"
    write_problem(prob, c, flags)
    r = subprocess.run([exe, str(prob), str(res)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    losses, W, H, Bs = read_result(res, c)
    assert rel_fro(W, c.WT) < 1e-4 and rel_fro(H, c.HT) < 1e-4
    for b, bt in zip(Bs, c.BT):
        assert rel_fro(b, bt) < 2e-4
    assert_loss_rows_close(losses, c.loss_history, n_cells=c.X.shape[0])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["kl_2cov_nan", "als_kl"])
def test_c_example_ranks_mode_native_rccl(name, tmp_path):
    """`fit_c --ranks 1`: the forked worker draws an RCCL id, joins a one-rank communicator and alpine_run enqueues the
    all-reduce itself; the spliced result must be BITWISE the single-process result (a one-rank sum changes nothing)."""
    exe = build_example()
    c = load_case(name)
    flags = 16 | (4 if c.params.get("use_als") else 0)
    prob, res1, res2 = tmp_path / "p.bin", tmp_path / "r1.bin", tmp_path / "r2.bin"
    write_problem(prob, c, flags)
    r = subprocess.run([exe, str(prob), str(res1)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "--ranks", "1", str(prob), str(res2)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert open(res1, "rb").read() == open(res2, "rb").read()
    assert not os.path.exists(str(res2) + ".id") and not os.path.exists(str(res2) + ".rank0")
    losses, W, H, Bs = read_result(res2, c)
    assert rel_fro(W, c.WT) < 1e-4 and rel_fro(H, c.HT) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["kl_2cov_nan", "als_kl"])
def test_c_example_devices_mode_native_rccl(name, tmp_path):
    """`fit_c --devices 1`: ONE process, alpine_comm_init_all (ncclCommInitAll over the ctx's device), a host thread drives the ctx and
    alpine_run enqueues the all-reduce itself; BITWISE the plain single-GPU result (a one-rank sum changes nothing)."""
    exe = build_example()
    c = load_case(name)
    flags = 16 | (4 if c.params.get("use_als") else 0)
    prob, res1, res2 = tmp_path / "p.bin", tmp_path / "r1.bin", tmp_path / "r2.bin"
    write_problem(prob, c, flags)
    r = subprocess.run([exe, str(prob), str(res1)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "--devices", "1", str(prob), str(res2)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    assert open(res1, "rb").read() == open(res2, "rb").read()
