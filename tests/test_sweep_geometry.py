"""The stream-K index arithmetic shared by the sweep kernels and the consumers of their pieces, checked on the host
(tests/native/geom_check.hip includes alpine_amd/csrc/kernels.hpp and is compiled with hipcc; nothing runs on a GPU)."""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spans_cover_everything_once_and_consumers_find_every_piece(tmp_path):
    exe = str(tmp_path / "geom_check")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-o", exe,
                        os.path.join(REPO, "tests", "native", "geom_check.hip")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, "3000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "0 bad" in r.stdout
