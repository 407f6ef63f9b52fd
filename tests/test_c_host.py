"""The C ABI from a plain C host program (examples/c_host.c): it compiles against include/mitdvp.h
and links against the shared library with gcc (CPU), and runs a short propagation (GPU)."""

import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pytdscf_amd", "csrc")


def _build(out):
    cmd = ["gcc", "-std=gnu11", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "examples", "c_host.c"), "-I" + os.path.join(ROOT, "include"),
           "-L" + CSRC, "-lmitdvp", "-lm", "-Wl,-rpath," + CSRC, "-o", str(out)]
    return subprocess.run(cmd, capture_output=True, text=True)


def test_c_host_compiles_and_links(tmp_path):
    if not os.path.exists(os.path.join(CSRC, "libmitdvp.so")):
        pytest.skip("library not built")
    r = _build(tmp_path / "c_host")
    assert r.returncode == 0, r.stderr


@pytest.mark.gpu
def test_c_host_runs(tmp_path):
    r = _build(tmp_path / "c_host")
    assert r.returncode == 0, r.stderr
    run = subprocess.run([str(tmp_path / "c_host")], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "C-HOST OK" in run.stdout, run.stdout + run.stderr
