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


def test_ctypes_structs_match_the_header(tmp_path):
    """sizeof / offsetof of the two ABI structs as gcc lays them out vs the ctypes mirror."""
    import ctypes as C

    from pytdscf_amd import _lib

    src = tmp_path / "lay.c"
    fields_cfg = [f for f, _ in _lib.Config._fields_]
    fields_cnt = [f for f, _ in _lib.Counters._fields_]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "mitdvp.h"', "int main(void) {",
             '  printf("%zu %zu\\n", sizeof(mitdvp_config), sizeof(mitdvp_counters));']
    for f in fields_cfg:
        lines.append(f'  printf("cfg {f} %zu\\n", offsetof(mitdvp_config, {f}));')
    for f in fields_cnt:
        lines.append(f'  printf("cnt {f} %zu\\n", offsetof(mitdvp_counters, {f}));')
    lines += ["  return 0;", "}"]
    src.write_text("\n".join(lines))
    exe = tmp_path / "lay"
    r = subprocess.run(["gcc", "-std=gnu11", str(src), "-I" + os.path.join(ROOT, "include"), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True).stdout.splitlines()
    assert out[0].split() == [str(C.sizeof(_lib.Config)), str(C.sizeof(_lib.Counters))]
    for line in out[1:]:
        kind, name, off = line.split()
        cls = _lib.Config if kind == "cfg" else _lib.Counters
        assert getattr(cls, name).offset == int(off), line
