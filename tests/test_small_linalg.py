"""CPU: the host-side k x k Krylov-space algebra of the C-ABI library
(csrc/small_linalg.h), compiled with g++ into a throw-away shim and compared
with SciPy -- the routines the reference calls (_integrator.py:402-408, :617-637)."""

import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import scipy.linalg

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    out = tmp_path_factory.mktemp("sl") / "libsl.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(HERE, "helpers", "small_linalg_shim.cpp"), "-o", str(out)], check=True)
    return C.CDLL(str(out))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.mark.parametrize("k", [1, 2, 5, 13, 20, 64])
def test_tridiag_eigvec(shim, k):
    rng = np.random.default_rng(k)
    a, b = rng.standard_normal(k), np.abs(rng.standard_normal(max(k - 1, 1))) + 0.1
    w, v = scipy.linalg.eigh_tridiagonal(a, b[: k - 1]) if k > 1 else (a.copy(), np.ones((1, 1)))
    for root in (0, -1):
        vec, val = np.zeros(k), C.c_double()
        assert shim.sl_tridiag_eigvec(_dp(a), _dp(b), k, root, _dp(vec), C.byref(val)) == 0
        ref = v[:, root] * np.sign(v[0, root] if v[0, root] != 0 else 1.0)
        assert abs(val.value - w[root]) < 1e-13 * max(1, abs(w).max())
        np.testing.assert_allclose(vec, ref, atol=1e-10)


@pytest.mark.parametrize("k", [2, 7, 20])
def test_expm_tridiag_and_dense(shim, k):
    rng = np.random.default_rng(100 + k)
    a, b = rng.standard_normal(k), np.abs(rng.standard_normal(k)) + 0.1
    T = np.diag(a) + np.diag(b[: k - 1], 1) + np.diag(b[: k - 1], -1)
    for scale in (-0.3j, -0.8, 0.2 + 0.5j):
        out = np.zeros(2 * k)
        z = complex(scale)
        shim.sl_expm_tridiag(_dp(a), _dp(b), k, C.c_double(z.real), C.c_double(z.imag), _dp(out))
        ref = scipy.linalg.expm(scale * T)[:, 0]
        np.testing.assert_allclose(out[0::2] + 1j * out[1::2], ref, atol=1e-13)
    G = rng.standard_normal((k, k)) + 1j * rng.standard_normal((k, k))  # non-normal (Arnoldi Hessenberg-like)
    G = np.triu(G, -1) * 0.7
    m = np.ascontiguousarray(np.stack([G.real, G.imag], axis=-1)).reshape(-1)
    out = np.zeros(2 * k)
    shim.sl_expm_col0(_dp(m), k, _dp(out))
    np.testing.assert_allclose(out[0::2] + 1j * out[1::2], scipy.linalg.expm(G)[:, 0], atol=1e-12, rtol=1e-12)


def test_host_algebra_under_address_and_ub_sanitizers(tmp_path):
    """The host-side algebra compiled with -fsanitize=address,undefined and run standalone
    (GPU sanitizers are unavailable on the pool; this is the part of the library that can be)."""
    exe = tmp_path / "sl_san"
    src = os.path.join(HERE, "helpers", "small_linalg_sanitize.cpp")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", src, "-o", str(exe)],
                       capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in (r.stderr or ""):
        pytest.skip("sanitizer runtime not installed: " + r.stderr.splitlines()[-1])
    assert r.returncode == 0, r.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", LD_PRELOAD=""))
    assert run.returncode == 0, run.stdout + run.stderr
    assert "0 failures" in run.stdout
