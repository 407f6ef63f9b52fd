"""GPU: the gauge-free thin QR of the sweep's gauge moves (csrc/qr_gram.hip, csrc/qr.h qr_thin).

SiteCoef.gauge_trf (_site_cls.py:138-292) hands the sweep LAPACK's Q and R; what the sweep uses of them is Q R = psi and
Q^H Q = 1 -- the signs of diag(R) are a gauge freedom of the bond ("compare Q R and
Q^H Q, not Q itself").  The gauge-free path returns R with a POSITIVE diagonal, i.e. LAPACK's factors up to a diagonal
matrix of signs D: Q_lapack = Q D, R_lapack = D R."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rand(m, n, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((m, n)) + 1j * rng.standard_normal((m, n))


def _check(a, q, r, tol_rec=2e-13, tol_orth=1e-13):
    m, n = a.shape
    assert np.abs(q @ r - a).max() <= tol_rec * np.abs(a).max() * np.sqrt(n)
    assert np.abs(q.conj().T @ q - np.eye(n)).max() < tol_orth
    assert np.abs(np.tril(r, -1)).max() == 0.0


@pytest.mark.parametrize("shape", [(4096, 128), (1024, 128), (4096, 32), (2048, 512), (600, 200), (700, 72), (8192, 256), (400, 16), (900, 48), (1000, 64), (3000, 100)])
def test_gauge_free_factorisation_equals_lapack_up_to_signs(shape):
    from pytdscf_amd.engine import qr_thin

    m, n = shape
    a = _rand(m, n, 7 + m + n)
    q, r, info = qr_thin(a, gauge_free=True)
    assert info["gauge_free_path"], info
    _check(a, q, r)
    d = np.diag(r)
    assert np.abs(d.imag).max() == 0.0 and d.real.min() > 0.0  # the gauge fixed by this path: a positive diagonal
    qn, rn = np.linalg.qr(a)  # LAPACK
    sg = np.sign(np.diag(rn).real)
    assert np.abs(sg[:, None] * rn - r).max() < 1e-11 * np.abs(r).max()
    assert np.abs(qn * sg[None, :] - q).max() < 1e-11
    # the Householder path through the same entry point carries LAPACK's signs
    q2, r2, info2 = qr_thin(a, gauge_free=False)
    assert not info2["gauge_free_path"]
    assert np.abs(r2 - rn).max() < 1e-11 * np.abs(r).max() and np.abs(q2 - qn).max() < 1e-11


def test_ill_conditioned_and_rank_deficient_inputs_fall_back():
    """cond(A)^2 eps < 1 is what Cholesky-based orthogonalisation needs: a graded matrix (12 decades) and a rank-deficient
    one (the zero-padded states the reference starts from) fail the device-side pivot checks and
    are factored by the Householder panels -- same entry point, orthonormal Q either way."""
    from pytdscf_amd.engine import qr_thin

    m, n = 2048, 128
    a = _rand(m, n, 3) * np.logspace(0, -12, n)[None, :]
    u, _, vh = np.linalg.svd(_rand(n, n, 4))
    a = a @ vh  # graded singular values, no graded columns
    q, r, info = qr_thin(a, gauge_free=True)
    assert not info["gauge_free_path"]
    _check(a, q, r, tol_rec=1e-12)
    b = _rand(m, n, 5)
    b[:, 40:] = 0.0
    q, r, info = qr_thin(b, gauge_free=True)
    assert not info["gauge_free_path"]
    _check(b, q, r)
    # moderately graded (cond 1e5): well inside the gauge-free path's range, full accuracy
    c = _rand(m, n, 6) @ (np.logspace(0, -5, n)[:, None] * vh)
    q, r, info = qr_thin(c, gauge_free=True)
    assert info["gauge_free_path"]
    _check(c, q, r)


def test_c4_shape_and_timing():
    from pytdscf_amd.engine import qr_thin

    a = _rand(16384, 1024, 11)
    q, r, info = qr_thin(a, gauge_free=True, reps=3)
    assert info["gauge_free_path"]
    _check(a, q, r, tol_rec=3e-13, tol_orth=2e-13)
    _, _, ih = qr_thin(shape=(16384, 1024), gauge_free=False, reps=3)
    print(f"16384 x 1024: gauge-free {info['ms']:.2f} ms / {info['launches']} launches, Householder panels {ih['ms']:.2f} ms / {ih['launches']}")
    for shape in ((4096, 128), (2048, 512)):
        _, _, i1 = qr_thin(shape=shape, gauge_free=True, reps=10)
        _, _, i0 = qr_thin(shape=shape, gauge_free=False, reps=10)
        assert i1["gauge_free_path"]
        print(f"{shape}: gauge-free {i1['ms']:.3f} ms / {i1['launches']} launches, Householder panels {i0['ms']:.3f} ms / {i0['launches']}")
        assert i1["ms"] < i0["ms"]


def test_sweep_is_the_same_state_with_either_gauge(monkeypatch):
    """A mid-size chain propagated with the gauge-free moves and with LAPACK's signs: the same state (fidelity), the same
    energy and norm, the same Krylov counts -- bond bases differ by signs only."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine
    from pytdscf_amd import synthetic as syn

    _sweep_either_gauge(monkeypatch, 5, 8, 64, 6, 0.7)


def test_small_regime_sweep_is_the_same_state_with_either_gauge(monkeypatch):
    """The same at BASELINE configs[1]'s shape, where every gauge move is the one-workgroup kernel: its sign chain is
    skipped when MITDVP_QR_SMALL_GAUGE_FREE=1 asks for it (off by default: the small-size parity tests compare tensors)."""
    monkeypatch.setenv("MITDVP_QR_SMALL_GAUGE_FREE", "1")
    _sweep_either_gauge(monkeypatch, 10, 10, 32, 6, 2.0)


def _sweep_either_gauge(monkeypatch, L, d, D, M, dt):
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine
    from pytdscf_amd import synthetic as syn

    mpo = syn.synthetic_mpo(L, d, M, seed=0)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MITDVP_QR_GAUGE_FREE", mode)
        e = TDVPEngine(L)
        e.set_mpo(mpo)
        e.init_random([d] * L, D, seed=5)
        for _ in range(3):
            e.propagate(dt)
        out[mode] = (e.get_mps(), e.krylov_stats(), e.expectation(), e.norm())
        e.close()
    a, b = out["1"], out["0"]
    assert a[1] == b[1]
    assert abs(abs(orc.overlap(a[0], b[0])) - 1) < 1e-10
    assert abs(a[2] - b[2]) < 1e-10 * abs(b[2]) and abs(a[3] - 1) < 1e-12 and abs(b[3] - 1) < 1e-12
    # the two runs really took different gauges somewhere (otherwise this test compares a path with itself)
    assert max(np.abs(x - y).max() for x, y in zip(a[0], b[0])) > 1e-3
