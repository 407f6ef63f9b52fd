"""GPU: kernel-level checks of the HIP building blocks through the C ABI,
against NumPy (complex128).  Tolerances: rounding-level (1e-12 relative to the
operand norms) -- these are the same arithmetic in a different summation order."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def crandn(rng, *shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


def test_mfma_layout_and_peak():
    from pytdscf_amd import engine as E

    lay = E.mfma_layout_probe()
    assert (lay >= 0).all()  # every lane/register value found in the reference product
    lanes = np.arange(64)
    assert (lay[:, :, 1] == (lanes & 15)[:, None]).all()  # col = lane & 15
    tf = E.mfma_peak_probe()
    print(f"v_mfma_f64_16x16x4_f64 issue-rate probe: {tf:.1f} TFLOP/s")
    assert tf > 20.0


@pytest.mark.parametrize("k", [1, 3, 15, 16, 17, 31, 32, 33, 47, 48, 49, 64, 65, 81])
def test_zgemm_short_and_ragged_contractions(k):
    """The K loop is peeled: tiles whose prefetch target lies inside K run a form without selects, the last two a general
    one whose out-of-range elements are buffer loads that return zero, the tile loop is unrolled by the staging-register
    parity.  Every count of K tiles from one to six, with and without a tail, in the four operand forms and both complex
    products, against NumPy."""
    from pytdscf_amd import engine as E

    rng = np.random.default_rng(1000 + k)
    m, n = 130, 70
    for tA, tB, cA, cB in [(0, 0, 0, 0), (0, 1, 0, 1), (1, 0, 1, 0), (1, 1, 0, 0)]:
        A = crandn(rng, *((k, m) if tA else (m, k)))
        B = crandn(rng, *((n, k) if tB else (k, n)))
        opA = (A.T if tA else A)
        opB = (B.T if tB else B)
        opA = opA.conj() if cA else opA
        opB = opB.conj() if cB else opB
        ref = opA @ opB
        for mode in ("3m", "4m"):
            E.set_gemm_mode(mode)
            try:
                for tile in (1, 2):
                    out = E.zgemm(A, B, None, bool(tA), bool(cA), bool(tB), bool(cB), tile_cfg=tile)
                    assert np.abs(out - ref).max() / np.abs(ref).max() < 1e-13, (k, tA, tB, mode, tile)
            finally:
                E.set_gemm_mode("3m")


@pytest.mark.parametrize("tile", [0, 1, 2])
@pytest.mark.parametrize(
    "tA,cA,tB,cB", [(0, 0, 0, 0), (0, 0, 1, 0), (1, 1, 0, 0), (1, 0, 0, 0), (0, 0, 1, 1), (1, 1, 1, 1), (0, 1, 0, 1)]
)
def test_zgemm_forms(tile, tA, cA, tB, cB):
    from pytdscf_amd import engine as E

    rng = np.random.default_rng(tile * 100 + tA * 8 + cA * 4 + tB * 2 + cB)
    m, n, k = 150, 77, 45  # ragged on purpose
    A = crandn(rng, *((k, m) if tA else (m, k)))
    B = crandn(rng, *((n, k) if tB else (k, n)))
    C0 = crandn(rng, m, n)
    opA = A.T if tA else A
    opB = B.T if tB else B
    if cA:
        opA = opA.conj()
    if cB:
        opB = opB.conj()
    alpha, beta = 0.7 - 0.2j, -0.3 + 1.1j
    ref = alpha * (opA @ opB) + beta * C0
    out = E.zgemm(A, B, C0, bool(tA), bool(cA), bool(tB), bool(cB), alpha, beta, tile_cfg=tile)
    err = np.abs(out - ref).max() / np.abs(ref).max()
    assert err < 1e-13


@pytest.mark.parametrize("shape", [(1, 1, 1), (1, 5, 300), (300, 1, 7), (16, 16, 4), (257, 129, 1), (33, 65, 130)])
def test_zgemm_edges(shape):
    from pytdscf_amd import engine as E

    m, n, k = shape
    rng = np.random.default_rng(m * 7 + n * 3 + k)
    A, B = crandn(rng, m, k), crandn(rng, k, n)
    for tile in (-1, 0, 1, 2):
        out = E.zgemm(A, B, tile_cfg=tile)
        assert np.abs(out - A @ B).max() <= 1e-13 * max(1.0, np.abs(A @ B).max())


@pytest.mark.parametrize("shape", [(32, 992, 16384), (32, 32, 8192), (70, 130, 4100), (1, 1, 5000)])
@pytest.mark.parametrize("mode", ["3m", "4m"])
def test_zgemm_split_k(shape, mode):
    """Skinny outputs with a long contraction take the deterministic split-K path."""
    from pytdscf_amd import engine as E

    m, n, k = shape
    rng = np.random.default_rng(m + n + k)
    A, B, C0 = crandn(rng, k, m), crandn(rng, k, n), crandn(rng, m, n)
    E.set_gemm_mode(mode)
    try:
        out = E.zgemm(A, B, C0, transA=True, conjA=True, alpha=-1.0, beta=0.5 + 0.25j)
        out2 = E.zgemm(A, B, C0, transA=True, conjA=True, alpha=-1.0, beta=0.5 + 0.25j)
    finally:
        E.set_gemm_mode("3m")
    ref = -(A.conj().T @ B) + (0.5 + 0.25j) * C0
    assert np.abs(out - ref).max() < 1e-13 * np.abs(ref).max() * np.sqrt(k)
    assert np.array_equal(out, out2)  # deterministic combine


def test_zgemm_large_and_rate():
    from pytdscf_amd import engine as E

    rng = np.random.default_rng(5)
    m, n, k = 1024, 1024, 512
    A, B = crandn(rng, m, k), crandn(rng, k, n)
    out, ms = E.zgemm(A, B, reps=3)
    ref = A @ B
    assert np.abs(out - ref).max() / np.abs(ref).max() < 1e-13
    print(f"zgemm {m}x{n}x{k}: {ms:.3f} ms = {8.0 * m * n * k / ms / 1e9:.2f} TFLOP/s")


def test_heff_keff_env_vs_numpy():
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import engine as E

    rng = np.random.default_rng(11)
    for dl, d, dr, ml, mr in [(5, 3, 4, 3, 2), (1, 4, 6, 1, 5), (7, 2, 1, 4, 1), (40, 6, 33, 7, 5)]:
        L, R = crandn(rng, dl, ml, dl), crandn(rng, dr, mr, dr)
        W, psi = crandn(rng, ml, d, d, mr), crandn(rng, dl, d, dr)
        ref = orc.heff_apply(L, W, R, psi)
        out = E.heff_apply(L, W, R, psi)
        assert np.abs(out - ref).max() < 1e-12 * np.abs(ref).max()
        A = crandn(rng, dl, d, dr)
        refl = orc.env_update_left(L, A, W)
        outl = E.env_update(L, A, W, left=True)
        assert np.abs(outl - refl).max() < 1e-12 * np.abs(refl).max()
        refr = orc.env_update_right(R, A, W)
        outr = E.env_update(R, A, W, left=False)
        assert np.abs(outr - refr).max() < 1e-12 * np.abs(refr).max()
        Rk, sv = crandn(rng, dr, ml, dr), crandn(rng, dl, dr)
        refk = orc.keff_apply(L, Rk, sv)
        outk = E.keff_apply(L, Rk, sv)
        assert np.abs(outk - refk).max() < 1e-12 * np.abs(refk).max()


@pytest.mark.parametrize("shape", [(4, 3, 5), (1, 6, 4), (8, 5, 8), (40, 4, 70), (33, 9, 33), (64, 3, 100)])
def test_gauge_trf_vs_lapack(shape):
    """Householder QR with LAPACK's conventions: for full-rank input Q and R
    agree with scipy's (the reference's, _site_cls.py:264-282) to rounding."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import engine as E

    dl, d, dr = shape
    rng = np.random.default_rng(dl * 100 + d * 10 + dr)
    psi = crandn(rng, dl, d, dr)
    if dl * d >= dr:
        A, s = E.gauge_trf(psi, "Psi2Asigma")
        Ar, sr = orc.qr_psi2Asigma(psi)
        Am = A.reshape(dl * d, dr)
        assert np.abs(Am.conj().T @ Am - np.eye(dr)).max() < 1e-14 * dr
        assert np.abs(np.tensordot(A, s, axes=(2, 0)) - psi).max() < 1e-13 * np.abs(psi).max() * dr
        assert np.abs(np.tril(s, -1)).max() == 0.0
        np.testing.assert_allclose(A, Ar, atol=1e-11)
        np.testing.assert_allclose(s, sr, atol=1e-11)
    if dr * d >= dl:
        B, s = E.gauge_trf(psi, "Psi2sigmaB")
        sr, Br = orc.qr_psi2sigmaB(psi)
        Bm = B.reshape(dl, d * dr)
        assert np.abs(Bm @ Bm.conj().T - np.eye(dl)).max() < 1e-14 * dl
        assert np.abs(np.tensordot(s, B, axes=(1, 0)) - psi).max() < 1e-13 * np.abs(psi).max() * dl
        np.testing.assert_allclose(B, Br, atol=1e-11)
        np.testing.assert_allclose(s, sr, atol=1e-11)


def test_gauge_trf_rank_deficient():
    """Zero-padded product state (reference initial states, _site_cls.py:444-448):
    Q must still be a complete orthonormal set."""
    from pytdscf_amd import engine as E

    psi = np.zeros((6, 4, 6), dtype=np.complex128)
    psi[0, :, 0] = [0.5, 0.5, 0.5, 0.5]
    A, s = E.gauge_trf(psi, "Psi2Asigma")
    Am = A.reshape(24, 6)
    assert np.abs(Am.conj().T @ Am - np.eye(6)).max() < 1e-14
    assert np.abs(np.tensordot(A, s, axes=(2, 0)) - psi).max() < 1e-14


@pytest.mark.parametrize("shape", [(128, 4, 128), (512, 4, 512), (100, 5, 77), (64, 3, 160), (40, 16, 300)])
def test_qr_panels_by_cholesky_qr2_and_householder_reconstruction(shape):
    """The panel factorisation of the gauge move (csrc/qr_fast.hip): CholeskyQR2 of each 32-column panel, then LAPACK's
    Householder vectors / T / tau / signs of diag(R) reconstructed from the orthonormal panel.  Q and R must equal
    scipy's (the reference's zgeqrf + zungqr, _site_cls.py:264-282) to 1e-11 like the per-column Householder kernels,
    and both forms must agree with each other far below that."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import engine as E

    dl, d, dr = shape
    rng = np.random.default_rng(dl + 7 * d + 13 * dr)
    psi = crandn(rng, dl, d, dr)
    res = {}
    for fast in (True, False):
        E.set_qr_fast(fast)
        try:
            res[fast] = [E.gauge_trf(psi, "Psi2Asigma") if dl * d >= dr else None,
                         E.gauge_trf(psi, "Psi2sigmaB") if dr * d >= dl else None]
        finally:
            E.set_qr_fast(True)
    if res[True][0] is not None:
        A, s = res[True][0]
        Ar, sr = orc.qr_psi2Asigma(psi)
        np.testing.assert_allclose(A, Ar, atol=1e-11)
        np.testing.assert_allclose(s, sr, atol=1e-11 * np.abs(sr).max())
        Am = A.reshape(dl * d, dr)
        assert np.abs(Am.conj().T @ Am - np.eye(dr)).max() < 1e-14 * dr
        assert np.abs(np.tril(s, -1)).max() == 0.0 and np.abs(np.diag(s).imag).max() == 0.0
        assert np.abs(A - res[False][0][0]).max() < 1e-12 and np.abs(s - res[False][0][1]).max() < 1e-12 * np.abs(sr).max()
    if res[True][1] is not None:
        B, s = res[True][1]
        sr, Br = orc.qr_psi2sigmaB(psi)
        np.testing.assert_allclose(B, Br, atol=1e-11)
        np.testing.assert_allclose(s, sr, atol=1e-11 * np.abs(sr).max())
        assert np.abs(B - res[False][1][0]).max() < 1e-12


def test_zgemm_wide_row_strides():
    """The staging loads of the MFMA GEMM address a tile with 32-bit per-thread byte offsets: a row stride is fine as long
    as (rows of the operand a tile spans) x stride x 16 B < 2^32 -- 2^20 elements (D = 2048, d = 16, M = 32) must work,
    a stride that would wrap is refused with EINVAL instead of being mis-addressed (advisor finding of round 3)."""
    from pytdscf_amd import engine as E

    rng = np.random.default_rng(5)
    k = (1 << 20) + 16
    A = (rng.standard_normal((40, k)) + 1j * rng.standard_normal((40, k))) / np.sqrt(k)
    B = rng.standard_normal((k, 24)) + 1j * rng.standard_normal((k, 24))
    C = E.zgemm(A, B)  # lda = 2^20 + 16 untransposed (split-K path), ldb = 24
    ref = A @ B
    assert np.abs(C - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())
    Bs = np.ascontiguousarray(B.T)  # (24, k): op(B) = Bs^T, ldb = 2^20 + 16
    C2 = E.zgemm(A, Bs, transB=True)
    assert np.abs(C2 - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())
    del A, B, Bs
    As = rng.standard_normal((32, k)) + 1j * rng.standard_normal((32, k))  # stored (K, M): op(A) = As^T, lda = 2^20 + 16
    B3 = rng.standard_normal((32, 8)) + 1j * rng.standard_normal((32, 8))
    C3 = E.zgemm(As, B3, transA=True)
    assert np.abs(C3 - As.T @ B3).max() < 1e-11 * np.abs(C3).max()
    del As, C3
    wide = (1 << 22) + 64  # a 128-row tile of an untransposed operand with this stride would wrap 32 bits
    A = np.zeros((2, wide), dtype=np.complex128)
    B = np.zeros((wide, 2), dtype=np.complex128)
    with pytest.raises(ValueError):
        E.zgemm(A, B, tile_cfg=0)


@pytest.mark.parametrize("shape", [(32, 10, 32), (10, 10, 20), (16, 2, 32), (8, 4, 32), (1, 5, 3), (32, 10, 1), (7, 3, 17),
                                   (20, 16, 31), (3, 1, 3)])
def test_small_qr_in_one_workgroup(shape):
    """Gauge moves of the small-bond regime (m <= 320, n <= 32; csrc/qr_fast.hip::k_qr_small_fast): CholeskyQR2 with the
    Gram matrices and applies on the matrix cores in ONE workgroup, LAPACK's signs of diag(R) from the LU chain of the
    Householder reconstruction.  Q and R must equal scipy's (_site_cls.py:264-282) like the per-column Householder kernel
    they replace, and agree with that kernel far below the tolerance."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import engine as E

    dl, d, dr = shape
    rng = np.random.default_rng(1000 * dl + 10 * d + dr)
    psi = crandn(rng, dl, d, dr)
    res = {}
    for fast in (True, False):
        E.set_qr_fast(fast)
        try:
            res[fast] = E.gauge_trf(psi, "Psi2Asigma")
        finally:
            E.set_qr_fast(True)
    A, s = res[True]
    Ar, sr = orc.qr_psi2Asigma(psi)
    np.testing.assert_allclose(A, Ar, atol=1e-11)
    np.testing.assert_allclose(s, sr, atol=1e-11 * np.abs(sr).max())
    Am = A.reshape(dl * d, dr)
    assert np.abs(Am.conj().T @ Am - np.eye(dr)).max() < 1e-14 * max(dr, 4)
    assert np.abs(np.tensordot(A, s, axes=(2, 0)) - psi).max() < 1e-13 * np.abs(psi).max() * max(dr, 4)
    assert np.abs(np.tril(s, -1)).max() == 0.0 and np.abs(np.diag(s).imag).max() == 0.0
    assert np.abs(A - res[False][0]).max() < 1e-12 and np.abs(s - res[False][1]).max() < 1e-12 * np.abs(sr).max()


@pytest.mark.parametrize("shape", [(10, 10, 32, 6), (8, 4, 12, 4)])
def test_small_regime_trajectory_is_the_same_with_either_qr(shape, monkeypatch):
    """A few time steps of the small-bond regime with the gauge moves by the one-workgroup CholeskyQR2 and by the
    per-column Householder kernel: asked for LAPACK's signs (MITDVP_QR_GAUGE_FREE=0; the sweep's default since round 5 is
    the gauge-free form, tests/test_gpu_qr_gauge_free.py) both return LAPACK's Q / R, so the propagated tensors themselves
    must agree, with equal Krylov counts (the reference's gauge move: _site_cls.py:138-292)."""
    from pytdscf_amd import TDVPEngine

    monkeypatch.setenv("MITDVP_QR_GAUGE_FREE", "0")
    from pytdscf_amd import engine as E
    from pytdscf_amd import synthetic as syn

    L, d, D, M = shape
    mpo = syn.synthetic_mpo(L, d, M, seed=0)
    res = {}
    for fast in (True, False):
        E.set_qr_fast(fast)
        try:
            eng = TDVPEngine(L)
            eng.set_mpo(mpo)
            eng.init_random([d] * L, D, seed=3)
            for _ in range(3):
                eng.propagate(2.0)
            res[fast] = (eng.get_mps(), eng.krylov_stats(), eng.norm())
            eng.close()
        finally:
            E.set_qr_fast(True)
    assert res[True][1] == res[False][1]
    assert abs(res[True][2] - 1) < 1e-12
    for a, b in zip(res[True][0], res[False][0]):
        assert np.abs(a - b).max() < 1e-10


def test_small_qr_random_shapes():
    """Forty random (m, n) with m <= 320, n <= min(m, 32) -- odd sizes, m not a multiple of the 4-row MFMA step or the
    16-row blocks, n on either side of the 16-column block edge -- against scipy's QR (LAPACK's signs)."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import engine as E

    rng = np.random.default_rng(2024)
    for _ in range(40):
        d = int(rng.integers(1, 11))
        dl = int(rng.integers(1, 320 // d + 1))
        dr = int(rng.integers(1, min(32, dl * d) + 1))
        psi = crandn(rng, dl, d, dr)
        A, s = E.gauge_trf(psi, "Psi2Asigma")
        Ar, sr = orc.qr_psi2Asigma(psi)
        assert np.abs(A - Ar).max() < 1e-10, (dl, d, dr)
        assert np.abs(s - sr).max() < 1e-10 * max(1.0, np.abs(sr).max()), (dl, d, dr)
        Am = A.reshape(dl * d, dr)
        assert np.abs(Am.conj().T @ Am - np.eye(dr)).max() < 1e-13, (dl, d, dr)


def test_small_qr_falls_back_on_rank_deficient_graded_and_nonfinite_input():
    """The one-workgroup CholeskyQR2 must hand zero-padded product states and strongly graded tensors to the Householder
    kernel queued behind it (no host decision in between), and its verdict must not leak into the next factorisation."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import engine as E

    psi = np.zeros((16, 4, 16), dtype=np.complex128)
    psi[0, :, 0] = [0.5, 0.5, 0.5, 0.5]
    rng = np.random.default_rng(3)
    good = crandn(rng, 16, 4, 16)
    for _ in range(2):
        A, s = E.gauge_trf(psi, "Psi2Asigma")
        Ar, sr = orc.qr_psi2Asigma(psi)
        Am = A.reshape(64, 16)
        assert np.abs(Am.conj().T @ Am - np.eye(16)).max() < 1e-14
        assert np.abs(np.tensordot(A, s, axes=(2, 0)) - psi).max() < 1e-14
        np.testing.assert_allclose(A, Ar, atol=1e-12)
        A2, s2 = E.gauge_trf(good, "Psi2Asigma")  # a well-conditioned one right after a failed one
        Ar2, sr2 = orc.qr_psi2Asigma(good)
        np.testing.assert_allclose(A2, Ar2, atol=1e-11)
        np.testing.assert_allclose(s2, sr2, atol=1e-11)
    x = crandn(rng, 16, 4, 24) * (10.0 ** (-1.0 * np.arange(24)))[None, None, :]  # columns graded over 24 decades
    A, s = E.gauge_trf(x, "Psi2Asigma")
    Ar, sr = orc.qr_psi2Asigma(x)
    Am = A.reshape(64, 24)
    assert np.abs(Am.conj().T @ Am - np.eye(24)).max() < 1e-13
    assert np.abs(np.tensordot(A, s, axes=(2, 0)) - x).max() < 1e-14
    np.testing.assert_allclose(s, sr, atol=1e-11)
    # moderately ill-conditioned (cond ~ 1e4): still the fast path's business, orthogonality must be LAPACK's
    y = crandn(rng, 32, 10, 32) * (10.0 ** (-4.0 * np.arange(32) / 31))[None, None, :]
    A, s = E.gauge_trf(y, "Psi2Asigma")
    Ar, sr = orc.qr_psi2Asigma(y)
    Am = A.reshape(320, 32)
    assert np.abs(Am.conj().T @ Am - np.eye(32)).max() < 1e-13
    np.testing.assert_allclose(A, Ar, atol=1e-10)
    np.testing.assert_allclose(s, sr, atol=1e-11)


def test_qr_fast_panels_fall_back_on_rank_deficient_and_graded_input():
    """CholeskyQR2 cannot factor a rank-deficient panel: the device-side pivot checks must send zero-padded product
    states (the reference's starts, _site_cls.py:444-448) and strongly graded tensors to the Householder kernels, whose
    orthonormal completion is what the reference's LAPACK call returns."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import engine as E

    assert E.get_qr_fast()
    psi = np.zeros((96, 4, 96), dtype=np.complex128)
    psi[0, :, 0] = [0.5, 0.5, 0.5, 0.5]
    for _ in range(3):  # repeated: the back-off after a failed attempt must not change results
        A, s = E.gauge_trf(psi, "Psi2Asigma")
        Am = A.reshape(384, 96)
        assert np.abs(Am.conj().T @ Am - np.eye(96)).max() < 1e-13
        assert np.abs(np.tensordot(A, s, axes=(2, 0)) - psi).max() < 1e-14
    rng = np.random.default_rng(8)
    x = crandn(rng, 96, 4, 96) * (10.0 ** (-0.25 * np.arange(96)))[None, None, :]  # columns graded over 24 decades
    A, s = E.gauge_trf(x, "Psi2Asigma")
    Ar, sr = orc.qr_psi2Asigma(x)
    Am = A.reshape(384, 96)
    assert np.abs(Am.conj().T @ Am - np.eye(96)).max() < 1e-13
    assert np.abs(np.tensordot(A, s, axes=(2, 0)) - x).max() < 1e-14
    np.testing.assert_allclose(s, sr, atol=1e-11)


@pytest.mark.parametrize("shape", [(1, 1), (2, 2), (7, 7), (33, 33), (64, 20), (20, 64), (128, 128), (257, 257), (140, 300), (300, 140)])
def test_jacobi_svd_vs_lapack(shape):
    """One-sided Jacobi SVD: singular values to working precision (small ones as
    well), orthonormal factors, exact reconstruction."""
    from pytdscf_amd import engine as E

    r, c = shape
    rng = np.random.default_rng(r * 1000 + c)
    k = min(r, c)
    A = crandn(rng, r, c) * (0.5 ** np.arange(c))[None, :] if c <= 40 else crandn(rng, r, c)
    U, S, Vh, sweeps = E.svd(A)
    Sref = np.linalg.svd(A, compute_uv=False)
    assert np.all(np.diff(S) <= 1e-300)  # descending
    np.testing.assert_allclose(S, Sref, rtol=1e-10, atol=1e-14 * Sref[0])
    assert np.abs(U.conj().T @ U - np.eye(k)).max() < 1e-13 * k
    assert np.abs(Vh @ Vh.conj().T - np.eye(k)).max() < 1e-13 * k
    assert np.abs((U * S) @ Vh - A).max() < 1e-13 * np.abs(A).max() * k
    assert sweeps <= 20


@pytest.mark.parametrize("precond", ["-1", "1", "0", "2"])
def test_jacobi_svd_graded_and_rank_deficient(precond):
    """The QR-preconditioned path (from 128 rows on; MITDVP_SVD_PRECOND in a child process: -1 = the default, a second LR step when R's diagonal is graded; 1 / 2 = one / two steps always; 0 = the plain path): a
    spectrum graded over twelve decades (singular values to the absolute accuracy eps * sigma_max of the input) needs few
    sweeps; a rank-deficient matrix (48 exact zeros) reconstructs, its non-zero part is orthonormal."""
    import subprocess
    import sys
    import textwrap

    code = textwrap.dedent("""
        import numpy as np
        from pytdscf_amd import engine as E
        rng = np.random.default_rng(11)
        n = 256
        A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        u, _, vh = np.linalg.svd(A)
        sv = np.logspace(0, -12, n)
        B = (u * sv) @ vh
        U, S, Vh, sweeps = E.svd(B)
        assert np.max(np.abs(S - sv) / sv) < %(tol)s, np.max(np.abs(S - sv) / sv)
        assert np.linalg.norm((U * S) @ Vh - B) < 1e-12
        assert np.abs(U.conj().T @ U - np.eye(n)).max() < 1e-11 and np.abs(Vh @ Vh.conj().T - np.eye(n)).max() < 1e-11
        assert sweeps <= %(sw)d, sweeps
        sv0 = np.linspace(1.0, 0.1, n); sv0[-48:] = 0.0
        C = (u * sv0) @ vh
        U, S, Vh, sweeps = E.svd(C)
        assert np.abs(S - sv0).max() < 1e-13 and np.linalg.norm((U * S) @ Vh - C) < 1e-12
        k = n - 48
        assert np.abs(U[:, :k].conj().T @ U[:, :k] - np.eye(k)).max() < 1e-11
        assert np.abs(Vh[:k] @ Vh[:k].conj().T - np.eye(k)).max() < 1e-11
        D = C[:, :180]                      # rectangular and rank-deficient
        U, S, Vh, sweeps = E.svd(D)
        assert np.abs(S - np.linalg.svd(D, compute_uv=False)).max() < 1e-13 and np.linalg.norm((U * S) @ Vh - D) < 1e-12
        print("ok", sweeps)
    """) % dict(tol="1e-3", sw=60 if precond == "0" else 20)
    env = dict(os.environ, MITDVP_SVD_PRECOND=precond)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("shape, rank", [((10, 10), 1), ((100, 4), 1), ((4, 100), 2), ((40, 40), 7), ((256, 256), 208), ((300, 140), 20)])
def test_jacobi_svd_completes_the_null_space(shape, rank):
    """A numerically rank-deficient input: BOTH factors come back unitary -- the singular vectors of the zero singular
    values are an orthonormal completion, as LAPACK's are (callers that keep the dimension rebuild environment blocks
    from A U / Vh B and lift zero singular values along those vectors: truncate_sigvec(keepdim=True),
    gauge_trf(regularize=True), _site_cls.py:207-246, :586-690).  Before round 5 they were rounding residue parallel to
    the leading vectors."""
    from pytdscf_amd import engine as E

    r, c = shape
    k = min(r, c)
    rng = np.random.default_rng(r * 7 + c)
    A = crandn(rng, r, rank) @ crandn(rng, rank, c)
    U, S, Vh, _ = E.svd(A)
    Sref = np.linalg.svd(A, compute_uv=False)
    assert np.abs(S - Sref).max() < 1e-12 * Sref[0]
    assert np.abs((U * S) @ Vh - A).max() < 1e-12 * Sref[0]
    assert np.abs(U.conj().T @ U - np.eye(k)).max() < 1e-12
    assert np.abs(Vh @ Vh.conj().T - np.eye(k)).max() < 1e-12


def test_concurrent_engines_from_host_threads():
    """Several engines (one HIP stream each) stepped concurrently from host threads give the
    same states as when they run one after the other: per-stream split-K workspaces, no shared
    mutable state between handles (an ensemble of trajectories shares one GPU this way)."""
    import threading

    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    L, d, M, D = 8, 4, 5, 24
    mpo = orc.synthetic_mpo(L, d, M, seed=1)

    def make(seed):
        e = TDVPEngine(L)
        e.set_mpo(mpo)
        e.init_random([d] * L, D, seed=seed)
        return e

    def run(e):
        for _ in range(4):
            e.propagate(0.7)

    serial = []
    for s in range(4):
        e = make(s + 1)
        run(e)
        serial.append(np.concatenate([c.reshape(-1) for c in e.get_mps()]))
        e.close()
    engs = [make(s + 1) for s in range(4)]
    th = [threading.Thread(target=run, args=(e,)) for e in engs]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e, ref in zip(engs, serial):
        got = np.concatenate([c.reshape(-1) for c in e.get_mps()])
        assert np.array_equal(got, ref)  # deterministic kernels: bit-identical
        e.close()


def test_persistent_launches_of_eight_engines_from_eight_host_threads():
    """8 engines x 200 time steps from 8 host threads on one GPU, every local exponential a persistent one-launch
    kernel (k_small_site needs all its workgroups resident together): persistent launches of ALL engines are admitted
    against a per-device budget of compute units (small_site.hip, PersistentLaunch), so the grids in flight always fit
    the chip together and no two partially resident grids can wait for each other until the exchange time-out.  Results are bit-identical to the same engines run one after the other."""
    import threading

    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    L, d, M, D, nstep, neng = 6, 4, 4, 16, 200, 8
    mpo = orc.synthetic_mpo(L, d, M, seed=2)

    def make(seed):
        e = TDVPEngine(L)
        e.set_mpo(mpo)
        e.init_random([d] * L, D, seed=seed)
        return e

    errors = []

    def run(e, n):
        try:
            for _ in range(n):
                e.propagate(0.3)
        except Exception as exc:  # noqa: BLE001 -- a time-out would surface here as MitdvpError
            errors.append(exc)

    ref = make(1)
    run(ref, nstep)
    assert not errors
    c0 = ref.counters()
    assert c0["n_launch"] < 40 * 2 * nstep  # the one-launch family is what ran (a multi-launch sweep needs > 500)
    want = np.concatenate([c.reshape(-1) for c in ref.get_mps()])
    ref.close()
    engs = [make(1) for _ in range(neng)]  # the same trajectory eight times: any cross-talk shows as a difference
    th = [threading.Thread(target=run, args=(e, nstep)) for e in engs]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for e in engs:
        got = np.concatenate([c.reshape(-1) for c in e.get_mps()])
        assert np.array_equal(got, want)
        assert abs(e.norm() - 1) < 1e-12
        e.close()


def test_block_sparse_w_stage_is_bitwise_the_dense_one(monkeypatch):
    """The W stage of an apply / environment update skips the zero (c, t) blocks of a finite-state-machine MPO
    (rows of W2 ordered (t, i), K-tile list per row tile, rows of Y mapped back): adding exact zeros changes
    nothing, so a sweep must equal the dense W stage bit for bit -- and the oracle to the usual tolerance."""
    import numpy as np

    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    L, d, M, D = 5, 16, 8, 64
    mpo = orc.synthetic_mpo(L, d, M, seed=0)
    mps = orc.synthetic_mps([d] * L, D, seed=1)
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("MITDVP_SPARSE_W", flag)
        monkeypatch.setenv("MITDVP_SMALL_KERNELS", "0")  # the general three-launch chain is what carries the sparse stage
        monkeypatch.setenv("MITDVP_EDGE_APPLY", "0")     # (not the two-product edge form, which has no W stage at all)
        eng = TDVPEngine(L)
        eng.set_mpo(mpo)
        eng.set_mps(mps)
        eng.propagate(0.3)
        out[flag] = (eng.get_mps(), eng.expectation(), eng.counters())
        eng.close()
    assert out["1"][2]["heff_flops_skipped"] > 0 and out["0"][2]["heff_flops_skipped"] == 0
    for a, b in zip(out["1"][0], out["0"][0]):
        assert np.array_equal(a, b)
    ref = orc.OracleMPS([c.copy() for c in mps], mpo)
    ref.propagate(0.3)
    assert abs(abs(orc.overlap(ref.cores, out["1"][0])) - 1) < 1e-10
    assert abs(out["1"][1] - ref.expectation()) < 1e-8 * abs(ref.expectation())


@pytest.mark.parametrize("form", ["nn", "nt"])
def test_zgemm_long_contraction_with_many_tiles_is_split(form):
    """>= 192 output tiles and K >= 8192 (stage S3 of an apply at large D): the contraction is split inside the launch
    and combined in a fixed order -- same numbers as the plain product to rounding, deterministic."""
    from pytdscf_amd import engine as E

    m, n, k = 1024, 896, 8192
    rng = np.random.default_rng(11)
    A = crandn(rng, m, k)
    B = crandn(rng, n, k) if form == "nt" else crandn(rng, k, n)
    out = E.zgemm(A, B, transB=form == "nt")
    out2 = E.zgemm(A, B, transB=form == "nt")
    ref = A @ (B.T if form == "nt" else B)
    assert np.abs(out - ref).max() < 1e-13 * np.abs(ref).max() * np.sqrt(k)
    assert np.array_equal(out, out2)


def test_fold_block_ranges_and_site_rdm_from_blocks():
    """mitdvp_fold_block_range / mitdvp_site_rdm_blocks (the per-rank pieces of the site-sharded observables) on a
    single engine with ragged bonds and tensors in no particular gauge, against plain einsum."""
    from pytdscf_amd import TDVPEngine

    rng = np.random.default_rng(5)
    dims = [(3, 2, 5), (5, 3, 4), (4, 2, 6), (6, 3, 2)]  # open outer bonds wider than 1: a block of a longer chain
    cores = [crandn(rng, *s) for s in dims]
    M = [2, 3, 2, 3, 2]
    mpo = [crandn(rng, M[p], dims[p][1], dims[p][1], M[p + 1]) for p in range(4)]
    eng = TDVPEngine(4)
    for p, c in enumerate(cores):
        eng.set_site(p, c, "C")
    eng.set_mpo(mpo, op_id=2)

    def left(t, x, conj=True):
        return np.einsum("ab,aic,bid->cd", t, x.conj() if conj else x, x, optimize=True)

    def right(t, x, conj=True):
        return np.einsum("cd,aic,bid->ab", t, x.conj() if conj else x, x, optimize=True)

    for conj in (True, False):
        t0 = crandn(rng, 3, 3)
        ref = t0
        for x in cores[:3]:
            ref = left(ref, x, conj)
        got = eng.fold_block(t0[:, None, :], op_id=-1, conj=conj, from_left=True, first=0, count=3)
        assert got.shape == (6, 1, 6) and np.abs(got[:, 0, :] - ref).max() < 1e-12 * np.abs(ref).max()
        t1 = crandn(rng, 2, 2)
        ref = t1
        for x in reversed(cores[1:]):
            ref = right(ref, x, conj)
        got = eng.fold_block(t1[:, None, :], op_id=-1, conj=conj, from_left=False, first=1, count=3)
        assert np.abs(got[:, 0, :] - ref).max() < 1e-12 * np.abs(ref).max()
    # operator block through all sites, both directions
    e0 = crandn(rng, 3, 2, 3)
    ref = e0
    for x, w in zip(cores, mpo):
        ref = np.einsum("acb,aix,cijt,bjy->xty", ref, x.conj(), w, x, optimize=True)
    got = eng.fold_block(e0, op_id=2, from_left=True)
    assert np.abs(got - ref).max() < 1e-12 * np.abs(ref).max()
    e1 = crandn(rng, 2, 2, 2)
    ref = e1
    for x, w in zip(reversed(cores), reversed(mpo)):
        ref = np.einsum("xty,aix,cijt,bjy->acb", ref, x.conj(), w, x, optimize=True)
    got = eng.fold_block(e1, op_id=2, from_left=False)
    assert np.abs(got - ref).max() < 1e-12 * np.abs(ref).max()
    # count = 0 hands the block back; a block of the wrong size is refused
    assert np.array_equal(eng.fold_block(e0, op_id=-1, first=2, count=0), e0)
    with pytest.raises(ValueError):
        eng.fold_block(crandn(rng, 4, 1, 4), op_id=-1, from_left=True)
    # one-site reduced density from the blocks on both sides
    tl, tr = crandn(rng, 5, 5), crandn(rng, 4, 4)
    ref = np.einsum("ab,bjs,ts,akt->jk", tl, cores[1], tr, cores[1].conj(), optimize=True)
    got = eng.site_rdm_blocks(1, tl, tr)
    assert np.abs(got - ref).max() < 1e-12 * np.abs(ref).max()
    eng.close()


def test_identity_block_of_the_right_environment_is_short_circuited(monkeypatch):
    """Bonds >= 256: when the last MPO-bond block of the right environment is the identity to 1e-13 (sites right of the
    centre right-canonical, "nothing applied yet" passed through by the MPO) its K block of stage S3 becomes a strided
    copy (the reference short-circuits such blocks too, _mps_mpo.py:510-523).  Same state as the full contraction to
    rounding; with a random dense MPO (no identity block) the check says no and nothing is skipped."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    L, d, M, D = 4, 16, 4, 256
    mps = orc.synthetic_mps([d] * L, D, seed=1)
    fsm = orc.synthetic_mpo(L, d, M, seed=0)
    rng = np.random.default_rng(3)
    dense = []
    for p in range(L):
        ml, mr = (1 if p == 0 else M), (1 if p == L - 1 else M)
        w = 0.02 * crandn(rng, ml, d, d, mr)
        dense.append(w)
    out = {}
    for name, mpo in (("fsm", fsm), ("dense", dense)):
        for flag in ("1", "0"):
            monkeypatch.setenv("MITDVP_TRIM_IDENTITY", flag)
            monkeypatch.setenv("MITDVP_SPARSE_W", "0")  # count only this shortcut in heff_flops_skipped
            eng = TDVPEngine(L, integrator="arnoldi" if name == "dense" else "lanczos")
            eng.set_mpo(mpo)
            eng.set_mps(mps)
            eng.sweep(0.2, True)
            out[name, flag] = (eng.get_mps(), eng.counters()["heff_flops_skipped"])
            eng.close()
    assert out["fsm", "1"][1] > 0 and out["fsm", "0"][1] == 0 and out["dense", "1"][1] == 0
    for name in ("fsm", "dense"):
        a, b = out[name, "1"][0], out[name, "0"][0]
        assert abs(abs(orc.overlap(a, b)) / np.sqrt(abs(orc.overlap(a, a)) * abs(orc.overlap(b, b))) - 1) < 1e-12
        assert max(np.abs(x - y).max() for x, y in zip(a, b)) < 1e-11


@pytest.mark.parametrize("kind", ["fsm", "liouville", "dense"])
def test_identity_states_are_skipped_in_the_bond_applies(kind, monkeypatch):
    """Bonds >= 256: the MPO-bond states whose left / right block is a multiple of the identity (canonical chain under a
    finite-state-machine MPO, or the direct sum of C5's Liouvillian: several such states, some carrying a sign) drop out
    of the first / second product of a K_eff apply -- scaled copies instead (Engine::keff_prepare; the reference skips such
    blocks, _mps_mpo.py:489-523).  Same state and Krylov counts as the full contraction, equal to the oracle; with a
    random dense MPO nothing qualifies and the plain apply runs."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine
    from pytdscf_amd import synthetic as syn

    rng = np.random.default_rng(5)
    if kind == "liouville":
        L, d, M, D, integ, cn = 8, 4, 16, 256, "arnoldi", False
        mpo = syn.synthetic_liouvillian_mpo(L, M, seed=0, gamma=0.002)
    else:
        L, d, M, D, integ, cn = 4, 16, 6, 256, "lanczos", True
        mpo = syn.synthetic_mpo(L, d, M, seed=0)
        if kind == "dense":
            mpo = [0.02 * crandn(rng, w.shape[0], d, d, w.shape[3]) for w in mpo]
            integ, cn = "arnoldi", False
    mps = orc.synthetic_mps([d] * L, D, seed=1)
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("MITDVP_KEFF_IDENT", flag)
        eng = TDVPEngine(L, integrator=integ, conserve_norm=cn)
        eng.set_mpo(mpo)
        eng.set_mps(mps)
        eng.counters_reset()
        eng.propagate(0.3)
        c = eng.counters()
        out[flag] = (eng.get_mps(), eng.krylov_stats(), c["n_launch"], c["n_keff"])
        eng.close()
    a, b = out["1"], out["0"]
    assert a[1] == b[1] and a[3] == b[3]
    nrm = np.sqrt(abs(orc.overlap(a[0], a[0])) * abs(orc.overlap(b[0], b[0])))
    assert abs(abs(orc.overlap(a[0], b[0])) / nrm - 1) < 1e-12
    # (Liouvillian chain: the bond of 256 = 4^4 is of full rank with singular values down to ~1e-6, the rows of the one
    # right-canonical core beside it that belong to them move by rounding / 1e-6; every other core agrees to 1e-13)
    assert max(np.abs(x - y).max() for x, y in zip(a[0], b[0])) < (1e-8 if kind == "liouville" else 1e-11)
    if kind == "dense":
        assert 0 <= a[2] - b[2] <= 16 * (L - 1)  # only the checking launches per bond exponential, nothing else changes
    else:
        assert a[2] != b[2]  # the compact form ran (different launch sequence)
    if kind != "dense":  # ... and against the oracle
        ref = orc.OracleMPS([c.copy() for c in mps], mpo, integrator=integ, conserve_norm=cn)
        ref.propagate(0.3)
        nr = np.sqrt(abs(orc.overlap(ref.cores, ref.cores)) * abs(orc.overlap(a[0], a[0])))
        assert abs(abs(orc.overlap(ref.cores, a[0])) / nr - 1) < 1e-10
        assert abs(abs(orc.overlap(ref.cores, ref.cores)) - abs(orc.overlap(a[0], a[0]))) < 1e-10


def test_ranged_fold_block_sizes_its_output_from_the_mpo_bond_at_the_end_of_the_range():
    """mitdvp_fold_block_range through the Python handle without out_shape: the block that comes back has the MPO
    bond of the LAST core of the range (not of the chain's end cores), in both directions; values equal the oracle's
    environment updates (contract_with_site_mpo, _contraction.py:148-397)."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    L, d, D = 5, 3, 6
    rng = np.random.default_rng(4)
    bonds = [1, 2, 4, 3, 2, 1]
    mpo = [crandn(rng, bonds[p], d, d, bonds[p + 1]) for p in range(L)]
    mps = orc.synthetic_mps([d] * L, D, seed=2)
    eng = TDVPEngine(L)
    eng.set_mpo(mpo)
    eng.set_mps(mps)
    dl = mps[1].shape[0]
    blk = crandn(rng, dl, bonds[1], dl)
    got = eng.fold_block(blk, op_id=0, from_left=True, first=1, count=2)
    want = orc.env_update_left(orc.env_update_left(blk, mps[1], mpo[1]), mps[2], mpo[2])
    assert got.shape == want.shape == (mps[2].shape[2], bonds[3], mps[2].shape[2])
    assert np.abs(got - want).max() < 1e-12 * np.abs(want).max()
    dr = mps[3].shape[2]
    blk = crandn(rng, dr, bonds[4], dr)
    got = eng.fold_block(blk, op_id=0, from_left=False, first=2, count=2)
    want = orc.env_update_right(orc.env_update_right(blk, mps[3], mpo[3]), mps[2], mpo[2])
    assert got.shape == want.shape == (mps[2].shape[0], bonds[2], mps[2].shape[0])
    assert np.abs(got - want).max() < 1e-12 * np.abs(want).max()
    eng.close()
