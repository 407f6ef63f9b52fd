"""GPU: oracle parity AT BASELINE.json's full shapes, through the C ABI's fine seam
(mitdvp_heff_apply / _env_update / _keff_apply / _gauge_trf) and through the sweep.

  * C4 interior site (D=1024, d=16, M=32): one H_eff apply compared with the oracle over the whole
    output; the left and right environment updates over column bands of the output (the oracle's cost is
    proportional to the band, a full block would take 20 s of 64 host threads each); K_eff apply and the
    QR gauge move in full.  Reference semantics: _contraction.py:148-397, :1182-1243, :1339-1352,
    _site_cls.py:138-292.
  * C5 interior site (D=512, d=4, M=16): the same four in full, plus a D=512 sweep of the non-Hermitian
    generator at reduced length with the size-independent checks.
  * C3 at full size (L=6, d=32, D=128, M=16): one whole time step (2 sweeps) against OracleMPS -- energy /
    autocorrelation to 1e-8, fidelity to 1e-10, equal Krylov counts.
  * a short chain with C4's interior shape (L=8, d=16, D=1024: the two middle sites are 1024 x 16 x 1024)
    through sweep(): exercises the full-size QR, the conjugated-operand environment GEMM and the split-K
    paths inside the real orchestration; checked by norm, energy conservation and reversibility.

Tolerances: relative 1e-12 in the max norm for single contractions (c128, K up to 32768 terms),
1e-11 for Q / R of the 16384 x 1024 QR (LAPACK-vs-device Householder ordering).
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _crandn(rng, *s):
    a = rng.standard_normal(s + (2,))
    return a.view(np.complex128).reshape(s)


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def _env_left_band(orc, L, A, W, j0, j1):
    """oracle env_update_left restricted to ket columns [j0, j1) of the new block."""
    return orc.env_update_left(L, A[:, :, j0:j1], W, bra=A)


def _check_interior(orc, E, D, d, M, full_env: bool):
    rng = np.random.default_rng(11)
    Lb, Rb = _crandn(rng, D, M, D), _crandn(rng, D, M, D)
    W = _crandn(rng, M, d, d, M)
    psi = _crandn(rng, D, d, D)
    psi /= np.linalg.norm(psi)

    # a4: H_eff apply, whole output
    got = E.heff_apply(Lb, W, Rb, psi)
    ch = max(1, min(D, int(6.4e7 // (M * d * D))))
    want = orc.heff_apply_chunked(Lb, W, Rb, psi, ch)
    assert _rel(got, want) < 1e-12
    del got, want

    # a5: K_eff apply
    sv = _crandn(rng, D, D)
    assert _rel(E.keff_apply(Lb, Rb, sv), orc.keff_apply(Lb, Rb, sv)) < 1e-12

    # a8: QR gauge moves (full-rank input: Q and R are unique given LAPACK's sign convention)
    A, sig = E.gauge_trf(psi, "Psi2Asigma")
    Ao, so = orc.qr_psi2Asigma(psi)
    assert _rel(sig, so) < 1e-11 and _rel(A, Ao) < 1e-11
    Am = A.reshape(D * d, D)
    assert np.abs(Am.conj().T @ Am - np.eye(D)).max() < 5e-14
    B, sigb = E.gauge_trf(psi, "Psi2sigmaB")
    sbo, Bo = orc.qr_psi2sigmaB(psi)
    Bo = np.ascontiguousarray(Bo)
    assert _rel(sigb, sbo) < 1e-11 and _rel(B, Bo) < 1e-11

    # a3: environment updates with the isometries just produced (left with A, right with B)
    gl = E.env_update(Lb, A, W, left=True)
    gr = E.env_update(Rb, B, W, left=False)
    if full_env:
        assert _rel(gl, orc.env_update_left(Lb, A, W)) < 1e-12
        assert _rel(gr, orc.env_update_right(Rb, B, W)) < 1e-12
    else:
        scale_l, scale_r = np.abs(gl).max(), np.abs(gr).max()
        for j0 in (0, D // 2 - 32, D - 64):  # first, middle (unaligned to the tile grid) and last band
            wl = _env_left_band(orc, Lb, A, W, j0, j0 + 64)
            assert np.abs(gl[:, :, j0 : j0 + 64] - wl).max() < 1e-12 * scale_l
            wr = orc.env_update_right(Rb, B[j0 : j0 + 64], W, bra=B)
            assert np.abs(gr[:, :, j0 : j0 + 64] - wr).max() < 1e-12 * scale_r


def test_c4_interior_site_against_oracle():
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import engine as E

    _check_interior(orc, E, 1024, 16, 32, full_env=False)


def test_c5_interior_site_against_oracle():
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import engine as E

    _check_interior(orc, E, 512, 4, 16, full_env=True)


def test_c3_full_size_step_against_oracle():
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    L, d, D, M, dt = 6, 32, 128, 16, 1.0
    mpo = orc.synthetic_mpo(L, d, M, seed=0)
    mps = orc.synthetic_mps([d] * L, D, seed=1)
    eng = TDVPEngine(L)
    eng.set_mpo(mpo)
    eng.set_mps(mps)
    ref = orc.OracleMPS([c.copy() for c in mps], mpo)
    e0g, e0r = eng.expectation(), ref.expectation()
    assert abs(e0g - e0r) < 1e-10 * abs(e0r)
    eng.propagate(dt)
    ref.propagate(dt)
    assert eng.krylov_stats() == [ref.kprev[i] for i in range(L)]
    eg, er = eng.expectation(), ref.expectation()
    ag, ar = eng.autocorr(), ref.autocorr()
    assert abs(eg - er) < 1e-8 * abs(er) and abs(ag - ar) < 1e-8 * abs(ar)
    assert abs(eng.norm() - 1) < 1e-12
    fid = abs(orc.overlap(ref.cores, eng.get_mps()))
    assert abs(fid - 1) < 1e-10
    eng.close()


def test_c4_shape_chain_through_sweep():
    from pytdscf_amd import TDVPEngine
    from pytdscf_amd import synthetic as syn

    L, d, D, M, dt = 8, 16, 1024, 32, 0.5
    eng = TDVPEngine(L)
    eng.set_mpo(syn.synthetic_mpo(L, d, M, seed=0))
    eng.init_random([d] * L, D, seed=1)
    assert eng.get_site_shape(3)[:3] == (D, d, D) and eng.get_site_shape(4)[:3] == (D, d, D)
    e0, a0 = eng.expectation(), eng.autocorr()
    assert abs(e0.imag) < 1e-12 * max(1.0, abs(e0))
    eng.propagate(dt)
    assert abs(eng.norm() - 1) < 1e-12
    assert abs(eng.expectation() - e0) < 1e-8 * max(1.0, abs(e0))
    eng.propagate(-dt)
    assert abs(eng.norm() - 1) < 1e-12
    assert abs(eng.autocorr() - a0) < 1e-7 * abs(a0)
    assert abs(eng.expectation() - e0) < 1e-8 * max(1.0, abs(e0))
    assert max(eng.krylov_stats()) <= 20
    eng.close()


def test_c5_shape_liouvillian_sweep_properties():
    """D=512 reached at reduced length (L=14, d=4: bonds 4,16,64,256,512,...): with zero damping the
    vectorised von Neumann generator is Hermitian, so the Arnoldi / conserve_norm=False path keeps the
    norm by itself and a step is reversible."""
    from pytdscf_amd import TDVPEngine
    from pytdscf_amd import synthetic as syn

    L, D, dt = 14, 512, 0.5
    eng = TDVPEngine(L, integrator="arnoldi", conserve_norm=False)
    eng.set_mpo(syn.synthetic_liouvillian_mpo(L, 16, seed=0, gamma=0.0))
    eng.init_random([4] * L, D, seed=3)
    assert max(eng.bond_dims()) == D
    a0 = eng.autocorr()
    eng.propagate(dt)
    assert abs(eng.norm() - 1) < 1e-8  # unitary up to thresh_sil accumulation; nothing renormalises
    eng.propagate(-dt)
    assert abs(eng.autocorr() - a0) < 1e-6 * abs(a0)
    eng.close()
