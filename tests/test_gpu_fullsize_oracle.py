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


def test_c2_exact_shape_three_steps_against_oracle():
    """BASELINE configs[1] at its exact shape and time step as bench.py runs it (L=10, d=10, D=32, M=6, dt=2.0 a.u.): every
    local exponential is ONE launch of k_small_site (csrc/small_site.hip) -- three time steps against OracleMPS
    (short_iterative_lanczos semantics, _integrator.py:453-655, incl. the warm-up memory :178-186): equal Krylov counts
    after every step, energy / autocorrelation to 1e-8 relative, fidelity to 1e-10, norm to 1e-12."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    L, d, D, M, dt = 10, 10, 32, 6, 2.0
    mpo = orc.synthetic_mpo(L, d, M, seed=0)
    mps = orc.synthetic_mps([d] * L, D, seed=1)
    eng = TDVPEngine(L)
    eng.set_mpo(mpo)
    eng.set_mps(mps)
    ref = orc.OracleMPS([c.copy() for c in mps], mpo)
    eng.counters_reset()
    for step in range(3):
        eng.propagate(dt)
        ref.propagate(dt)
        assert eng.krylov_stats() == [ref.kprev[i] for i in range(L)], step
        eg, er = eng.expectation(), ref.expectation()
        ag, ar = eng.autocorr(), ref.autocorr()
        assert abs(eg - er) < 1e-8 * abs(er) and abs(ag - ar) < 1e-8 * abs(ar), step
        assert abs(eng.norm() - 1) < 1e-12
        assert abs(abs(orc.overlap(ref.cores, eng.get_mps())) - 1) < 1e-10, step
    # the one-launch family really ran: 6 sweeps of 10 site + 9 bond exponentials, no multi-launch apply in between
    cnt = eng.counters()
    assert cnt["n_exp_site"] == 6 * L and cnt["n_exp_bond"] == 6 * (L - 1)
    assert cnt["n_host_waits"] <= 6 * 4, cnt["n_host_waits"]  # norm / energy reads only: no wait inside a local exponential
    eng.close()


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


def test_c4_interior_apply_and_environment_update_through_the_sweeps_own_kernels(monkeypatch):
    """The kernels bench.py times at D = 1024, against the oracle over the whole output (_contraction.py:1182-1243;
    identity shortcuts _mps_mpo.py:510-523): the edge form (two products with the reducing 4 x 4 x 4 epilogue, cores in
    fragment order: what the size rule selects at C4 since round 5), and the three-stage chain it replaced -- block-sparse W
    stage (list kernel + row map) selected by the finite-state-machine core of the bench's generator, gathered-row first
    stage and copy + accumulate third stage selected by the identity blocks of CANONICAL environments.  A 7-site chain
    reaches the C4 interior shape (1024 x 16 x 1024) at its middle site; mitdvp_heff_apply_center issues the apply as a
    local exponential does and reports which variants ran."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine
    from pytdscf_amd import synthetic as syn

    L, d, D, M = 7, 16, 1024, 32
    mpo = syn.synthetic_mpo(L, d, M, seed=0)
    eng = TDVPEngine(L)
    eng.set_mpo(mpo)
    eng.init_random([d] * L, D, seed=1)
    c = 3
    assert eng.get_site_shape(c)[:3] == (D, d, D)
    eng.build_envs(1)
    for _ in range(c):  # centre to the middle site: left-canonical sites and their blocks behind it
        eng.split_center(True)
        eng.absorb_bond(True)
    got, flags = eng.heff_apply_center()
    assert flags & 0x10, flags  # the edge form: the bench's configuration
    Lb, Rb, psi = eng.get_env(0, c), eng.get_env(1, c + 1), eng.get_site(c)
    assert np.abs(Lb[:, 0, :] - np.eye(D)).max() < 1e-12 and np.abs(Rb[:, M - 1, :] - np.eye(D)).max() < 1e-12
    ch = max(1, min(D, int(6.4e7 // (M * d * D))))
    want = orc.heff_apply_chunked(Lb, mpo[c], Rb, psi, ch)
    assert _rel(got, want) < 1e-12
    del want
    # the three-stage chain on the same operands (a second engine: the form is chosen at construction)
    monkeypatch.setenv("MITDVP_EDGE_APPLY", "0")
    eng2 = TDVPEngine(L)
    monkeypatch.delenv("MITDVP_EDGE_APPLY")
    eng2.set_mpo(mpo)
    eng2.init_random([d] * L, D, seed=1)
    eng2.build_envs(1)
    for _ in range(c):
        eng2.split_center(True)
        eng2.absorb_bond(True)
    assert np.array_equal(eng2.get_site(c), psi)  # same state, same blocks
    got2, flags2 = eng2.heff_apply_center()
    assert flags2 & 7 == 7 and not flags2 & 0x10, flags2  # S1 trimmed, S3 trimmed, block-sparse W stage
    assert _rel(got2, got) < 1e-12
    eng2.close()
    del got, got2
    # a second, random input vector through the same operands (nothing about the apply may depend on x being the state)
    x = _crandn(np.random.default_rng(5), D, d, D)
    got, _ = eng.heff_apply_center(x)
    assert _rel(got, orc.heff_apply_chunked(Lb, mpo[c], Rb, x, ch)) < 1e-12
    del got, x
    # the environment update of the same site inside split_center (its W stage is block-sparse too), three bands of the
    # new block against the oracle as above, plus the identity block the next site's apply will rely on
    eng.split_center(True)
    A, gl = eng.get_site(c), eng.get_env(0, c + 1)
    scale = np.abs(gl).max()
    for j0 in (0, D // 2 - 32, D - 64):
        wl = _env_left_band(orc, Lb, A, mpo[c], j0, j0 + 64)
        assert np.abs(gl[:, :, j0 : j0 + 64] - wl).max() < 1e-12 * scale
    assert np.abs(gl[:, 0, :] - np.eye(D)).max() < 1e-12
    eng.close()


def test_c5_real_workload_one_time_step():
    """BASELINE configs[4] as bench.py runs it: the dissipative, non-Hermitian generator synthetic_liouvillian_mpo(L=128,
    M=16, gamma=0.002), D = 512, Arnoldi, conserve_norm off, the full chain.  One whole time step in both complex-product
    modes (3M / 4M: different arithmetic, different tiles) must agree; a step is undone by the step with -dt (the
    splitting is symmetric whatever the generator); one interior site exponential (512 x 4 x 512, M = 16) equals the
    oracle's short-iterative Arnoldi (_integrator.py:287-432) with the same Krylov count."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine
    from pytdscf_amd import engine as E
    from pytdscf_amd import synthetic as syn

    L, D, dt = 128, 512, 0.5
    mpo = syn.synthetic_liouvillian_mpo(L, 16, seed=0, gamma=0.002)
    res = {}
    for mode in ("3m", "4m"):
        E.set_gemm_mode(mode)
        try:
            eng = TDVPEngine(L, integrator="arnoldi", conserve_norm=False)
            eng.set_mpo(mpo)
            eng.init_random([4] * L, D, seed=1)
            assert max(eng.bond_dims()) == D
            a0, n0 = eng.autocorr(), eng.norm()
            eng.propagate(dt)
            res[mode] = (eng.norm(), eng.autocorr(), eng.expectation(), eng.krylov_stats())
            if mode == "3m":
                assert abs(res[mode][0] - n0) > 1e-6  # the generator is dissipative: the norm is NOT conserved
                eng.propagate(-dt)
                assert abs(eng.norm() - n0) < 1e-6 and abs(eng.autocorr() - a0) < 1e-6 * abs(a0)
                # one interior site exponential against the oracle's Arnoldi
                eng.build_envs(1)
                c = L // 2
                for _ in range(c):
                    eng.split_center(True)
                    eng.absorb_bond(True)
                assert eng.get_site_shape(c)[:3] == (D, 4, D)
                Lb, Rb, psi = eng.get_env(0, c), eng.get_env(1, c + 1), eng.get_site(c)
                k0 = eng.krylov_memory(c)
                eng.site_exp(dt)
                want, k = orc.sil_arnoldi(-0.5j * dt, lambda v: orc.heff_apply(Lb, mpo[c], Rb, v), psi, 1e-9, k0, False, None)
                assert eng.krylov_memory(c) == k
                assert _rel(eng.get_site(c), want) < 1e-9
            eng.close()
        finally:
            E.set_gemm_mode("3m")
    n3, a3, e3, k3 = res["3m"]
    n4, a4, e4, k4 = res["4m"]
    assert k3 == k4
    assert abs(n3 - n4) < 1e-10 and abs(a3 - a4) < 1e-9 * abs(a3) and abs(e3 - e4) < 1e-9 * abs(e3)
