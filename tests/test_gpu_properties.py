"""GPU: property-based tests (hypothesis) of the kernels through the C ABI: random shapes and
operand forms of the MFMA complex GEMM (both complex-product modes), random site shapes of the
apply / environment-update / gauge-move building blocks against NumPy / the oracle."""

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

pytestmark = pytest.mark.gpu
SET = dict(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck))


def _crandn(rng, *shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


@settings(**SET)
@given(st.integers(1, 150), st.integers(1, 150), st.integers(0, 260), st.booleans(), st.booleans(), st.booleans(), st.booleans(),
       st.sampled_from([-1, 0, 1, 2]), st.booleans(), st.integers(0, 2**31 - 1))
def test_zgemm_random_shapes(m, n, k, ta, ca, tb, cb, cfg, with_beta, seed):
    from pytdscf_amd import engine as E

    if k == 0:
        k = 1
    rng = np.random.default_rng(seed)
    A = _crandn(rng, k, m) if ta else _crandn(rng, m, k)
    B = _crandn(rng, n, k) if tb else _crandn(rng, k, n)
    C0 = _crandn(rng, m, n)
    alpha, beta = 0.7 - 0.3j, (0.2 + 0.5j if with_beta else 0.0)
    opA = (A.T if ta else A)
    opA = opA.conj() if ca else opA
    opB = (B.T if tb else B)
    opB = opB.conj() if cb else opB
    ref = alpha * (opA @ opB) + beta * C0
    for mode in ("4m", "3m"):
        E.set_gemm_mode(mode)
        try:
            out = E.zgemm(A, B, C0, transA=ta, conjA=ca, transB=tb, conjB=cb, alpha=alpha, beta=beta, tile_cfg=cfg)
        finally:
            E.set_gemm_mode("3m")
        np.testing.assert_allclose(out, ref, atol=1e-11 * max(1.0, np.sqrt(k)))


@settings(**SET)
@given(st.integers(1, 9), st.integers(1, 5), st.integers(1, 9), st.integers(1, 4), st.integers(1, 4), st.integers(0, 2**31 - 1))
def test_building_blocks_random_shapes(dl, d, dr, ml, mr, seed):
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import engine as E

    rng = np.random.default_rng(seed)
    L, R = _crandn(rng, dl, ml, dl), _crandn(rng, dr, mr, dr)
    W, psi = _crandn(rng, ml, d, d, mr), _crandn(rng, dl, d, dr)
    np.testing.assert_allclose(E.heff_apply(L, W, R, psi), orc.heff_apply(L, W, R, psi), atol=1e-11)
    np.testing.assert_allclose(E.env_update(L, psi, W, left=True), orc.env_update_left(L, psi, W), atol=1e-11)
    np.testing.assert_allclose(E.env_update(R, psi, W, left=False), orc.env_update_right(R, psi, W), atol=1e-11)
    sig = _crandn(rng, dl, dr)
    Lk, Rk = _crandn(rng, dl, ml, dl), _crandn(rng, dr, ml, dr)
    np.testing.assert_allclose(E.keff_apply(Lk, Rk, sig), orc.keff_apply(Lk, Rk, sig), atol=1e-11)
    if dl * d >= dr:  # Psi2Asigma needs at least as many rows as columns
        A, s = E.gauge_trf(psi, "Psi2Asigma")
        Ao, so = orc.qr_psi2Asigma(psi)
        np.testing.assert_allclose(A, Ao, atol=1e-10)
        np.testing.assert_allclose(s, so, atol=1e-10)
    if d * dr >= dl:
        B, s = E.gauge_trf(psi, "Psi2sigmaB")
        so, Bo = orc.qr_psi2sigmaB(psi)
        np.testing.assert_allclose(B, Bo, atol=1e-10)
        np.testing.assert_allclose(s, so, atol=1e-10)


@settings(max_examples=20, deadline=None, suppress_health_check=list(HealthCheck))
@given(st.integers(2, 5), st.integers(2, 3), st.integers(1, 3), st.integers(2, 4), st.integers(1, 3), st.sampled_from(["lanczos", "arnoldi"]),
       st.sampled_from([0.0, 0.25]), st.sampled_from([0.3, 1.0]), st.integers(0, 2**31 - 1))
def test_random_adaptive_chains_against_oracle(L, d, D0, M, dD, integ, shift, dt, seed):
    """Adaptive bond dimension on random small chains: same rank decisions, Krylov counts and
    observables as the oracle (loose p_proj so that the ranks do grow)."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    rng = np.random.default_rng(seed)
    mpo = orc.synthetic_mpo(L, d, max(M, 3), seed=seed % 997)
    init = [_crandn(rng, a, d, b) for a, b in orc.bond_dims([d] * L, D0)]
    kw = dict(Dmax=D0 + 3, dD=dD, p_proj=1e-9)
    cn = integ == "lanczos"
    st_ = orc.OracleMPS(orc.canonicalize_site0(init), mpo, integrator=integ, conserve_norm=cn, shift=shift, adaptive=True, **kw)
    eng = TDVPEngine(L, integrator=integ, conserve_norm=cn)
    eng.set_mpo(mpo, shift=shift)
    eng.set_mps(init, canonicalize=True)
    eng.set_adaptive(True, **kw)
    for _ in range(2):
        st_.propagate(dt)
        eng.propagate(dt)
        assert eng.bond_dims() == [c.shape[2] for c in st_.cores[:-1]]
    assert eng.krylov_stats() == [st_.kprev[i] for i in range(L)]
    assert abs(eng.norm() - st_.norm()) < 1e-9 * st_.norm()
    e_o, e_e = st_.expectation(), eng.expectation()
    assert abs(e_o - e_e) < 1e-7 * max(abs(e_o), 1e-10)
    eng.close()


@settings(max_examples=30, deadline=None, suppress_health_check=list(HealthCheck))
@given(st.integers(1, 5), st.integers(1, 3), st.integers(1, 6), st.integers(1, 3), st.sampled_from(["lanczos", "arnoldi"]),
       st.booleans(), st.sampled_from([0.0, 0.3, -0.2 + 0.1j]), st.sampled_from([0.05, 0.4, 1.5]), st.integers(0, 2**31 - 1))
def test_random_small_chains_against_oracle(L, d, D, M, integ, cn, shift, dt, seed):
    """Whole time steps on random small chains (including one-site chains, d = 1 sites, D = 1
    bonds, M = 1 operators) in every integrator / normalisation / scalar-term combination."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    if integ == "lanczos" and isinstance(shift, complex):
        shift = shift.real  # Lanczos is for Hermitian effective Hamiltonians
    rng = np.random.default_rng(seed)
    if integ == "lanczos":  # a Hermitian operator
        mpo = orc.synthetic_mpo(L, d, max(M, 3), seed=seed % 1000) if L > 1 else [
            (lambda w: 0.5 * (w + w.conj().transpose(0, 2, 1, 3)))(0.3 * _crandn(rng, 1, d, d, 1))]
    else:
        mpo = [0.3 * _crandn(rng, 1 if p == 0 else M, d, d, 1 if p == L - 1 else M) for p in range(L)]
    init = [_crandn(rng, a, d, b) for a, b in orc.bond_dims([d] * L, D)]
    st_ = orc.OracleMPS(orc.canonicalize_site0(init), mpo, integrator=integ, conserve_norm=cn, shift=shift)
    eng = TDVPEngine(L, integrator=integ, conserve_norm=cn)
    eng.set_mpo(mpo, shift=shift)
    eng.set_mps(init, canonicalize=True)
    try:
        for _ in range(2):
            st_.propagate(dt)
    except ValueError:  # Krylov space too small for this dt: the engine must refuse as well
        with pytest.raises(ValueError):
            for _ in range(2):
                eng.propagate(dt)
        eng.close()
        return
    for _ in range(2):
        eng.propagate(dt)
    assert eng.krylov_stats() == [st_.kprev[i] for i in range(L)]
    assert abs(eng.norm() - st_.norm()) < 1e-10 * max(st_.norm(), 1e-300)
    e_o, e_e = st_.expectation(), eng.expectation()
    assert abs(e_o - e_e) < 1e-8 * max(abs(e_o), 1e-12)
    a_o, a_e = st_.autocorr(), eng.autocorr()
    assert abs(a_o - a_e) < 1e-8 * max(abs(a_o), 1e-12)
    eng.close()


@settings(max_examples=25, deadline=None, suppress_health_check=list(HealthCheck))
@given(st.integers(1, 4), st.integers(2, 3), st.integers(1, 3), st.lists(st.integers(1, 5), min_size=3, max_size=3),
       st.sampled_from(["lanczos", "arnoldi"]), st.booleans(), st.booleans(), st.sampled_from([0.05, 0.3]),
       st.integers(0, 2**31 - 1))
def test_random_multistate_chains_against_oracle(L, d, S, Ds, integ, cn, couple, dt, seed):
    """Whole time steps with several electronic states: random numbers of states, bond dimensions
    per state, present / absent off-diagonal blocks and scalar terms, both integrators."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import MultiStateEngine

    rng = np.random.default_rng(seed)
    herm = integ == "lanczos"
    mpo = [[None] * S for _ in range(S)]
    cj = [[0.0] * S for _ in range(S)]
    bonds = lambda M: list(zip([1] + [M] * (L - 1), [M] * (L - 1) + [1]))  # noqa: E731
    adj = lambda ws: [np.ascontiguousarray(np.conj(w.transpose(0, 2, 1, 3))) for w in ws]  # noqa: E731
    for i in range(S):
        if L > 1 and herm:
            mpo[i][i] = orc.synthetic_mpo(L, d, 3, seed=(seed + i) % 1000)
        else:
            w = [0.3 * _crandn(rng, a, d, d, b) for a, b in bonds(2)]
            if herm:  # one site: a Hermitian matrix
                w = [0.5 * (w[0] + w[0].conj().transpose(0, 2, 1, 3))]
            mpo[i][i] = w
        cj[i][i] = float(rng.normal()) * 0.1 if couple else 0.0
        for j in range(i + 1, S):
            if rng.random() < 0.6:
                w = [0.2 * _crandn(rng, a, d, d, b) for a, b in bonds(2)]
                mpo[i][j] = w
                mpo[j][i] = adj(w) if herm else [0.2 * _crandn(rng, a, d, d, b) for a, b in bonds(2)]
            if couple and rng.random() < 0.6:
                c = complex(rng.normal(), rng.normal()) * 0.1
                cj[i][j], cj[j][i] = c, (c.conjugate() if herm else complex(rng.normal(), 0.0) * 0.1)
    raw = [[_crandn(rng, a, d, b) for a, b in orc.bond_dims([d] * L, Ds[s])] for s in range(S)]
    weights = rng.random(S) + 0.05
    w = weights / weights.sum()
    init = [orc.canonicalize_site0(raw[s], float(np.sqrt(w[s]))) for s in range(S)]
    st_ = orc.OracleMultiMPS(init, mpo, cj, integrator=integ, conserve_norm=cn)
    eng = MultiStateEngine(L, S, integrator=integ, conserve_norm=cn)
    eng.set_hamiltonian(mpo, cj)
    eng.set_states(raw, weights=list(weights))
    try:
        for _ in range(2):
            st_.propagate(dt)
    except ValueError:
        with pytest.raises(ValueError):
            for _ in range(2):
                eng.propagate(dt)
        eng.close()
        return
    for _ in range(2):
        eng.propagate(dt)
    assert eng.krylov_stats() == [st_.kprev[i] for i in range(L)]
    np.testing.assert_allclose(eng.pop_states(), st_.pop_states(), rtol=1e-9, atol=1e-12)
    e_o, e_e = st_.expectation(), eng.expectation()
    assert abs(e_o - e_e) < 1e-8 * max(abs(e_o), 1e-10)
    a_o, a_e = st_.autocorr(), eng.autocorr()
    assert abs(a_o - a_e) < 1e-8 * max(abs(a_o), 1e-10)
    eng.close()
