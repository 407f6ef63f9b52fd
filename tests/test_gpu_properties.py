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
