"""GPU: the 3M (Karatsuba) complex product of csrc/zgemm.hip against the 4M product, COMPONENT by component.

3M forms Im = P3 - P1 - P2 by cancellation: it is normwise, not componentwise, stable, while the reference multiplies with
zgemm's 4M product (tensordot, _contraction.py:1165-1173).  What a TDVP run can notice is a small imaginary part beside a
large real one: the t/2-trick autocorrelation <psi(t/2)*|psi(t/2)> of a nearly real state (wavefunction.py:226-257).  In
complex128 the absolute error of either product is ~1e-16 |A||B| per term, so a component 1e4 times smaller than the norm
still has 1e-12 of its own magnitude -- these tests pin that margin against the 1e-8 bar of north_star.

  * the reference's chain_lanczos fixture driven through the MFMA kernels (small-bond kernels off, which use plain 4-term
    products and no MFMA) in both modes: real and imaginary part of the autocorrelation each to 1e-8 of ITS OWN magnitude
    against the reference's numbers, equal Krylov counts;
  * a nearly real state (real start, real-symmetric MPO, small time step: Im <psi*|psi> ~ 1e-4 Re): 3M and 4M against the
    oracle, again per component, plus the imaginary parts of the propagated tensors 3M vs 4M.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _comp(got, want, tol):
    """each component to tol of its own magnitude"""
    assert abs(got.real - want.real) < tol * abs(want.real), (got, want)
    assert abs(got.imag - want.imag) < tol * abs(want.imag), (got, want)


@pytest.mark.parametrize("mode", ["3m", "4m"])
def test_reference_chain_through_the_mfma_kernels_per_component(golden, mode):
    from pytdscf_amd import TDVPEngine
    from pytdscf_amd import engine as E

    g = golden("chain_lanczos.npz")
    n = int(g["nsite"])
    mpo = [g[f"mpo{i}"] for i in range(n)]
    init = [g[f"init{i}"] for i in range(n)]
    dt = float(g["dt_au"])
    E.set_gemm_mode(mode)
    try:
        for ns in (1, 4):
            eng = TDVPEngine(n)
            eng.set_small_kernels(False)  # every contraction through zgemm_kernel
            eng.set_mpo(mpo)
            eng.set_mps(init, canonicalize=True)
            for _ in range(ns):
                eng.propagate(dt)
            assert eng.krylov_stats() == list(g[f"n{ns}_krylov"])
            _comp(eng.autocorr(), complex(g[f"n{ns}_autocorr"]), 1e-8)
            ef = float(g[f"n{ns}_energy_final"].real)
            assert abs(eng.expectation().real - ef) < 1e-8 * abs(ef)
            assert abs(eng.norm() - 1) < 1e-12
            eng.close()
    finally:
        E.set_gemm_mode("3m")


def test_nearly_real_state_small_imaginary_parts():
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine
    from pytdscf_amd import engine as E

    L, d, D, M, dt = 6, 8, 48, 6, 0.01
    mpo = [np.ascontiguousarray(w.real.astype(np.complex128)) for w in orc.synthetic_mpo(L, d, M, seed=3)]  # real symmetric
    rng = np.random.default_rng(5)
    mps = orc.canonicalize_site0([rng.standard_normal((a, d, b)).astype(np.complex128) / np.sqrt(a * d)
                                  for a, b in orc.bond_dims([d] * L, D)])  # real tensors, site-0-centred, normalised
    assert max(np.abs(c.imag).max() for c in mps) == 0.0
    ref = orc.OracleMPS([c.copy() for c in mps], mpo)
    ref.propagate(dt)
    a_ref, e_ref = ref.autocorr(), ref.expectation()
    assert abs(a_ref.imag) < 1e-2 * abs(a_ref.real) and a_ref.imag != 0.0  # the case this test is about
    res = {}
    try:
        for mode in ("3m", "4m"):
            E.set_gemm_mode(mode)
            eng = TDVPEngine(L)
            eng.set_small_kernels(False)
            eng.set_mpo(mpo)
            eng.set_mps(mps)
            eng.propagate(dt)
            assert eng.krylov_stats() == [ref.kprev[i] for i in range(L)]
            a = eng.autocorr()
            _comp(a, a_ref, 1e-8)
            assert abs(eng.expectation().real - e_ref.real) < 1e-8 * abs(e_ref.real)
            assert abs(eng.norm() - 1) < 1e-12
            res[mode] = (a, eng.get_mps())
            eng.close()
    finally:
        E.set_gemm_mode("3m")
    # the propagated tensors themselves: imaginary parts are O(dt) of the real ones; 3M against 4M per component
    for c3, c4 in zip(res["3m"][1], res["4m"][1]):
        im = np.abs(c4.imag).max()
        assert im > 0 and im < 0.1 * np.abs(c4.real).max()
        assert np.abs(c3.imag - c4.imag).max() < 1e-8 * im
        assert np.abs(c3.real - c4.real).max() < 1e-8 * np.abs(c4.real).max()
    _comp(res["3m"][0], res["4m"][0], 1e-10)
