"""The reference's strongest physics pin, ``tests/test_mixedstate.py``: a spin-1 system between two
spin-1/2 baths with Haberkorn loss and Lindblad jumps, five ways of propagating it, each compared with the dense
``expm`` solution of the vectorised Liouvillian (``:104-236``) at the reference's tolerances (``atol=1e-12`` where the
method is exact at full bond dimension; ``1e-2 / scale**2`` where the dissipator is split off as a gate or a Kraus map).

The exact solution is recomputed here (tests/helpers/spin_bath.py), the matrix-product operators are written out as
sums of products (PyMPO is absent from the image) and checked against the dense operators.  CPU: the NumPy oracle
against the exact solution.  GPU: the HIP path through the PyTDSCF-shaped shell, exactly as the reference's test drives
its Simulator, against the exact solution and the oracle."""

import numpy as np
import pytest

from helpers import spin_bath as sb

# what the reference asserts: (case builder, kwargs, atol at t = 0, atol at the last step)
CASES = {
    "trajectories": (sb.case_trajectories, {}, 1e-12, 1e-12),                       # :239-318
    "liouville": (sb.case_liouville, {"supergate": False}, 1e-12, 1e-12),           # :321-457, supergate=False
    "liouville_supergate": (sb.case_liouville, {"supergate": True, "scale": 2}, 1e-12, 1e-2 / 4),  # numpy leg: scale 2
    "purified": (sb.case_purified, {}, 1e-12, 1e-12),                               # :460-558
    "kraus_single_site": (sb.case_kraus_single, {"scale": 2}, 1e-12, 1e-2 / 4),     # :561-683, numpy leg: scale 2
    "kraus_two_site": (sb.case_kraus_two_site, {"scale": 2}, 1e-12, 1e-2 / 4),      # :686-811
}


def _check(rdms, case, atol0, atol1):
    exact = sb.exact_rdms(**case["exact"])
    s = case["scale"]
    last, ex_last = rdms[(sb.NSTEPS - 1) * s], exact[sb.NSTEPS - 1]
    # the reference's assertions, literally (numpy's default rtol = 1e-7 rides on its atol)
    np.testing.assert_allclose(rdms[0], exact[0], atol=atol0)
    np.testing.assert_allclose(last, ex_last, atol=atol1)
    # ... and without the relative allowance: where the method is exact at full bond dimension what is left is the
    # Krylov threshold (1e-9 on successive approximants, ~1e-12 per local exponential, 11 steps): 2e-12 .. 5e-12 measured
    # for the oracle and for the reference-equivalent iteration counts; 1e-11 is the bound asserted.  The split cases
    # carry the O(dt^2) splitting error of the gate / Kraus map: 3.6e-4 at scale 2 against the reference's 2.5e-3.
    err = np.abs(last - ex_last).max()
    assert np.abs(rdms[0] - exact[0]).max() < 1e-15 + atol0 and err < (1e-11 if atol1 <= 1e-12 else 5e-4), err
    return err


def test_operators_written_out_by_hand_equal_the_dense_ones():
    from pytdscf_amd.operators import mpo_to_dense

    H = sb.hamiltonian_dense() - 0.5j * sb.K_HAB * np.eye(12)
    np.testing.assert_allclose(mpo_to_dense(sb.case_trajectories()["mpo"]), H, atol=1e-14)
    e4 = np.eye(4)
    np.testing.assert_allclose(mpo_to_dense(sb.case_purified()["mpo"]), np.kron(np.kron(np.eye(2), H), np.eye(2)), atol=1e-14)
    # Liouville space: site-local vectorisation (ket (x) bra per site) of i L = H (x) 1 - 1 (x) H^T - i k + i D
    n = 12
    Lv = np.kron(H + 0.5j * sb.K_HAB * np.eye(n), np.eye(n)) - np.kron(np.eye(n), (H + 0.5j * sb.K_HAB * np.eye(n)).T) - 1j * sb.K_HAB * np.eye(n * n)
    Lv = Lv + 1j * sb.dissipator([sb.k3(sb.E2, L, sb.E2) for L in sb.JUMPS], np.eye(n))
    got = mpo_to_dense(sb.case_liouville(False)["mpo"])  # index order (k0 b0)(k1 b1)(k2 b2)
    got = got.reshape(2, 2, 3, 3, 2, 2, 2, 2, 3, 3, 2, 2).transpose(0, 2, 4, 1, 3, 5, 6, 8, 10, 7, 9, 11).reshape(n * n, n * n)
    np.testing.assert_allclose(got, Lv, atol=1e-13)
    assert e4.shape == (4, 4)
    # the exact solution's own sanity: trace decays with the Haberkorn rate, the density stays Hermitian
    ex = sb.exact_rdms()
    np.testing.assert_allclose(np.trace(ex, axis1=1, axis2=2), np.exp(-sb.K_HAB * sb.DT * np.arange(sb.NSTEPS)), atol=1e-13)
    np.testing.assert_allclose(ex, ex.conj().transpose(0, 2, 1), atol=1e-14)


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_against_the_exact_solution(name):
    build, kw, a0, a1 = CASES[name]
    case = build(**kw)
    _check(sb.run_oracle(case), case, a0, a1)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_hip_path_against_the_exact_solution(name, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    build, kw, a0, a1 = CASES[name]
    case = build(**kw)
    got = sb.run_shell(case, name)
    err = _check(got, case, a0, a1)
    # ... and every step against the oracle (same splitting error, so this one is tight for all five cases)
    ref = sb.run_oracle(case)
    np.testing.assert_allclose(got, ref, atol=1e-10)
    print(f"{name}: max |rdm - exact| at the last step {err:.2e}")
