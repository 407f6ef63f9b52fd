"""GPU parity of Simulator.operate (variational application of an operator, WFunc.apply_dipole)
through the C ABI against the reference's golden run and the pinned oracle."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_operate_golden(golden):
    from pytdscf_amd import TDVPEngine

    g = golden("operate_chain.npz")
    n = int(g["nsite"])
    mpo = [g[f"mpo{i}"] for i in range(n)]
    init = [g[f"init{i}"] for i in range(n)]
    for ns in (1, 10):
        eng = TDVPEngine(n)
        eng.set_mpo(mpo)
        eng.set_mps(init, canonicalize=True)
        nrm, iters = eng.operate(0, maxstep=ns)
        ref = float(g[f"n{ns}_norm"])
        assert abs(nrm - ref) < 1e-10 * ref and iters == ns
        assert abs(eng.norm() - 1) < 1e-12
        for i, c in enumerate(eng.get_mps()):
            np.testing.assert_allclose(c, g[f"n{ns}_final{i}"], atol=1e-9)
        assert [eng.get_site_shape(i)[3] for i in range(n)] == [0] + [2] * (n - 1)  # Psi B B ...
        eng.propagate(0.1)  # the environments were dropped: a step from the new state works
        assert abs(eng.norm() - 1) < 1e-12
        eng.close()


def test_operate_exact_when_the_bond_dimension_suffices():
    """A product operator keeps the bond dimensions: the fit is exact after one double sweep and
    converges on the second; with a shift (coupleJ) the identity part is added."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    L, d, D = 5, 3, 4
    rng = np.random.default_rng(3)
    ops = [rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d)) for _ in range(L)]
    mpo = [o[None, :, :, None] for o in ops]
    init = [rng.standard_normal((a, d, b)) + 1j * rng.standard_normal((a, d, b)) for a, b in orc.bond_dims([d] * L, D)]
    cores = orc.canonicalize_site0(init)
    target = [np.einsum("ij,ajb->aib", o, c) for o, c in zip(ops, cores)]
    tnorm = np.sqrt(abs(orc.overlap(target, target)))
    eng = TDVPEngine(L)
    eng.set_mpo(mpo)
    eng.set_mps(init, canonicalize=True)
    nrm, iters = eng.operate(0, maxstep=10)
    assert iters == 2 and abs(nrm - tnorm) < 1e-10 * tnorm
    got = eng.get_mps()
    assert abs(abs(orc.overlap(got, target)) / tnorm - 1) < 1e-10
    eng.close()
    # against the oracle with a shift
    nrm_o, bra_o, it_o = orc.operate(cores, mpo, maxstep=3, shift=0.4 - 0.2j)
    eng = TDVPEngine(L)
    eng.set_mpo(mpo, shift=0.4 - 0.2j)
    eng.set_mps(init, canonicalize=True)
    nrm, iters = eng.operate(0, maxstep=3)
    assert iters == it_o and abs(nrm - nrm_o) < 1e-10 * nrm_o
    for a, b in zip(eng.get_mps(), bra_o):
        np.testing.assert_allclose(a, b, atol=1e-9)
    eng.close()
