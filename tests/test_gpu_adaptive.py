"""GPU parity of the adaptive bond dimension (a1TDVP) path: SiteCoef.thin_to_full,
the rank selection and the zero-padded propagation, through the C ABI, against the
reference's golden vectors and the pinned NumPy oracle."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _isometry(rng, rows, cols):
    q, _ = np.linalg.qr(rng.standard_normal((rows, cols)) + 1j * rng.standard_normal((rows, cols)))
    return q


@pytest.mark.parametrize(
    "gauge,shape,extra",
    [
        ("A", (3, 4, 5), 2), ("A", (1, 8, 1), 4), ("A", (5, 3, 7), 8), ("A", (2, 2, 4), 3), ("A", (4, 3, 2), 0),
        ("A", (9, 5, 40), 5), ("A", (16, 4, 33), 31),
        ("B", (5, 4, 3), 2), ("B", (1, 8, 1), 4), ("B", (7, 3, 5), 8), ("B", (2, 2, 1), 1), ("B", (2, 3, 4), 0),
        ("B", (40, 5, 9), 5), ("B", (33, 4, 16), 31),
    ],
)
def test_thin_to_full_vs_lapack(gauge, shape, extra):
    """The added orthogonal-complement vectors are LAPACK's (scipy.linalg.qr(mode="full")),
    not merely some orthonormal completion: the rank selection walks them in order."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import engine as E

    rng = np.random.default_rng(hash((gauge, shape, extra)) % 2**32)
    l, c, r = shape
    if gauge == "A":
        core = _isometry(rng, l * c, r).reshape(l, c, r)
    else:
        core = np.ascontiguousarray(_isometry(rng, c * r, l).T).reshape(l, c, r)
    ref = orc.thin_to_full(core, gauge, extra)
    out = E.thin_to_full(core, gauge, extra)
    assert out.shape == ref.shape
    np.testing.assert_allclose(out, ref, atol=1e-12)
    m = out.reshape(-1, out.shape[2]) if gauge == "A" else out.reshape(out.shape[0], -1).T
    np.testing.assert_allclose(m.conj().T @ m, np.eye(m.shape[1]), atol=1e-13)


def test_thin_to_full_bad_arguments():
    from pytdscf_amd import engine as E

    with pytest.raises(ValueError):
        E.thin_to_full(np.zeros((2, 2, 2), complex), "Psi", 1)
    with pytest.raises(ValueError):
        E.thin_to_full(np.zeros((2, 2, 9), complex), "A", 1)  # 4 x 9 cannot be an isometry


def _run_engine(mpo, init, dt, ns, kw, integrator="lanczos", conserve_norm=True, shift=0.0):
    from pytdscf_amd import TDVPEngine

    n = len(mpo)
    eng = TDVPEngine(n, integrator=integrator, conserve_norm=conserve_norm)
    eng.set_mpo(mpo, shift=shift)
    eng.set_mps(init, canonicalize=True)
    eng.set_adaptive(True, **kw)
    e_last = None
    for _ in range(ns):
        e_last = eng.expectation()
        eng.propagate(dt)
    return eng, e_last


@pytest.mark.parametrize("name", ["adaptive_chain.npz", "adaptive_chain_shift.npz"])
def test_adaptive_chain_golden(golden, name):
    g = golden(name)
    n = int(g["nsite"])
    mpo = [g[f"mpo{i}"] for i in range(n)]
    init = [g[f"init{i}"] for i in range(n)]
    dt = float(g["dt_au"])
    kw = dict(Dmax=int(g["Dmax"]), dD=int(g["dD"]), p_proj=float(g["p_proj"]))
    for ns in (1, 3):
        eng, e_last = _run_engine(mpo, init, dt, ns, kw, shift=float(g["coupleJ"]) if "coupleJ" in g.files else 0.0)
        assert eng.bond_dims() == list(g[f"n{ns}_bonddim"])
        assert eng.krylov_stats() == list(g[f"n{ns}_krylov"])
        el = float(g[f"n{ns}_energy_last"])
        assert abs(e_last.real - el) < 1e-8 * abs(el)
        assert abs(eng.norm() - float(g[f"n{ns}_norm"])) < 1e-12
        ac = complex(g[f"n{ns}_autocorr"])
        assert abs(eng.autocorr() - ac) < 1e-8 * abs(ac)
        ef = float(g[f"n{ns}_energy_final"].real)
        assert abs(eng.expectation().real - ef) < 1e-8 * abs(ef)
        for i, c in enumerate(eng.get_mps()):  # same Householder convention => even the tensors agree
            np.testing.assert_allclose(c, g[f"n{ns}_final{i}"], atol=1e-8)
        eng.close()


def test_adaptive_exciton_golden(golden):
    """The reference's tests/test_a1tdvp.py model from the bond-dimension-1 product state.
    Ranks and Krylov counts are exact; observables to 1e-5 only, because two of the five
    directions the bonds grow into carry rounding noise (see tests/test_oracle_golden.py)."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import mps as M
    from pytdscf_amd import operators as O

    g = golden("adaptive_exciton.npz")
    pot = [g[f"pot{i}"] for i in range(4)]
    kin = [g[f"kin{i}"] for i in range(3)]
    mpo = O.merge_operator_terms([(pot, [0, 1, 2, 3]), (kin, [0, 1, 2])], dims=[8, 8, 8, 2])
    w = [g[f"w{i}"] for i in range(3)] + [np.array([0.0, 1.0])]
    init = M.product_state_cores(w, bond_dim=1)
    dt = float(g["dt_au"])
    kw = dict(Dmax=int(g["Dmax"]), dD=int(g["dD"]), p_proj=float(g["p_proj"]))
    for ns in (2, 10):
        eng, e_last = _run_engine(mpo, init, dt, ns, kw)
        assert eng.bond_dims() == list(g[f"n{ns}_bonddim"])
        assert eng.krylov_stats() == list(g[f"n{ns}_krylov"])
        np.testing.assert_allclose(e_last.real, float(g[f"n{ns}_energy_last"]), rtol=1e-5)
        assert abs(eng.norm() - float(g[f"n{ns}_norm"])) < 1e-12
        np.testing.assert_allclose(eng.autocorr(), complex(g[f"n{ns}_autocorr"]), atol=1e-4)
        ref = [g[f"n{ns}_final{i}"] for i in range(4)]
        assert abs(abs(orc.overlap(ref, eng.get_mps())) - 1) < 1e-7
        eng.close()


@pytest.mark.parametrize("integ,cn,shift", [("lanczos", True, 0.3 - 0.0j), ("arnoldi", False, 0.0), ("arnoldi", True, -0.2 + 0.05j)])
def test_adaptive_vs_oracle(integ, cn, shift):
    """Seeded chain against the pinned oracle: coupleJ shift term on the padded vectors,
    Arnoldi, several steps of growth up to the cap."""
    from oracle import tdvp_oracle as orc

    L, d, M, D0 = 7, 3, 5, 2
    mpo = orc.synthetic_mpo(L, d, M, seed=11)
    rng = np.random.default_rng(5)
    init = [rng.standard_normal((a, d, b)) + 1j * rng.standard_normal((a, d, b)) for a, b in orc.bond_dims([d] * L, D0)]
    kw = dict(Dmax=9, dD=2, p_proj=1e-9)
    dt = 0.7
    st = orc.OracleMPS(orc.canonicalize_site0(init), mpo, integrator=integ, conserve_norm=cn, shift=shift, adaptive=True, **kw)
    dims = []
    for _ in range(3):
        st.propagate(dt)
        dims.append([c.shape[2] for c in st.cores[:-1]])
    eng, _ = _run_engine(mpo, init, dt, 3, kw, integrator=integ, conserve_norm=cn, shift=shift)
    assert eng.bond_dims() == dims[-1]
    assert max(dims[-1]) > D0  # the case does grow
    assert eng.krylov_stats() == [st.kprev[i] for i in range(L)]
    assert abs(eng.norm() - st.norm()) < 1e-10
    assert abs(eng.autocorr() - st.autocorr()) < 1e-8 * abs(st.autocorr())
    assert abs(eng.expectation() - st.expectation()) < 1e-8 * abs(st.expectation())
    for a, b in zip(eng.get_mps(), st.cores):
        np.testing.assert_allclose(a, b, atol=1e-8)
    eng.close()


def test_adaptive_relax_rejected():
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    mpo = orc.synthetic_mpo(4, 2, 3, seed=1)
    eng = TDVPEngine(4, relax=True)
    eng.set_mpo(mpo)
    eng.set_mps(orc.synthetic_mps([2] * 4, 2), canonicalize=True)
    eng.set_adaptive(True, Dmax=4, dD=1, p_proj=1e-6)
    with pytest.raises(ValueError):
        eng.propagate(0.1)
    eng.close()
