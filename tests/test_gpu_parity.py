"""GPU parity tests proper: the HIP path, called through the C ABI, against
(a) the reference's golden vectors and (b) the NumPy oracle on seeded inputs.

Tolerances (north_star): autocorrelation / populations / energies within 1e-8
relative, norm conserved to 1e-12."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fidelity(orc, a, b):
    return abs(orc.overlap(a, b)) / np.sqrt(abs(orc.overlap(a, a)) * abs(orc.overlap(b, b)))


def test_unit_golden_apply(golden):
    from pytdscf_amd import engine as E

    g = golden("unit_apply.npz")
    out = E.heff_apply(g["L"], g["W"], g["R"], g["psi"])
    np.testing.assert_allclose(out, g["sigma"], rtol=1e-12, atol=1e-12)
    out = E.keff_apply(g["L"], g["Rk"], g["sval"])
    np.testing.assert_allclose(out, g["sigma_k"], rtol=1e-12, atol=1e-12)
    g = golden("unit_env.npz")
    np.testing.assert_allclose(E.env_update(g["L"], g["A"], g["W"], left=True), g["outA"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(E.env_update(g["R"], g["B"], g["W"], left=False), g["outB"], rtol=1e-12, atol=1e-12)


def test_unit_golden_gauge(golden):
    from pytdscf_amd import engine as E

    g = golden("unit_gauge.npz")
    A, s = E.gauge_trf(g["psi"], "Psi2Asigma")
    np.testing.assert_allclose(A, g["A"], atol=1e-12)
    np.testing.assert_allclose(s, g["sigA"], atol=1e-12)
    B, s = E.gauge_trf(g["psi"], "Psi2sigmaB")
    np.testing.assert_allclose(B, g["B"], atol=1e-12)
    np.testing.assert_allclose(s, g["sigB"], atol=1e-12)


@pytest.mark.parametrize("tag", ["lan_dt001", "lan_dt01", "lan_real", "arn_dt001", "arn_dt01"])
def test_unit_golden_krylov(golden, tag):
    from pytdscf_amd import engine as E

    g = golden("unit_krylov.npz")
    mat = g["Hh"] if tag.startswith("lan") else g["Hn"]
    integ = "lanczos" if tag.startswith("lan") else "arnoldi"
    cn = bool(g[tag + "_cn"])
    scale = complex(g[tag + "_scale"])
    x = g[tag + "_in"]
    y1, k1 = E.expm_dense(mat, x.reshape(-1), scale, integ, cn, 1e-9, 0)
    y2, k2 = E.expm_dense(mat, y1, scale, integ, cn, 1e-9, k1)
    assert [k1, k2] == list(g[tag + "_k"])
    np.testing.assert_allclose(y1.reshape(x.shape), g[tag + "_y1"], atol=1e-11)
    np.testing.assert_allclose(y2.reshape(x.shape), g[tag + "_y2"], atol=1e-11)


@pytest.mark.parametrize(
    "name,integ,cn,steps",
    [("chain_lanczos.npz", "lanczos", True, (1, 4)), ("chain_arnoldi.npz", "arnoldi", False, (1, 3))],
)
def test_chain_golden(golden, name, integ, cn, steps):
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    g = golden(name)
    n = int(g["nsite"])
    mpo = [g[f"mpo{i}"] for i in range(n)]
    init = [g[f"init{i}"] for i in range(n)]
    dt = float(g["dt_au"])
    for ns in steps:
        eng = TDVPEngine(n, integrator=integ, conserve_norm=cn)
        eng.set_mpo(mpo)
        eng.set_mps(init, canonicalize=True)
        e_last = None
        for _ in range(ns):
            e_last = eng.expectation()
            eng.propagate(dt)
        ref = [g[f"n{ns}_final{i}"] for i in range(n)]
        fin = eng.get_mps()
        assert eng.krylov_stats() == list(g[f"n{ns}_krylov"])
        assert abs(e_last.real - float(g[f"n{ns}_energy_last"])) < 1e-8 * abs(float(g[f"n{ns}_energy_last"]))
        assert abs(eng.norm() - float(g[f"n{ns}_norm"])) < 1e-12
        ac = complex(g[f"n{ns}_autocorr"])
        assert abs(eng.autocorr() - ac) < 1e-8 * abs(ac)
        ef = float(g[f"n{ns}_energy_final"].real)
        assert abs(eng.expectation().real - ef) < 1e-8 * abs(ef)
        assert abs(_fidelity(orc, ref, fin) - 1) < 1e-10
        eng.close()


def test_exciton_reference_pin(golden):
    """The reference's own regression pin (tests/test_exiciton_propagate.py:174-184):
    potential (3-leg diagonal cores) + kinetic (sites 0-2 only) merged into one MPO
    by direct sum; rank-deficient (product state, bond_dim=2) start."""
    from pytdscf_amd import TDVPEngine
    from pytdscf_amd.operators import merge_operator_terms

    g = golden("exciton.npz")
    pot = [g[f"pot{i}"] for i in range(4)]
    kin = [g[f"kin{i}"] for i in range(3)]
    mpo = merge_operator_terms([(pot, [0, 1, 2, 3]), (kin, [0, 1, 2])], dims=[8, 8, 8, 2])
    cores = [g[f"w{i}"].reshape(1, 8, 1) for i in range(3)] + [np.array([0.0, 1.0]).reshape(1, 2, 1)]
    from pytdscf_amd.mps import product_state_cores

    init = product_state_cores([c.reshape(-1) for c in cores], bond_dim=2)
    dt = float(g["dt_au"])
    eng = TDVPEngine(4)
    eng.set_mpo(mpo)
    eng.set_mps(init, canonicalize=True)
    e = None
    for step in range(20):
        if step == 19:
            rdm = eng.site_rdm(3)
        e = eng.expectation()
        eng.propagate(dt)
    assert e.real == pytest.approx(float(g["ref_pin_energy"]))  # reference's own tolerance (rel 1e-6)
    np.testing.assert_allclose(rdm, g["ref_pin_rdm33"], atol=1e-9)
    assert abs(eng.norm() - 1.0) < 1e-12


def test_oracle_parity_random_chain():
    """Seeded larger chain: HIP engine vs NumPy oracle, several steps."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    L, d, M, D = 8, 4, 5, 16
    mpo = orc.synthetic_mpo(L, d, M, seed=3)
    mps = orc.synthetic_mps([d] * L, D, seed=4)
    dt = 0.5
    st = orc.OracleMPS([c.copy() for c in mps], mpo)
    eng = TDVPEngine(L)
    eng.set_mpo(mpo)
    eng.set_mps(mps)
    for _ in range(3):
        st.propagate(dt)
        eng.propagate(dt)
    assert eng.krylov_stats() == [st.kprev[i] for i in range(L)]
    assert abs(eng.norm() - 1.0) < 1e-12
    e0, e1 = st.expectation(), eng.expectation()
    assert abs(e0 - e1) < 1e-8 * abs(e0)
    a0, a1 = st.autocorr(), eng.autocorr()
    assert abs(a0 - a1) < 1e-8 * abs(a0)
    assert abs(_fidelity(orc, st.cores, eng.get_mps()) - 1) < 1e-10


def test_device_random_state_properties():
    """Size-independent properties on a device-generated state: canonical form,
    norm and energy conservation, time reversal (propagate dt then -dt)."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    L, d, M, D = 10, 6, 6, 48
    eng = TDVPEngine(L)
    eng.set_mpo(orc.synthetic_mpo(L, d, M, seed=7))
    eng.init_random([d] * L, D, seed=5)
    cores = eng.get_mps()
    assert [c.shape[:1] + c.shape[2:] for c in cores] == orc.bond_dims([d] * L, D)
    for c in cores[1:]:
        m = c.reshape(c.shape[0], -1)
        assert np.abs(m @ m.conj().T - np.eye(c.shape[0])).max() < 1e-13
    assert abs(eng.norm() - 1) < 1e-14
    e0 = eng.expectation()
    assert abs(e0.imag) < 1e-12
    before = eng.get_mps()
    eng.propagate(0.4)
    eng.propagate(0.4)
    assert abs(eng.norm() - 1) < 1e-12
    assert abs(eng.expectation() - e0) < 1e-6 * abs(e0)  # TDVP conserves <H>
    eng.propagate(-0.4)
    eng.propagate(-0.4)
    assert abs(_fidelity(orc, before, eng.get_mps()) - 1) < 1e-7  # reversible up to thresh_sil accumulation


def test_nonconvergence_raises_value_error():
    """Like the reference (_integrator.py:653): more than 20 Krylov vectors -> ValueError,
    re-raised through the C ABI status code MITDVP_ENOTCONV."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    L, d, M, D = 6, 4, 4, 8
    eng = TDVPEngine(L)
    eng.set_mpo([50.0 * w for w in orc.synthetic_mpo(L, d, M, seed=0)])
    eng.set_mps(orc.synthetic_mps([d] * L, D, seed=1))
    with pytest.raises(ValueError, match="not converged"):
        eng.propagate(50.0)


def test_bad_arguments_raise():
    from pytdscf_amd import TDVPEngine

    eng = TDVPEngine(3)
    with pytest.raises(ValueError):
        eng.set_site(5, np.zeros((1, 2, 1)))
    with pytest.raises(ValueError):
        eng.propagate(0.1)  # nothing set
    eng.set_site(0, np.ones((1, 2, 2)), "Psi")
    eng.set_site(1, np.ones((3, 2, 2)), "B")  # bond mismatch 2 != 3
    eng.set_site(2, np.ones((2, 2, 1)), "B")
    with pytest.raises(ValueError, match="mismatch"):
        eng.expectation()
    # NULL pointers at the C boundary are argument errors, not crashes
    import ctypes as C

    from pytdscf_amd import _lib

    lib = _lib.load()
    assert lib.mitdvp_norm(eng._h, None) == _lib.EINVAL
    assert lib.mitdvp_get_site(eng._h, 0, None) == _lib.EINVAL
    assert lib.mitdvp_set_site(eng._h, 0, None, 1, 2, 2, 0) == _lib.EINVAL
    assert lib.mitdvp_expect(eng._h, 0, None) == _lib.EINVAL
    assert lib.mitdvp_norm(None, C.byref(C.c_double())) == _lib.EINVAL
    assert b"null" in lib.mitdvp_last_error(eng._h)
    eng.close()


def test_oracle_parity_liouvillian_arnoldi():
    """Non-Hermitian (vectorised-Lindblad-like) MPO, Arnoldi, conserve_norm=False:
    the C5 workload at a size the oracle finishes in seconds."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    L, D = 8, 16
    mpo = orc.synthetic_liouvillian_mpo(L, 16, seed=0, gamma=0.02)
    mps = orc.synthetic_mps([4] * L, D, seed=2)
    st = orc.OracleMPS([c.copy() for c in mps], mpo, integrator="arnoldi", conserve_norm=False)
    eng = TDVPEngine(L, integrator="arnoldi", conserve_norm=False)
    eng.set_mpo(mpo)
    eng.set_mps(mps)
    for _ in range(3):
        st.propagate(0.5)
        eng.propagate(0.5)
    assert eng.krylov_stats() == [st.kprev[i] for i in range(L)]
    assert abs(eng.norm() - st.norm()) < 1e-10 * st.norm()  # the norm decays: compare, do not expect 1
    assert st.norm() < 0.99
    e0, e1 = st.expectation(), eng.expectation()
    assert abs(e0 - e1) < 1e-8 * abs(e0)
    assert abs(_fidelity(orc, st.cores, eng.get_mps()) - 1) < 1e-10


def test_chain_relaxation_golden(golden):
    """Imaginary-time relaxation (Simulator.relax(improved=False)) against the reference."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    g = golden("chain_relax.npz")
    n = int(g["nsite"])
    mpo = [g[f"mpo{i}"] for i in range(n)]
    init = [g[f"init{i}"] for i in range(n)]
    dt = float(g["dt_au"])
    eng = TDVPEngine(n, relax=True)
    eng.set_mpo(mpo)
    eng.set_mps(init, canonicalize=True)
    e_last = None
    for _ in range(5):
        e_last = eng.expectation()
        eng.propagate(dt)
    assert eng.krylov_stats() == list(g["n5_krylov"])
    assert abs(e_last.real - float(g["n5_energy_last"])) < 1e-8 * abs(float(g["n5_energy_last"]))
    assert abs(eng.expectation().real - float(g["n5_energy_final"])) < 1e-8 * abs(float(g["n5_energy_final"]))
    assert abs(eng.norm() - 1) < 1e-12
    ref = [g[f"n5_final{i}"] for i in range(n)]
    assert abs(_fidelity(orc, ref, eng.get_mps()) - 1) < 1e-10


def test_reduced_densities_golden(golden):
    """get_reduced_densities for the key types of the reference's exciton test
    ((3,3), (0,0), (0,0,3,3)) and diagonal-only keys, after 19 steps."""
    from pytdscf_amd import TDVPEngine
    from pytdscf_amd.mps import product_state_cores
    from pytdscf_amd.operators import merge_operator_terms

    g = golden("exciton.npz")
    pot = [g[f"pot{i}"] for i in range(4)]
    kin = [g[f"kin{i}"] for i in range(3)]
    mpo = merge_operator_terms([(pot, [0, 1, 2, 3]), (kin, [0, 1, 2])], dims=[8, 8, 8, 2])
    init = product_state_cores([g[f"w{i}"] for i in range(3)] + [np.array([0.0, 1.0])], bond_dim=2)
    eng = TDVPEngine(4)
    eng.set_mpo(mpo)
    eng.set_mps(init, canonicalize=True)
    for _ in range(19):
        eng.propagate(float(g["dt_au"]))
    for tag, legs in (("rdm33", (0, 0, 0, 2)), ("rdm00", (2,)), ("rdm0033", (2, 0, 0, 2)), ("rdm1", (0, 1)), ("rdm013", (1, 2, 0, 1))):
        out = eng.reduced_density(legs)
        assert out.shape == g[f"n19_{tag}"].shape
        np.testing.assert_allclose(out, g[f"n19_{tag}"], atol=1e-9)
    with pytest.raises(ValueError):
        eng.reduced_density((0, 3))


def test_reduced_density_vs_oracle_full_rank():
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    L, d, D = 6, 3, 9
    mps = orc.synthetic_mps([d] * L, D, seed=9)
    eng = TDVPEngine(L)
    eng.set_mpo(orc.synthetic_mpo(L, d, 4, seed=1))
    eng.set_mps(mps)
    for legs in [(2,), (0, 2), (1, 0, 2), (2, 2), (0, 1, 1, 2), (2, 0, 0, 0, 0, 2), (1, 1, 1)]:
        ref = orc.reduced_density(mps, legs)
        out = eng.reduced_density(legs)
        np.testing.assert_allclose(out, ref, atol=1e-13)


@pytest.mark.parametrize(
    "dims,D,M",
    [([4], 8, 1), ([3, 5], 7, 3), ([2, 5, 3, 4], 1000, 4), ([6, 2, 2, 2, 2, 6], 5, 3), ([2] * 12, 16, 5)],
)
def test_edge_chains_vs_oracle(dims, D, M):
    """Single site, two sites, ragged physical dimensions, exact regime (D larger than
    the Hilbert space) and bond dimensions that are not multiples of any tile size."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    L = len(dims)
    rng = np.random.default_rng(sum(dims) + D)

    def herm(d, s):
        g = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
        return s * (g + g.conj().T) / 2

    mpo = []
    for p, d in enumerate(dims):  # H = sum_p h_p + sum_p a_p a_{p+1} (+ more coupling channels)
        w = np.zeros((M, d, d, M), dtype=np.complex128)
        w[0, :, :, 0] = np.eye(d)
        w[M - 1, :, :, M - 1] = np.eye(d)
        w[0, :, :, M - 1] = herm(d, 0.3)
        for k in range(1, M - 1):
            a = herm(d, 0.2)
            w[0, :, :, k] = a
            w[k, :, :, M - 1] = a
        if p == 0:
            w = w[0:1] if M > 1 else w
        if p == L - 1:
            w = w[:, :, :, M - 1 : M]
        if L == 1:
            w = herm(d, 0.3).reshape(1, d, d, 1)
        mpo.append(np.ascontiguousarray(w))
    mps = orc.synthetic_mps(dims, D, seed=11)
    st = orc.OracleMPS([c.copy() for c in mps], mpo)
    eng = TDVPEngine(L)
    eng.set_mpo(mpo)
    eng.set_mps(mps)
    for _ in range(3):
        st.propagate(0.3)
        eng.propagate(0.3)
    assert eng.krylov_stats() == [st.kprev[i] for i in range(L)]
    assert abs(eng.norm() - 1) < 1e-12
    assert abs(eng.expectation() - st.expectation()) < 1e-8 * max(abs(st.expectation()), 1e-3)
    assert abs(eng.autocorr() - st.autocorr()) < 1e-8
    assert abs(_fidelity(orc, st.cores, eng.get_mps()) - 1) < 1e-10


def test_chain_improved_relaxation_golden(golden):
    """Simulator.relax(improved=True) of the reference: energies and the state (up to the
    free global phase of an eigenvector) after 1 and 3 sweeps."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    g = golden("chain_improved_relax.npz")
    n = int(g["nsite"])
    mpo = [g[f"mpo{i}"] for i in range(n)]
    init = [g[f"init{i}"] for i in range(n)]
    for ns in (1, 3):
        eng = TDVPEngine(n, relax="improved")
        eng.set_mpo(mpo)
        eng.set_mps(init, canonicalize=True)
        e_last = None
        for _ in range(ns):
            e_last = eng.expectation()
            eng.propagate(0.0)
        assert abs(e_last.real - float(g[f"n{ns}_energy_last"])) < 1e-8 * abs(float(g[f"n{ns}_energy_last"]))
        assert abs(eng.expectation().real - float(g[f"n{ns}_energy_final"])) < 1e-8 * abs(float(g[f"n{ns}_energy_final"]))
        assert abs(eng.norm() - 1) < 1e-12
        ref = [g[f"n{ns}_final{i}"] for i in range(n)]
        assert abs(_fidelity(orc, ref, eng.get_mps()) - 1) < 1e-9


def test_liouville_space_golden(golden):
    """space="Liouville" through the shell: trace-normalised start, Arnoldi,
    conserve_norm forced off; Tr(O rho) and partial traces against the reference."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import Exciton, Model, Simulator, TensorHamiltonian, TensorOperator

    g = golden("chain_liouville.npz")
    n = int(g["nsite"])
    mpo = [g[f"mpo{i}"] for i in range(n)]
    obs1 = TensorHamiltonian(ndof=n, potential=[[{((2, 2),): TensorOperator(mpo=[g["sz"]], legs=(2, 2))}]], kinetic=None)
    obs2 = TensorHamiltonian(ndof=n, potential=[[{((1, 1), (3, 3)): TensorOperator(mpo=[g["sz"], g["sx"]], legs=(1, 1, 3, 3))}]], kinetic=None)
    model = Model([Exciton(nstate=4) for _ in range(n)], operators={"hamiltonian": mpo, "sz2": obs1, "sz1sx3": obs2},
                  bond_dim=int(g["bond_dim"]), space="liouville")
    model.init_HartreeProduct = [[g[f"rho{i}"] for i in range(n)]]
    dt_fs = 0.02
    for ns in (1, 3):
        sim = Simulator("liou", model, backend="hip")
        import os, tempfile

        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as td:
            os.chdir(td)
            try:
                _, wf = sim.propagate(stepsize=dt_fs, maxstep=ns, integrator="arnoldi", autocorr=False, energy=False)
            finally:
                os.chdir(cwd)
        assert wf.engine.krylov_stats() == list(g[f"n{ns}_krylov"])
        assert abs(wf.norm() - float(g[f"n{ns}_norm"])) < 1e-10 * float(g[f"n{ns}_norm"])
        assert abs(wf.expectation(model.observables["sz2"]) - float(g[f"n{ns}_sz2"])) < 1e-8 * abs(float(g[f"n{ns}_sz2"]))
        assert abs(wf.expectation(model.observables["sz1sx3"]) - float(g[f"n{ns}_sz1sx3"])) < 1e-8 * abs(float(g[f"n{ns}_sz1sx3"]))
        for tag, legs in (("pt2", (0, 0, 2)), ("pt04", (2, 0, 0, 0, 2)), ("pt1d", (0, 1)), ("pt13", (0, 2, 0, 1))):
            out = wf.get_reduced_densities(legs)[0]
            assert out.shape == g[f"n{ns}_{tag}"].shape
            np.testing.assert_allclose(out, g[f"n{ns}_{tag}"], atol=1e-10)
        ref = [g[f"n{ns}_final{i}"] for i in range(n)]
        assert abs(_fidelity(orc, ref, wf.get_mps()) - 1) < 1e-10


@pytest.mark.parametrize("tag,p", [("p1e-2", 1e-2), ("p1e-6", 1e-6), ("p0", 0.0)])
def test_truncate_bond_golden(golden, tag, p):
    """truncate_sigvec (_site_cls.py:586-690) on A sigma B from the reference: same kept
    rank, same normalised singular values, same contracted two-site tensor."""
    from pytdscf_amd import TDVPEngine

    g = golden("unit_truncate.npz")
    A, sigma, B = g["A"], g["sigma"], g["B"]
    psi = np.tensordot(A, sigma, axes=(2, 0))
    # embed as a 4-site chain: [left cap] [Psi] [B] [right cap]
    capl = np.linalg.qr(np.eye(5)[:, :5])[0].reshape(1, 5, 5)
    capr = np.eye(4).reshape(4, 4, 1)
    eng = TDVPEngine(4)
    eng.set_site(0, capl, "A")
    eng.set_site(1, psi, "Psi")
    eng.set_site(2, B, "B")
    eng.set_site(3, capr, "B")
    nd, sv = eng.truncate_bond(p)
    ref_sig = g[f"{tag}_sig"]
    assert nd == ref_sig.shape[0]
    np.testing.assert_allclose(sv, np.diag(ref_sig).real, rtol=1e-10)
    two = np.einsum("ajk,kmr->ajmr", eng.get_site(1), eng.get_site(2))
    np.testing.assert_allclose(two, g[f"{tag}_two_site"], atol=1e-12)
    Bn = eng.get_site(2).reshape(nd, -1)
    assert np.abs(Bn @ Bn.conj().T - np.eye(nd)).max() < 1e-13  # B stays right-canonical


def test_long_trace_against_oracle():
    """40 time steps on a seeded chain: the autocorrelation / energy / norm traces (the
    acceptance quantities) stay within 1e-8 of the oracle at EVERY step and the Krylov
    iteration memory follows the same path (identical counts at the end)."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    L, d, M, D = 7, 3, 4, 9
    mpo = orc.synthetic_mpo(L, d, M, seed=4)
    rng = np.random.default_rng(17)
    init = [rng.standard_normal((a, d, b)) + 1j * rng.standard_normal((a, d, b)) for a, b in orc.bond_dims([d] * L, D)]
    st = orc.OracleMPS(orc.canonicalize_site0(init), mpo)
    eng = TDVPEngine(L)
    eng.set_mpo(mpo)
    eng.set_mps(init, canonicalize=True)
    dt = 0.8
    worst = 0.0
    for step in range(40):
        a_o, a_e = st.autocorr(), eng.autocorr()
        e_o, e_e = st.expectation(), eng.expectation()
        worst = max(worst, abs(a_o - a_e) / abs(a_o), abs(e_o - e_e) / abs(e_o))
        assert abs(eng.norm() - 1) < 1e-12
        st.propagate(dt)
        eng.propagate(dt)
    assert worst < 1e-8, worst
    assert eng.krylov_stats() == [st.kprev[i] for i in range(L)]
    assert abs(_fidelity(orc, st.cores, eng.get_mps()) - 1) < 1e-10
    eng.close()


def test_overlap_with_saved_reference_state():
    """<Psi(0)|Psi(t)> (autocorrelation without the t/2 trick) against the oracle, also when the
    bond dimensions of Psi(t) have grown adaptively since the copy was taken."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    L, d, M, D = 6, 3, 4, 2
    mpo = orc.synthetic_mpo(L, d, M, seed=6)
    rng = np.random.default_rng(2)
    init = [rng.standard_normal((a, d, b)) + 1j * rng.standard_normal((a, d, b)) for a, b in orc.bond_dims([d] * L, D)]
    c0 = orc.canonicalize_site0(init)
    kw = dict(Dmax=6, dD=2, p_proj=1e-9)
    st = orc.OracleMPS([c.copy() for c in c0], mpo, adaptive=True, **kw)
    eng = TDVPEngine(L)
    eng.set_mpo(mpo)
    eng.set_mps(init, canonicalize=True)
    eng.set_adaptive(True, **kw)
    with pytest.raises(ValueError):
        eng.overlap_reference()
    eng.save_reference()
    assert abs(eng.overlap_reference() - 1) < 1e-12
    for _ in range(3):
        st.propagate(0.6)
        eng.propagate(0.6)
        ref = orc.overlap(c0, st.cores)
        assert abs(eng.overlap_reference() - ref) < 1e-9 * abs(ref)
    assert max(eng.bond_dims()) > D
    eng.close()
