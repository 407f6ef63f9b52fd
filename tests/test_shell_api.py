"""The PyTDSCF-shaped user surface: scripts of the reference's own
tests, re-typed against ``pytdscf_amd`` (same class names / kwargs)."""

import numpy as np
import pytest

import os

from oracle import tdvp_oracle as orc

ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _exciton_model(g):
    from pytdscf_amd import Exciton, HarmonicOscillator as HO, Model, TensorHamiltonian, TensorOperator

    prim_info = [HO(8, f, units="cm-1") for f in (1000, 2000, 3000)] + [Exciton(nstate=2, names=["S0", "S1"])]
    pot = [g[f"pot{i}"] for i in range(4)]
    kin = [g[f"kin{i}"] for i in range(3)]
    potential = [[{(0, 1, 2, (3, 3)): TensorOperator(mpo=pot, legs=(0, 1, 2, 3, 3))}]]
    kinetic = [[{((0, 0), (1, 1), (2, 2)): TensorOperator(mpo=kin, legs=(0, 0, 1, 1, 2, 2))}]]
    hamiltonian = TensorHamiltonian(ndof=4, potential=potential, kinetic=kinetic, backend="hip")
    model = Model(prim_info, {"hamiltonian": hamiltonian}, bond_dim=2)
    model.init_HartreeProduct = [[ho.get_unitary()[0].tolist() for ho in prim_info[:3]] + [np.array([0.0, 1.0]).tolist()]]
    return model


def test_model_reduction_cpu(golden):
    """Model -> (one MPO, initial cores) reproduces the reference pins through the oracle."""
    g = golden("exciton.npz")
    model = _exciton_model(g)
    raw = model.hamiltonian.as_mpo(model.dims, compress=False)
    assert [w.shape for w in raw] == [(1, 8, 8, 5), (5, 8, 8, 6), (6, 8, 8, 4), (4, 2, 2, 1)]
    mpo = model.hamiltonian.as_mpo(model.dims)  # lossless rounding of the direct sum
    assert [w.shape for w in mpo] == [(1, 8, 8, 4), (4, 8, 8, 5), (5, 8, 8, 3), (3, 2, 2, 1)]
    from pytdscf_amd.operators import mpo_to_dense

    np.testing.assert_allclose(mpo_to_dense(mpo), mpo_to_dense(raw), atol=1e-15)
    st = orc.OracleMPS(orc.canonicalize_site0(model.initial_cores()), mpo)
    e = None
    for _ in range(20):
        e = st.expectation()
        st.propagate(float(g["dt_au"]))
    assert e.real == pytest.approx(float(g["ref_pin_energy"]))


def test_units_match_reference(golden):
    from pytdscf_amd import units

    g = golden("henon_heiles.npz")
    assert units.au_in_fs == float(g["au_in_fs"])
    assert units.au_in_cm1 == float(g["au_in_cm1"])


def test_unsupported_combinations_raise():
    from pytdscf_amd import Exciton, Model, Simulator

    core = np.zeros((1, 2, 2, 1))
    m_sub = Model([Exciton(4)], [np.zeros((1, 4, 4, 1))], bond_dim=2, space="liouville", subspace_inds={0: (0, 3)})
    assert m_sub.projected_dims() == [2] and m_sub.project_mpo([np.zeros((1, 4, 4, 1))])[0].shape == (1, 2, 2, 1)
    with pytest.raises(ValueError):  # index outside the site's 4 physical entries
        Model([Exciton(4)], [np.zeros((1, 4, 4, 1))], bond_dim=2, space="liouville", subspace_inds={0: (0, 4)})
    with pytest.raises(ValueError):
        Model([Exciton(2)], [core], bond_dim=2, space="fock")
    with pytest.raises(ValueError):
        Model([Exciton(2), Exciton(2)], {"hamiltonian": [core]}, bond_dim=2)
    m = Model([Exciton(2)], [core], bond_dim=2)
    with pytest.raises(NotImplementedError):
        Simulator("j", m, ci_type="MCTDH")


def test_parallel_split_indices_must_match_the_ranks(golden, monkeypatch):
    """_const_cls.py:237 -- one site range per rank; unsupported combinations are refused before any rendezvous."""
    from pytdscf_amd import Simulator

    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    sim = Simulator("j", _exciton_model(golden("exciton.npz")), backend="hip")
    with pytest.raises(ValueError, match="2 ranges but the job has 1 rank"):
        sim.propagate(maxstep=1, parallel_split_indices=[(0, 1), (2, 3)])
    with pytest.raises(ValueError, match="2 ranges but the job has 1 rank"):  # adaptive ranks across junctions are built (round 5)
        sim.propagate(maxstep=1, parallel_split_indices=[(0, 1), (2, 3)], adaptive=True)
    with pytest.raises(NotImplementedError):
        sim.propagate(maxstep=1, parallel_split_indices=[(0, 1), (2, 3)], restart=True)


def test_sharded_adaptive_plumbing_of_the_shell(golden, tmp_path, monkeypatch):
    """``propagate(parallel_split_indices=..., adaptive=True, ...)``: the adaptive settings reach the sharded engine
    (const.adaptive / Dmax / dD / p_proj, _const_cls.py:212-216; p_svd and the junction regularisation as before), the
    bond dimensions of ``bonddim.dat`` come from the shapes the ranks share after every step (properties.py:255-262), and
    the serial engine's ``set_adaptive`` is not called on the sharded one.  Host logic only: the sharded engine and the
    communicator are stand-ins."""
    from pytdscf_amd import Simulator, api, dist, parallel_sites

    seen = {}

    class FakeDist:
        @staticmethod
        def broadcast_object_list(box, src=0):
            return None

    class FakeComm:
        rank, world, dist = 0, 2, FakeDist()

    class FakeShard:
        def __init__(self, comm, mpo, **kw):
            seen.update(kw, nsite=len(mpo))
            self.rank, self.steps = 0, 0

        def autocorr(self):
            return 1.0 + 0.0j

        def norm(self):
            return 1.0

        def expectation(self, op=None):
            return 0.01 + 0.0j

        def step(self, dt):
            self.steps += 1

        def bond_dims(self):
            return [1, 1, 1] if self.steps == 0 else [8, 7, 2]

        def gather(self):
            return None

        def close(self):
            seen["closed"] = True

    monkeypatch.setattr(dist, "world_comm", lambda site_sharding=False: FakeComm())
    monkeypatch.setattr(parallel_sites, "SiteShardedTDVP", FakeShard)
    monkeypatch.setattr(api.Simulator, "_gathered_wfunc", lambda self, eng, *a: None)
    monkeypatch.chdir(tmp_path)
    sim = Simulator("plumb", _exciton_model(golden("exciton.npz")), backend="hip")
    ener, wf = sim.propagate(stepsize=0.05, maxstep=3, parallel_split_indices=[(0, 1), (2, 3)], adaptive=True, adaptive_Dmax=60,
                             adaptive_dD=60, adaptive_p_proj=1e-5, adaptive_p_svd=1e-6)
    assert seen["adaptive"] == dict(Dmax=60, dD=60, p_proj=1e-5) and seen["p_svd"] == 1e-6 and seen["regularize"] is True
    assert seen["split"] == [(0, 1), (2, 3)] and seen["nsite"] == 4 and seen["closed"]
    assert ener == pytest.approx(0.01) and wf is None
    lines = open(tmp_path / "plumb_prop" / "bonddim.dat").read().splitlines()
    assert [l.split()[1:] for l in lines[1:]] == [["1", "1", "1"], ["8", "7", "2"], ["8", "7", "2"]]
    seen.clear()
    sim.propagate(stepsize=0.05, maxstep=1, parallel_split_indices=[(0, 1), (2, 3)])
    assert seen["adaptive"] is None


@pytest.mark.gpu
def test_exciton_script_on_gpu(golden, tmp_path, monkeypatch):
    """tests/test_exiciton_propagate.py of the reference, through the shell."""
    from pytdscf_amd import Simulator

    monkeypatch.chdir(tmp_path)
    g = golden("exciton.npz")
    simulator = Simulator("LVC_Exciton_test", _exciton_model(g), backend="hip")
    ener_calc, wf = simulator.propagate(stepsize=0.1, maxstep=20, reduced_density=([(3, 3)], 1))
    assert pytest.approx(ener_calc) == 0.010000180312707298
    t, rdm = simulator.rdm_trace[-1]
    np.testing.assert_allclose(rdm[(3, 3)], g["ref_pin_rdm33"], atol=1e-9)
    assert abs(wf.norm() - 1) < 1e-12
    lines = open(tmp_path / "LVC_Exciton_test_prop" / "autocorr.dat").read().splitlines()
    assert lines[0].startswith("# time [fs]") and len(lines) == 21
    # reduced_density.nc in the reference's layout (properties.py:156-209), read back by the analysis-side reader of
    # the reference (util/read_nc.py)
    from pytdscf_amd.util import read_nc

    data = read_nc(str(tmp_path / "LVC_Exciton_test_prop" / "reduced_density.nc"), [(3, 3)])
    assert data["time"].shape == (20,) and data[(3, 3)].shape == (20, 2, 2)
    np.testing.assert_allclose(data[(3, 3)][-1], g["ref_pin_rdm33"], atol=1e-9)
    # and the spectrum chain on the auto-correlation file just written
    from pytdscf_amd import spectra

    t_fs, ac = spectra.load_autocorr(str(tmp_path / "LVC_Exciton_test_prop" / "autocorr.dat"))
    assert t_fs[1] == pytest.approx(0.2) and len(ac) == 20  # t/2 trick: the file's time axis is 2 t
    freq, inten = spectra.ifft_autocorr(t_fs, ac)
    assert len(freq) == len(inten) > 0


@pytest.mark.gpu
def test_henon_heiles_script_on_gpu(golden, tmp_path, monkeypatch):
    """tests/test_henon_heiles.py NumPy case through the shell (MPO cores from the fixture)."""
    from pytdscf_amd import HarmonicOscillator as HO, Model, Simulator

    monkeypatch.chdir(tmp_path)
    g = golden("henon_heiles.npz")
    dvr_prims = [HO(5, 2000) for _ in range(2)]
    operators = {"potential": [g["pot0"], g["pot1"]], "kinetic": [g["kin0"], g["kin1"]]}
    model = Model(dvr_prims, operators=operators, bond_dim=4)
    model.init_weight_VIBSTATE = [[[0.0, 1.0, 0.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0, 0.0]]]
    simulator = Simulator(jobname="henon_heiles", model=model, backend="hip")
    ener_calc, wf = simulator.propagate(maxstep=3, stepsize=0.001)
    assert pytest.approx(ener_calc) == 0.018225341011652626
    ref = [g["n3_final0"], g["n3_final1"]]
    assert abs(abs(orc.overlap(ref, wf.get_mps())) - 1) < 1e-9


@pytest.mark.gpu
def test_a1tdvp_script_on_gpu(golden, tmp_path, monkeypatch):
    """tests/test_a1tdvp.py of the reference through the shell: the exciton model from
    bond dimension 1 with ``propagate(adaptive=True, ...)``.  The reference's test only
    checks that it runs; here the ranks are pinned to its golden run as well."""
    from pytdscf_amd import Exciton, HarmonicOscillator as HO, Model, Simulator, TensorHamiltonian, TensorOperator

    monkeypatch.chdir(tmp_path)
    g = golden("adaptive_exciton.npz")
    prim_info = [HO(8, f, units="cm-1") for f in (1000, 2000, 3000)] + [Exciton(nstate=2, names=["S0", "S1"])]
    pot = [g[f"pot{i}"] for i in range(4)]
    kin = [g[f"kin{i}"] for i in range(3)]
    hamiltonian = TensorHamiltonian(
        ndof=4,
        potential=[[{(0, 1, 2, (3, 3)): TensorOperator(mpo=pot, legs=(0, 1, 2, 3, 3))}]],
        kinetic=[[{((0, 0), (1, 1), (2, 2)): TensorOperator(mpo=kin, legs=(0, 0, 1, 1, 2, 2))}]],
        backend="hip",
    )
    model = Model(prim_info, {"hamiltonian": hamiltonian}, bond_dim=1)
    model.init_HartreeProduct = [[ho.get_unitary()[0].tolist() for ho in prim_info[:3]] + [np.array([0.0, 1.0]).tolist()]]
    simulator = Simulator("a1tdvp", model, backend="hip")
    ener, wf = simulator.propagate(
        stepsize=0.1, maxstep=10, adaptive=True, adaptive_Dmax=int(g["Dmax"]), adaptive_dD=int(g["dD"]),
        adaptive_p_proj=float(g["p_proj"]),
    )
    assert wf.bonddim() == list(g["n10_bonddim"])
    assert ener == pytest.approx(float(g["n10_energy_last"]), rel=1e-5)
    assert abs(wf.norm() - 1) < 1e-12
    lines = open(tmp_path / "a1tdvp_prop" / "bonddim.dat").read().splitlines()
    assert len(lines) == 11 and lines[1].split()[1:] == ["1", "1", "1"] and lines[-1].split()[1:] == ["5", "5", "2"]


def test_one_gate_model_validation():
    from pytdscf_amd import Exciton, Model, TensorHamiltonian, TensorOperator

    core = np.zeros((1, 2, 2, 1))
    g2 = TensorHamiltonian(2, potential=[[{((0, 0),): TensorOperator(mpo=[np.eye(2)[None, :, :, None]], legs=(0, 0)),
                                           (0,): TensorOperator(mpo=[np.ones(2)[None, :, None]], legs=(0,))}]], kinetic=None)
    m = Model([Exciton(2), Exciton(2)], {"hamiltonian": [core, core]}, bond_dim=2, one_gate_to_apply=g2)
    with pytest.raises(ValueError, match="Multiple one gate"):
        m.one_gate_to_apply.one_site_gates(m.dims)
    with pytest.raises(TypeError):
        Model([Exciton(2), Exciton(2)], {"hamiltonian": [core, core]}, bond_dim=2, one_gate_to_apply=[core])


@pytest.mark.gpu
def test_supergate_script_on_gpu(golden, tmp_path, monkeypatch):
    """tests/test_mixedstate.py::test_vectorised_density_matrix(supergate=True) pattern through
    the shell: Liouville-space Model with one_gate_to_apply, pinned to the reference's run."""
    from pytdscf_amd import Exciton, Model, Simulator, TensorHamiltonian, TensorOperator

    monkeypatch.chdir(tmp_path)
    g = golden("gate_liouville.npz")
    n = int(g["nsite"])
    gate = TensorHamiltonian(n, potential=[[{((2, 2),): TensorOperator(mpo=[g["G2"][None, :, :, None]], legs=(2, 2))}]], kinetic=None, backend="hip")
    sz2 = TensorHamiltonian(n, potential=[[{((2, 2),): TensorOperator(mpo=[g["sz"]], legs=(2, 2))}]], kinetic=None, backend="hip")
    model = Model([Exciton(nstate=4) for _ in range(n)], operators={"hamiltonian": [g[f"mpo{i}"] for i in range(n)], "sz2": sz2},
                  bond_dim=int(g["bond_dim"]), space="Liouville", one_gate_to_apply=gate)
    model.init_HartreeProduct = [[g[f"rho{i}"] for i in range(n)]]
    sim = Simulator("supergate", model, backend="hip")
    _, wf = sim.propagate(stepsize=0.02, maxstep=3, integrator="arnoldi", autocorr=False, energy=False, conserve_norm=False)
    assert wf.expectation(model.observables["sz2"]) == pytest.approx(float(g["n3_sz2"]), rel=1e-8)
    np.testing.assert_allclose(wf.get_reduced_densities((0, 0, 2))[0], g["n3_pt2"], atol=1e-10)


@pytest.mark.gpu
def test_kraus_script_on_gpu(golden, tmp_path, monkeypatch):
    """tests/test_mixedstate.py::test_purified_mps_kraus_single_site pattern through the shell:
    Model(kraus_op={(site,): Bs}), Arnoldi, conserve_norm=False; ancilla-traced density pinned
    to the reference's run (trace_kraus_dim, kraus.py:434-455)."""
    from pytdscf_amd import Exciton, Model, Simulator

    monkeypatch.chdir(tmp_path)
    g = golden("kraus_single.npz")
    n, d, K = 4, int(g["d"]), int(g["K"])
    dims = [g[f"mpo{i}"].shape[1] for i in range(n)]
    model = Model([Exciton(nstate=x) for x in dims], operators={"hamiltonian": [g[f"mpo{i}"] for i in range(n)]},
                  kraus_op={(1,): g["B"]}, bond_dim=int(g["bond_dim"]))
    model.init_HartreeProduct = [[g[f"w{i}"] for i in range(n)]]
    sim = Simulator("kraus", model, backend="hip")
    _, wf = sim.propagate(stepsize=0.05, maxstep=4, integrator="arnoldi", conserve_norm=False, autocorr=False, energy=False)
    assert wf.norm() == pytest.approx(float(g["n4_norm"]), rel=1e-9)
    r1 = wf.get_reduced_densities((0, 2))[0]
    np.testing.assert_allclose(np.einsum("dKxK->dx", r1.reshape(d, K, d, K)), g["n4_rdm1"], atol=1e-9)


@pytest.mark.gpu
def test_restart_from_saved_wavefunction(golden, tmp_path, monkeypatch):
    """relax(savefile_ext="_gs") -> propagate(restart=True, loadfile_ext="_gs") and a propagation
    continued from its own checkpoint equal the uninterrupted runs (simulator_cls.py:413-418, :500-506)."""
    from pytdscf_amd import Simulator

    monkeypatch.chdir(tmp_path)
    g = golden("exciton.npz")
    sim = Simulator("ckpt", _exciton_model(g), backend="hip")
    e_all, wf_all = sim.propagate(stepsize=0.1, maxstep=6, savefile_ext="_all")
    ref = wf_all.get_mps()
    sim2 = Simulator("ckpt", _exciton_model(g), backend="hip")
    sim2.propagate(stepsize=0.1, maxstep=3, savefile_ext="_half")
    assert (tmp_path / "wf_ckpt_half.pkl").exists()
    # the dill checkpoint carries the attribute graph the reference's readers walk (simulator_cls.py:577-589,
    # :501-507): wf.ci_coef.superblock_states[istate][isite].data / .gauge / .isite
    import dill

    with open(tmp_path / "wf_ckpt_half.pkl", "rb") as f:
        saved = dill.load(f)
    sites = saved.ci_coef.superblock_states[0]
    assert saved.ci_coef.nsite == 4 and [s_.gauge for s_ in sites] == ["Psi", "B", "B", "B"] and [s_.isite for s_ in sites] == [0, 1, 2, 3]
    assert sites[1].data.shape[1] == 8 and sites[3].data.dtype == np.complex128
    e2, wf2 = sim2.propagate(stepsize=0.1, maxstep=3, restart=True, loadfile_ext="_half", savefile_ext="_rest")
    assert e2 == pytest.approx(e_all, rel=1e-9)
    assert abs(abs(orc.overlap(ref, wf2.get_mps())) - 1) < 1e-9
    with pytest.raises(FileNotFoundError):
        sim2.propagate(stepsize=0.1, maxstep=1, restart=True, loadfile_ext="_nope")
    # ground state by relaxation, then real time from it: the energy stays the relaxed one
    e_gs, _ = sim2.relax(stepsize=2.0, maxstep=6, improved=True, savefile_ext="_gs")
    e0, _ = sim2.propagate(stepsize=0.1, maxstep=2, restart=True, loadfile_ext="_gs")
    assert e0 == pytest.approx(e_gs, abs=1e-6)


@pytest.mark.gpu
def test_relax_operate_propagate_workflow(golden, tmp_path, monkeypatch):
    """The reference's spectrum workflow (tests/test_harmonic_dvr_func_full_mpssm_jax.py):
    relax -> operate(restart=True) with a "dipole" Model -> propagate(restart=True)."""
    from pytdscf_amd import Exciton, Model, Simulator

    monkeypatch.chdir(tmp_path)
    g = golden("operate_chain.npz")
    n = int(g["nsite"])
    basis = [Exciton(nstate=3) for _ in range(n)]
    ham = orc.synthetic_mpo(n, 3, 4, seed=0)
    dip = [g[f"mpo{i}"] for i in range(n)]
    m_h = Model(basis, operators={"hamiltonian": ham}, bond_dim=int(g["bond_dim"]))
    m_h.init_HartreeProduct = [[g[f"init{i}"] for i in range(n)]]
    m_d = Model(basis, operators={"hamiltonian": dip}, bond_dim=int(g["bond_dim"]))
    e_gs, wf_gs = Simulator("wfl", m_h, backend="hip").relax(stepsize=2.0, maxstep=4, improved=True)
    gs = wf_gs.get_mps()
    norm, wf_op = Simulator("wfl", m_d, backend="hip").operate(restart=True, maxstep=10)
    n_o, ref, _ = orc.operate(gs, dip, maxstep=10)
    assert norm == pytest.approx(n_o, rel=1e-9)
    assert abs(abs(orc.overlap(ref, wf_op.get_mps())) - 1) < 1e-9
    assert (tmp_path / "wf_wfl_operate.pkl").exists()
    _, wf_t = Simulator("wfl", m_h, backend="hip").propagate(stepsize=0.05, maxstep=2, restart=True)  # loadfile_ext="_operate"
    from pytdscf_amd import units

    st = orc.OracleMPS([c.copy() for c in ref], ham)
    for _ in range(2):
        st.propagate(0.05 / units.au_in_fs)
    assert abs(abs(orc.overlap(st.cores, wf_t.get_mps())) - 1) < 1e-8


def _harmonic_prims():
    from pytdscf_amd import HarmonicOscillator

    return [HarmonicOscillator(5, 1500, 0.0), HarmonicOscillator(5, 2000, 0.0), HarmonicOscillator(5, 2500, 0.0)]


def test_grid_tensor_decomposition_and_kinetic_mpo():
    """construct_fulldimensional / construct_kinetic_operator / TensorOperator.decompose: the
    MPO restores the grid function within the contribution rate; the kinetic MPO is the sum of
    the one-site second-derivative matrices."""
    from pytdscf_amd import TensorHamiltonian, TensorOperator
    from pytdscf_amd.dvr_operator_cls import construct_fulldimensional, construct_kinetic_mpo, construct_kinetic_operator
    from pytdscf_amd.operators import mpo_to_dense

    prims = _harmonic_prims()
    f = lambda a, b, c: 0.3 * a * a + 0.1 * a * b - 0.2 * np.sin(c) * b + 0.05 * a * b * c  # noqa: E731
    op = construct_fulldimensional(dvr_prims=prims, func=f, ref_ene=0.01)[(0, 1, 2)]
    ref = op.tensor_orig.copy()
    cores = op.decompose(decompose_type="SVD", rate=0.999999999999)
    assert [c.ndim for c in cores] == [3, 3, 3] and cores[0].shape[0] == 1 and cores[-1].shape[-1] == 1
    np.testing.assert_allclose(op.get_tensor_full(), ref, atol=1e-6 * abs(ref).max())
    low = TensorOperator(tensor=ref, only_diag=True).decompose(bond_dimension=1)
    assert all(c.shape[0] == 1 and c.shape[2] == 1 for c in low)
    with pytest.raises(ValueError):
        TensorOperator(tensor=ref, only_diag=True).decompose(rate=1.5)
    kin = construct_kinetic_mpo(prims)
    dense = mpo_to_dense(kin)
    t = [-0.5 * p.get_2nd_derivative_matrix_dvr() for p in prims]
    eye = np.eye(5)
    want = np.kron(np.kron(t[0], eye), eye) + np.kron(np.kron(eye, t[1]), eye) + np.kron(np.kron(eye, eye), t[2])
    np.testing.assert_allclose(dense, want, atol=1e-13)
    sop = construct_kinetic_operator(prims, forms="sop")
    h = TensorHamiltonian(3, potential=None, kinetic=[[sop]])
    np.testing.assert_allclose(mpo_to_dense(h.as_mpo([5, 5, 5])), want, atol=1e-13)


@pytest.mark.gpu
def test_reference_relax_operate_propagate_pins(tmp_path, monkeypatch):
    """tests/test_harmonic_dvr_func_full_mpssm_jax.py of the reference, the three functions in
    sequence, re-typed against the shell: relax -> operate(restart=True) -> propagate(restart=True)
    with the reference's own known answers."""
    from pytdscf_amd import BasInfo, Model, Simulator, TensorHamiltonian, units
    from pytdscf_amd.dvr_operator_cls import construct_fulldimensional, construct_kinetic_operator

    monkeypatch.chdir(tmp_path)
    prim_info = [_harmonic_prims()]
    basinfo = BasInfo(prim_info)

    def PES(q1, q2, q3):
        return ((1500 / units.au_in_cm1) ** 2 / 2 * q1**2 + (2000 / units.au_in_cm1) ** 2 / 2 * q2**2
                + (2500 / units.au_in_cm1) ** 2 / 2 * q3**2)

    def DMS(q1, q2, q3):
        return 0.1 * q1 + 0.1 * q2 + 0.1 * q3

    def ham():
        potential = [[construct_fulldimensional(dvr_prims=prim_info[0], func=PES)]]
        kinetic = [[construct_kinetic_operator(dvr_prims=prim_info[0])]]
        return TensorHamiltonian(ndof=3, potential=potential, kinetic=kinetic, decompose_type="SVD", rate=0.9999999, backend="hip")

    jobname = "harmonic_dvr_hip"
    model = Model(basinfo, {"hamiltonian": ham()})
    model.m_aux_max = 4
    ener_calc, wf = Simulator(jobname, model, backend="hip").relax(maxstep=3, stepsize=0.1)
    assert pytest.approx(ener_calc) == 0.013669005758739458

    dip = TensorHamiltonian(ndof=3, potential=[[construct_fulldimensional(dvr_prims=prim_info[0], func=DMS)]], kinetic=None,
                            decompose_type="SVD", rate=0.9999999, backend="hip")
    dip.coupleJ = [[1.0]]  # scalar term
    model = Model(basinfo, {"hamiltonian": dip})
    model.m_aux_max = 4
    norm_calc, wf = Simulator(jobname, model, backend="hip").operate(restart=True, maxstep=5)
    assert pytest.approx(norm_calc) == 1.6490051381599562

    model = Model(basinfo, {"hamiltonian": ham()})
    model.m_aux_max = 4
    ener_calc, wf = Simulator(jobname, model, backend="hip").propagate(maxstep=3, stepsize=0.1, restart=True)
    assert pytest.approx(ener_calc) == 0.019185297685193108


@pytest.mark.gpu
def test_autocorr_without_t2_trick(golden, tmp_path, monkeypatch):
    """Simulator(t2_trick=False): autocorr.dat holds <Psi(0)|Psi(t)> at t (not 2t).  (The
    reference's Properties asserts in this mode -- properties.py:62 -- so the pin is the oracle.)"""
    from pytdscf_amd import Exciton, Model, Simulator, units

    monkeypatch.chdir(tmp_path)
    g = golden("chain_lanczos.npz")
    n = int(g["nsite"])
    mpo = [g[f"mpo{i}"] for i in range(n)]
    init = [g[f"init{i}"] for i in range(n)]
    model = Model([Exciton(nstate=3) for _ in range(n)], operators={"hamiltonian": mpo}, bond_dim=6)
    model.init_HartreeProduct = [init]
    Simulator("not2", model, backend="hip", t2_trick=False).propagate(stepsize=0.05, maxstep=3)
    rows = [l.split() for l in open(tmp_path / "not2_prop" / "autocorr.dat").read().splitlines()[1:]]
    c0 = orc.canonicalize_site0(init)
    st = orc.OracleMPS([c.copy() for c in c0], mpo)
    for i, (t, a) in enumerate(rows):
        assert float(t) == pytest.approx(i * 0.05, abs=1e-9)
        assert complex(a.replace(" ", "")) == pytest.approx(orc.overlap(c0, st.cores), abs=2e-9)
        st.propagate(0.05 / units.au_in_fs)


@pytest.mark.gpu
def test_henon_heiles_reference_script_unchanged(tmp_path, monkeypatch):
    """tests/test_henon_heiles.py of the reference, NumPy case, typed as it is there (only the
    package name differs): construct_nMR_recursive + construct_kinetic_mpo + Model + Simulator."""
    import pytdscf_amd as pytdscf
    from pytdscf_amd import HarmonicOscillator as HO, Model, Simulator, units
    from pytdscf_amd.dvr_operator_cls import construct_kinetic_mpo, construct_nMR_recursive

    monkeypatch.chdir(tmp_path)
    assert pytdscf.units is units
    backend, ω, λ, f, N, m, Δt = "hip", 2000, 1.0e-03, 2, 5, 4, 0.001
    dvr_prims = [HO(N, ω) for _ in range(f)]
    ω_au = ω / units.au_in_cm1
    func = {}
    func[(0,)] = lambda Q1: pow(ω_au, 2) / 2 * Q1**2
    func[(0, 1)] = lambda Q1, Q2: λ * pow(ω_au, 3 / 2) * (Q1**2 * Q2)
    func[(1,)] = lambda Qf: pow(ω_au, 2) / 2 * Qf**2 - λ * pow(ω_au, 3 / 2) / 3 * Qf**3
    potential_mpo = construct_nMR_recursive(dvr_prims, nMR=2, func=func, rate=0.99999999999)
    kinetic_mpo = construct_kinetic_mpo(dvr_prims)
    operators = {"potential": potential_mpo, "kinetic": kinetic_mpo}
    model = Model(dvr_prims, operators=operators, bond_dim=m)
    vib_gs = [1.0] + [0.0] * (N - 1)
    vib_es = [0.0, 1.0] + [0.0] * (N - 2)
    model.init_weight_VIBSTATE = [[vib_es] + [vib_gs] * (f - 1)]
    simulator = Simulator(jobname="henon_heiles", model=model, backend=backend)
    ener_calc, wf = simulator.propagate(maxstep=3, stepsize=Δt)
    assert pytest.approx(ener_calc) == 0.018225341011652626


def _multistate_model(g):
    """The two-state model of tests/golden/multistate_chain.npz typed the way the generating
    script typed it for the reference: TensorHamiltonian(potential=[[..], [..]]) with one
    operator dictionary per (bra, ket) state pair, basis = one list per electronic state."""
    from pytdscf_amd import Exciton, Model, TensorHamiltonian, TensorOperator

    n, S = int(g["nsite"]), int(g["nstate"])
    d = g["init0_0"].shape[1]
    blk = {(i, j): [g[f"mpo{i}{j}_{p}"] for p in range(n)] for i in range(S) for j in range(S)}
    diag = lambda w: np.ascontiguousarray(np.einsum("ciit->cit", w))  # noqa: E731
    pot = [[None] * S for _ in range(S)]
    pot[0][0] = {tuple((p, p) for p in range(n)): TensorOperator(mpo=blk[(0, 0)])}
    pot[1][1] = {tuple(range(n)): TensorOperator(mpo=[diag(w) for w in blk[(1, 1)]], legs=tuple(range(n))),
                 (): float(g["coupleJ"][1, 1].real)}
    pot[0][1] = {((0, 0), (1, 1), (2, 2), 3, 4): TensorOperator(
        mpo=blk[(0, 1)][:3] + [diag(w) for w in blk[(0, 1)][3:]], legs=(0, 0, 1, 1, 2, 2, 3, 4))}
    pot[1][0] = {((0, 0), (1, 1), (2, 2), (3, 3), 4): TensorOperator(
        mpo=blk[(1, 0)][:4] + [diag(blk[(1, 0)][4])], legs=(0, 0, 1, 1, 2, 2, 3, 3, 4))}
    ham = TensorHamiltonian(n, potential=pot, kinetic=None, backend="hip")
    basis = [[Exciton(nstate=d) for _ in range(n)] for _ in range(S)]
    model = Model(basis, operators={"hamiltonian": ham}, bond_dim=int(g["bond_dim"]))
    model.init_HartreeProduct = [[g[f"init{s}_{p}"] for p in range(n)] for s in range(S)]
    model.init_weight_ESTATE = list(g["weights"])
    return model


def test_multistate_model_validation():
    from pytdscf_amd import Exciton, Model, TensorHamiltonian, TensorOperator

    core = np.zeros((1, 2, 2, 1))
    one = TensorHamiltonian(1, potential={((0, 0),): TensorOperator(mpo=[core])})
    assert one.nstate == 1 and one.coupleJ == [[0.0]]
    two = TensorHamiltonian(1, potential=[[{((0, 0),): TensorOperator(mpo=[core])}, {(): 0.2}], [{(): 0.2}, {(): 0.1}]])
    assert two.nstate == 2 and two.coupleJ == [[0.0, 0.2], [0.2, 0.1]]
    assert two.block_mpo(0, 1, [2]) is None and two.block_mpo(0, 0, [2])[0].shape == (1, 2, 2, 1)
    with pytest.raises(ValueError, match="square"):
        TensorHamiltonian(1, potential=[[{}, {}]])
    with pytest.raises(ValueError, match="electronic state"):
        Model([[Exciton(2)], [Exciton(2)]], {"hamiltonian": one}, bond_dim=2)
    with pytest.raises(NotImplementedError):
        Model([[Exciton(2)], [Exciton(3)]], {"hamiltonian": two}, bond_dim=2)
    m = Model([[Exciton(2)], [Exciton(2)]], {"hamiltonian": two}, bond_dim=2)
    assert m.get_nstate() == 2 and m.estate_weights() == [1.0, 0.0]
    m.init_weight_ESTATE = [1.0, 3.0]
    assert m.estate_weights() == [0.25, 0.75]


@pytest.mark.gpu
def test_multistate_script_on_gpu(golden, tmp_path, monkeypatch):
    """Two electronic states through the shell: energies, populations.dat, restart."""
    from pytdscf_amd import Simulator

    monkeypatch.chdir(tmp_path)
    g = golden("multistate_chain.npz")
    sim = Simulator("ms", _multistate_model(g), backend="hip")
    ener, wf = sim.propagate(stepsize=0.05, maxstep=3)
    assert ener == pytest.approx(float(g["n3_energy_last"]), abs=1e-10)
    np.testing.assert_allclose(wf.pop_states(), g["n3_pops"], atol=1e-10)
    assert wf.norm() == pytest.approx(1.0, abs=1e-12)
    assert abs(wf.autocorr() - complex(g["n3_autocorr"])) < 1e-10
    assert wf.expectation("hamiltonian") == pytest.approx(float(g["n3_energy_final"]), abs=1e-10)
    rows = np.loadtxt(tmp_path / "ms_prop" / "populations.dat")
    assert rows.shape == (3, 3)
    np.testing.assert_allclose(rows[0, 1:], g["weights"] / g["weights"].sum(), atol=1e-9)
    np.testing.assert_allclose(rows[1, 1:], g["n1_pops"], atol=1e-9)
    fin = wf.get_mps()
    for s in range(2):
        for p in range(int(g["nsite"])):
            np.testing.assert_allclose(fin[s][p], g[f"n3_final{s}_{p}"], atol=1e-9)
    # restart: 1 + 2 steps equal 3 steps
    sim2 = Simulator("ms2", _multistate_model(g), backend="hip")
    sim2.propagate(stepsize=0.05, maxstep=1, savefile_ext="_a")
    e2, wf2 = sim2.propagate(stepsize=0.05, maxstep=2, restart=True, loadfile_ext="_a")
    assert e2 == pytest.approx(float(g["n3_energy_last"]), abs=1e-10)
    np.testing.assert_allclose(wf2.pop_states(), g["n3_pops"], atol=1e-10)
    # imaginary time
    e_r, wf_r = Simulator("ms3", _multistate_model(g), backend="hip").relax(stepsize=0.2, maxstep=3, improved=False)
    assert e_r == pytest.approx(float(g["relax_n3_energy_last"]), abs=1e-10)
    np.testing.assert_allclose(wf_r.pop_states(), g["relax_n3_pops"], atol=1e-10)
    e_i, _ = Simulator("ms4", _multistate_model(g), backend="hip").relax(maxstep=3)  # improved=True
    assert e_i == pytest.approx(float(g["improved_n3_energy_last"]), abs=1e-9)
    with pytest.raises(NotImplementedError):
        Simulator("ms5", _multistate_model(g), backend="hip").propagate(maxstep=1, adaptive=True)
    # relax -> operate -> propagate through the checkpoint files, like the single-state workflow
    sim6 = Simulator("ms6", _multistate_model(g), backend="hip")
    nrm, wf_o = sim6.operate(maxstep=10)
    assert nrm == pytest.approx(float(g["operate_n10_norm"]), rel=1e-10)
    np.testing.assert_allclose(wf_o.pop_states(), g["operate_n10_pops"], atol=1e-10)
    assert (tmp_path / "wf_ms6_operate.pkl").exists()
    e6, wf6 = sim6.propagate(stepsize=0.05, maxstep=2, restart=True)  # loadfile_ext="_operate"
    assert wf6.norm() == pytest.approx(1.0, abs=1e-12)
    np.testing.assert_allclose(sum(wf6.pop_states()), 1.0, atol=1e-12)


def test_grid_dvr_bases_match_reference(golden):
    """Sine / Exponential DVR primitives (pytdscf/basis/sin.py, exponential.py) against vectors
    produced by the reference's classes."""
    from pytdscf_amd import Exponential, Sine

    g = golden("basis_dvr.npz")
    cases = {"sine_t": Sine(7, 3.0, x0=0.5, units="angstrom", include_terminal=True),
             "sine_n": Sine(6, 4.0, x0=-1.0, units="bohr", include_terminal=False),
             "exp": Exponential(7, 2.0 * np.pi, x0=0.1)}
    for tag, b in cases.items():
        assert len(b) == b.nprim == len(g[f"{tag}_grids"])
        np.testing.assert_allclose(b.get_grids(), g[f"{tag}_grids"], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(list(b), g[f"{tag}_grids"], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(b.get_unitary(), g[f"{tag}_unitary"], rtol=0, atol=1e-13)
        np.testing.assert_allclose(b.get_sqrt_weights(), g[f"{tag}_sqrt_weights"], rtol=1e-13)
        for name, fn in (("d1_dvr", b.get_1st_derivative_matrix_dvr), ("d2_dvr", b.get_2nd_derivative_matrix_dvr),
                         ("d1_fbr", b.get_1st_derivative_matrix_fbr), ("d2_fbr", b.get_2nd_derivative_matrix_fbr)):
            ref = g[f"{tag}_{name}"]
            np.testing.assert_allclose(fn(), ref, rtol=0, atol=1e-11 * max(1.0, np.abs(ref).max()), err_msg=f"{tag} {name}")
        np.testing.assert_allclose([b.fbr_func(2, x) for x in b.get_grids()], g[f"{tag}_fbr2_at_grid"], rtol=0, atol=1e-13)
        np.testing.assert_allclose([b.dvr_func(3, x) for x in b.get_grids()], g[f"{tag}_dvr3_at_grid"], rtol=0, atol=1e-12)
        if tag != "exp":
            np.testing.assert_allclose(b.get_pos_rep_matrix(), g[f"{tag}_pos"], rtol=0, atol=1e-15)
    with pytest.raises(ValueError):
        Exponential(6, 1.0)
    # the kinetic-energy builders take these bases like the harmonic-oscillator DVR
    from pytdscf_amd.dvr_operator_cls import construct_kinetic_mpo

    kin = construct_kinetic_mpo([cases["sine_n"], cases["sine_t"]])
    assert len(kin) == 2


def test_spectra_known_answer(tmp_path):
    """The reference's own known-answer test for the auto-correlation -> spectrum chain
    (tests/test_spectra.py) on its data file, and a round trip through the shell's writer format."""
    import os

    from pytdscf_amd import spectra

    dat = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "autocorr.dat")
    time, autocorr = spectra.load_autocorr(dat)
    freq, intensity = spectra.ifft_autocorr(time, autocorr)
    assert max(intensity) == pytest.approx(28860.651565826236)
    assert freq[np.argmax(intensity)] == pytest.approx(2684.0796620397296)
    spectra.export_spectrum(freq, intensity, str(tmp_path / "s.dat"))
    back = np.loadtxt(tmp_path / "s.dat")
    assert back.shape == (len(freq), 2)
    # the format Simulator.propagate writes is what load_autocorr reads
    with open(tmp_path / "a.dat", "w") as f:
        f.write("# time [fs]\t auto-correlation\n")
        for t, a in ((0.0, 1.0 + 0j), (0.1, 0.9 - 0.1j), (0.2, 0.7 - 0.2j), (0.3, 0.5 - 0.2j), (0.4, 0.3 - 0.1j)):
            f.write(f"{t:6.9f}\t{a.real: 6.9f}{a.imag:+6.9f}j\n")
    t2, a2 = spectra.load_autocorr(str(tmp_path / "a.dat"))
    assert a2[1] == 0.9 - 0.1j and t2[-1] == 0.4


def test_lindblad_to_kraus():
    """Kraus tensor of one Lindblad step: reproduces exp(D dt), is trace preserving, and gives the
    same channel as the set stored in the reference-generated Kraus fixture."""
    import scipy.linalg

    from pytdscf_amd.kraus import lindblad_to_kraus

    rng = np.random.default_rng(0)
    d, dt = 3, 0.3
    Ls = [0.4 * (rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))), np.diag([0.0, 0.5, 1.0])]
    B = lindblad_to_kraus(list(Ls), dt)
    assert B.shape[1:] == (d, d) and B.dtype == np.complex128
    np.testing.assert_allclose(sum(b.conj().T @ b for b in B), np.eye(d), atol=1e-12)  # trace preserving
    rho = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
    rho = rho @ rho.conj().T
    rho /= np.trace(rho)
    # independent check: integrate the Lindblad equation as a linear ODE on vec(rho)
    eye = np.eye(d)
    D = sum(np.kron(L, L.conj()) - 0.5 * (np.kron(L.conj().T @ L, eye) + np.kron(eye, (L.conj().T @ L).T)) for L in Ls)
    exact = (scipy.linalg.expm(D * dt) @ rho.reshape(-1)).reshape(d, d)
    np.testing.assert_allclose(sum(b @ rho @ b.conj().T for b in B), exact, atol=1e-12)
    with pytest.raises(ValueError):
        lindblad_to_kraus([np.zeros((2, 3))], 0.1)


def test_lindblad_to_kraus_same_channel_as_reference(golden):
    """The Kraus tensor the reference's lindblad_to_kraus produced for the Kraus fixtures (stored as
    "B") and ours describe the same channel (the sets differ by a unitary mixing of q at most)."""
    from pytdscf_amd.kraus import lindblad_to_kraus

    g = golden("kraus_single.npz")
    rng_k = np.random.default_rng(31337)  # the generating script's stream (tests/golden/make_golden.py)
    d = int(g["d"])
    Lops = [0.4 * rng_k.standard_normal((d, d)), 0.3 * rng_k.standard_normal((d, d))]
    B = lindblad_to_kraus(Lops, 0.5)
    ch = lambda Bs: sum(np.kron(b, b.conj()) for b in Bs)  # noqa: E731
    assert B.shape == g["B"].shape
    np.testing.assert_allclose(ch(B), ch(g["B"]), atol=1e-12)


@pytest.mark.parametrize("rate,J", [(1.0, [2, 2, 2, 2, 2, 2]), (0.999999999999, [1, 2, 3, 1, 2, 3])])
def test_tensor_dict_to_mpo_reference_test(rate, J):
    """tests/test_compress_mpo.py of the reference on its own data file (H2CO grid tensors up to
    two-mode terms): the value of the MPO at a grid point equals the sum of the tensors there."""
    import os
    import pickle

    from pytdscf_amd.dvr_operator_cls import tensor_dict_to_mpo

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h2co.tensor"), "rb") as f:
        tensor_dict = pickle.load(f)
    J = np.array(J)
    mpo = tensor_dict_to_mpo(tensor_dict, rate=rate)
    val_tensor = sum(t[tuple(J[np.ix_(k)])] for k, t in tensor_dict.items() if k != ())
    r = mpo[0][0, J[0], :]
    for i, W in enumerate(mpo[1:], 1):
        r = r @ W[:, J[i], :]
    assert abs(val_tensor - r[0].real) < 1.0e-10
    assert [w.ndim for w in mpo] == [3] * 6 and max(w.shape[-1] for w in mpo) <= 14
    # every grid point, and a visibly lossy rate still returns a valid chain
    full = np.zeros((5,) * 6)
    for k, t in tensor_dict.items():
        if k != ():
            full = full + t.reshape([5 if i in k else 1 for i in range(6)])
    dense = mpo[0]
    for w in mpo[1:]:
        dense = np.tensordot(dense, w, axes=(dense.ndim - 1, 0))
    np.testing.assert_allclose(dense.reshape((5,) * 6), full, atol=1e-9)
    lossy = tensor_dict_to_mpo(tensor_dict, rate=0.99)
    assert max(w.shape[-1] for w in lossy) < max(w.shape[-1] for w in mpo)
    with pytest.raises(ValueError):
        tensor_dict_to_mpo(tensor_dict, rate=1.5)


def _h2o_model(g, nprim=6, bond_dim=4, weights=None):
    """tests/test_anharmonic_fbr_mpssm_propagate_np.py of the reference: force constants of its
    H2O potential (stored in the fixture), HO-eigenfunction primitives, polynomial Hamiltonian."""
    import math

    from pytdscf_amd import BasInfo, Model, PrimBas_HO, read_potential_nMR, units

    k_orig = {tuple(int(x) for x in key if x): float(v) for key, v in zip(g["h2o_keys"], g["h2o_vals"])}
    prim = [[PrimBas_HO(0.0, math.sqrt(k_orig[(i, i)]) * units.au_in_cm1, nprim) for i in (1, 2, 3)]]
    model = Model(BasInfo(prim), {"hamiltonian": read_potential_nMR(k_orig)}, bond_dim=bond_dim)
    if weights is not None:
        model.init_weight_VIBSTATE = weights
    return model


EXCITED = [[[0.0, 1.0, 0.0, 0.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0, 0.0, 0.0], [0.6, 0.8, 0.0, 0.0, 0.0, 0.0]]]


def test_polynomial_hamiltonian_as_mpo_cpu(golden):
    """Sum-of-products Hamiltonians of the MPS standard method become exact MPOs.  The fixture holds
    the reference's SoP runs (its MPSCoefSoP contractions).  Energies agree to 1e-12 (both reference
    pins included); the dynamics agree to ~1e-7: the reference's own SoP and MPO sweeps differ from
    each other by that much on the same operator (infidelity 4e-9 after one step at D = 4, checked in
    the development container; its MPO sweep on the converted operator equals the oracle to 2e-16 and
    is the one closer to the exact propagation)."""
    from pytdscf_amd import BasInfo, Model, PolynomialHamiltonian, PrimBas_HO
    from pytdscf_amd.operators import mpo_to_dense

    g = golden("polynomial_sm.npz")
    # BASELINE configs[0]: two harmonic modes in their own eigenbasis
    bi = BasInfo([[PrimBas_HO(0.0, 1500, 8), PrimBas_HO(0.0, 2000, 8)]])
    ham = PolynomialHamiltonian(ndof=2)
    ham.set_HO_potential(bi)
    m = Model(bi, {"hamiltonian": ham})
    dense = mpo_to_dense(m.hamiltonian.as_mpo(m.dims))
    np.testing.assert_allclose(dense, np.diag(np.diag(dense)), atol=1e-15)
    assert dense[0, 0].real == pytest.approx(float(g["harmonic_energy"]), abs=1e-15)
    assert float(g["harmonic_energy"]) == pytest.approx(0.007973586692598029)
    # anharmonic H2O
    for tag, wts in (("gs", None), ("ex", EXCITED)):
        for n in (1, 5):
            model = _h2o_model(g, weights=wts)
            st = orc.OracleMPS(orc.canonicalize_site0(model.initial_cores()), model.hamiltonian.as_mpo(model.dims),
                               shift=model.hamiltonian.coupleJ[0][0])
            for _ in range(n):
                e = st.expectation()
                st.propagate(float(g["dt_au"]))
            k = f"h2o_{tag}_n{n}"
            assert e.real == pytest.approx(float(g[f"{k}_energy_last"]), abs=1e-12)
            assert abs(st.autocorr() - complex(g[f"{k}_autocorr"])) < 2e-6
            assert abs(abs(orc.overlap([g[f"{k}_final{p}"] for p in range(3)], st.cores)) - 1) < 1e-6
    st = orc.OracleMPS(orc.canonicalize_site0(_h2o_model(g).initial_cores()), _h2o_model(g).hamiltonian.as_mpo([6, 6, 6]))
    assert st.expectation().real == pytest.approx(0.021360262338234466, abs=1e-12)  # the reference test's pin


@pytest.mark.gpu
def test_polynomial_scripts_on_gpu(golden, tmp_path, monkeypatch):
    """The two reference tests with polynomial Hamiltonians, typed against the shell."""
    from pytdscf_amd import BasInfo, Model, PolynomialHamiltonian, PrimBas_HO, Simulator

    monkeypatch.chdir(tmp_path)
    g = golden("polynomial_sm.npz")
    basinfo = BasInfo([[PrimBas_HO(0.0, 1500, 8), PrimBas_HO(0.0, 2000, 8)]])
    hamiltonian = PolynomialHamiltonian(ndof=2)
    hamiltonian.set_HO_potential(basinfo)
    simulator = Simulator("harmonic_fbr_sm", Model(basinfo, {"hamiltonian": hamiltonian}), ci_type="standard-method", backend="numpy")
    ener_calc, wf = simulator.propagate(maxstep=1)
    assert pytest.approx(ener_calc) == 0.007973586692598029
    simulator = Simulator("anharmonic_fbr_propagate_sm", _h2o_model(g), backend="numpy")
    ener_calc, wf = simulator.propagate(maxstep=2)
    assert pytest.approx(ener_calc) == 0.021360262338234466
    ener, wf = Simulator("h2o_dyn", _h2o_model(g, weights=EXCITED), backend="hip").propagate(stepsize=0.2, maxstep=5)
    assert ener == pytest.approx(float(g["h2o_ex_n5_energy_last"]), abs=1e-11)
    assert abs(wf.autocorr() - complex(g["h2o_ex_n5_autocorr"])) < 2e-6  # SoP vs MPO sweep of the reference, see the CPU test
    assert abs(abs(orc.overlap([g[f"h2o_ex_n5_final{p}"] for p in range(3)], wf.get_mps())) - 1) < 1e-6
    assert abs(wf.norm() - 1) < 1e-12
    # against the oracle (= the reference's MPO sweep) on the same converted operator: the usual bar
    model = _h2o_model(g, weights=EXCITED)
    st = orc.OracleMPS(orc.canonicalize_site0(model.initial_cores()), model.hamiltonian.as_mpo(model.dims))
    for _ in range(5):
        st.propagate(float(g["dt_au"]))
    assert abs(wf.autocorr() - st.autocorr()) < 1e-10
    assert abs(abs(orc.overlap(st.cores, wf.get_mps())) - 1) < 1e-10


@pytest.mark.gpu
def test_dipole_operate_script_on_gpu(tmp_path, monkeypatch):
    """tests/test_sample_CS_ovlp_np.py of the reference up to the coherent-state sampling: a dipole
    from ``read_potential_nMR(dipole_emu=...)`` applied to the HO ground state (its norm is the
    reference's known answer), then configuration coefficients of the result."""
    from pytdscf_amd import BasInfo, Model, PrimBas_HO, Simulator
    from pytdscf_amd.hamiltonian_cls import read_potential_nMR

    monkeypatch.chdir(tmp_path)
    prim_info = [[PrimBas_HO(0.0, 1500, 5), PrimBas_HO(0.0, 2000, 5), PrimBas_HO(0.0, 2500, 5)]]
    basinfo = BasInfo(prim_info)
    mu = {(0,): [1 / 30, 1 / 30, 1 / 30], (1,): [1 / 30, 1 / 30, 1 / 30], (2,): [1 / 30, 1 / 30, 1 / 30]}
    dipole = read_potential_nMR(potential_emu=None, dipole_emu=mu)
    model = Model(basinfo, {"hamiltonian": dipole})
    model.m_aux_max = 4
    simulator = Simulator("coherent_sample_FBR", model, backend="numpy")
    norm, wf = simulator.operate(maxstep=10, restart=False)
    assert pytest.approx(norm) == 1.3111895155460684
    # mu|000> = 0.1 sum_i q_i |000>: one quantum in one mode, amplitudes 0.1 / sqrt(2 w_i) / norm
    w = [p.freq_au for p in prim_info[0]]
    amps = [abs(wf.ci_coef.get_CI_coef_state(J=tuple(1 if k == i else 0 for k in range(3)))) for i in range(3)]
    np.testing.assert_allclose(amps, [0.1 / np.sqrt(2 * x) / norm for x in w], rtol=1e-9)
    assert abs(wf.ci_coef.get_CI_coef_state(J=(0, 0, 0))) < 1e-12
    dense = wf.get_mps()[0]
    for c in wf.get_mps()[1:]:
        dense = np.tensordot(dense, c, axes=(dense.ndim - 1, 0))
    t = [np.arange(1, 6) * (0.3 + 0.1j * k) for k in range(3)]
    ref = np.einsum("ijk,i,j,k->", dense.reshape(5, 5, 5), *t)
    assert abs(wf.ci_coef.get_CI_coef_state(trans_arrays=t) - ref) < 1e-12


@pytest.mark.gpu
def test_lvc_polynomial_two_states_on_gpu(tmp_path, monkeypatch):
    """Linear vibronic coupling typed like tests/test_LVC_propagate_np.py (PolynomialHamiltonian.set_LVC,
    coupleJ matrix, init_weight_ESTATE) but with the MPS standard method and one primitive basis for
    both states: the sum of products becomes per-pair MPO blocks of the multi-state engine.  Checked
    against the multi-state oracle on the same blocks and against exact propagation."""
    import scipy.linalg

    from pytdscf_amd import BasInfo, Model, PolynomialHamiltonian, PrimBas_HO, Simulator, units
    from pytdscf_amd.operators import mpo_to_dense

    monkeypatch.chdir(tmp_path)
    freqs = [1000, 2000, 3000]
    s0 = [PrimBas_HO(0.0, f, 5) for f in freqs]
    basinfo = BasInfo([s0, s0])
    ham = PolynomialHamiltonian(basinfo.get_ndof(), basinfo.get_nstate())
    ham.coupleJ = [[0, -0.004], [-0.004, 0.007]]
    lam = {(0, 1): {0: 0.002, 1: 0.002, 2: 0.002}, (1, 0): {0: 0.002, 1: 0.002, 2: 0.002}}
    ham.set_LVC(basinfo, lam)
    model = Model(basinfo, {"hamiltonian": ham}, bond_dim=5)
    model.init_weight_ESTATE = [1.0, 0.0]
    assert model.hamiltonian.nstate == 2 and model.hamiltonian.coupleJ[1][1] == pytest.approx(0.007)
    ener, wf = Simulator("LVC_sm", model, backend="hip").propagate(maxstep=3, stepsize=0.05)
    assert ener == pytest.approx(0.5 * sum(freqs) / units.au_in_cm1, abs=1e-12)  # <H> of |S0, 000>: the zero-point energy
    assert ener == pytest.approx(0.013669005758738601, rel=1e-9)  # = the value tests/test_LVC_propagate_np.py pins
    # multi-state oracle on the same blocks
    blocks = [[model.hamiltonian.block_mpo(i, j, model.dims) for j in range(2)] for i in range(2)]
    raw = [model.initial_cores(s) for s in range(2)]
    z = orc.canonicalize_site0(raw[1], 1.0)
    z[0] = z[0] * 0.0
    st = orc.OracleMultiMPS([orc.canonicalize_site0(raw[0], 1.0), z], blocks, model.hamiltonian.coupleJ)
    for _ in range(3):
        st.propagate(0.05 / units.au_in_fs)
    np.testing.assert_allclose(wf.pop_states(), st.pop_states(), atol=1e-10)
    assert abs(wf.autocorr() - st.autocorr()) < 1e-10
    # exact propagation in the 2 x 125-dimensional product basis (bond dimension 5 is the full rank here)
    n = 125
    H = np.zeros((2 * n, 2 * n), dtype=complex)
    for i in range(2):
        for j in range(2):
            blk = mpo_to_dense(blocks[i][j]) if blocks[i][j] is not None else 0.0
            H[i * n:(i + 1) * n, j * n:(j + 1) * n] = blk + complex(model.hamiltonian.coupleJ[i][j]) * np.eye(n)
    assert np.abs(H - H.conj().T).max() < 1e-15
    v0 = np.zeros(2 * n, dtype=complex)
    v0[0] = 1.0
    vt = scipy.linalg.expm(-1j * H * 3 * 0.05 / units.au_in_fs) @ v0
    np.testing.assert_allclose(wf.pop_states(), [np.linalg.norm(vt[:n]) ** 2, np.linalg.norm(vt[n:]) ** 2], atol=1e-7)
    assert wf.pop_states()[1] > 1e-6


def test_compat_aliases_resolve_reference_import_paths():
    """``pytdscf_amd.compat.install()``: the import lines of scripts written for the reference
    resolve to this package (no GPU needed to import)."""
    import subprocess
    import sys as _sys

    code = """
import sys
sys.path.insert(0, %r)
import pytdscf_amd.compat as c
names = c.install()
import pytdscf
from pytdscf import BasInfo, Model, Simulator, units, Exciton, Boson, TensorHamiltonian, TensorOperator, construct_kinetic_mpo
from pytdscf.basis import PrimBas_HO, Sine, Exponential, HarmonicOscillator
from pytdscf.basis._primints_cls import PrimBas_HO as P2
from pytdscf.hamiltonian_cls import PolynomialHamiltonian, TensorHamiltonian as T2, read_potential_nMR
from pytdscf.model_cls import BasInfo as B2, Model as M2
from pytdscf.simulator_cls import Simulator as S2
from pytdscf.dvr_operator_cls import TensorOperator as TO2, construct_nMR_recursive, construct_kinetic_operator, construct_fulldimensional
from pytdscf.kraus import lindblad_to_kraus
from pytdscf.util import read_nc
from pytdscf.wavefunction import WFunc
import pytdscf.spectra
from discvar import HarmonicOscillator as HO
import pytdscf_amd
assert Model is pytdscf_amd.Model is M2 and Simulator is S2 and P2 is PrimBas_HO and HO is pytdscf_amd.HarmonicOscillator
assert pytdscf.spectra.ifft_autocorr is pytdscf_amd.spectra.ifft_autocorr and pytdscf.__version__
c.uninstall()
assert "pytdscf" not in sys.modules and "discvar" not in sys.modules
print("ALIASES OK", len(names))
""" % ROOT_DIR
    r = subprocess.run([_sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ALIASES OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_compat_runs_a_reference_style_script(tmp_path):
    """A script with the reference's import lines (``import pytdscf``, ``from discvar import ...``,
    ``backend="numpy"``) runs unedited through ``python -m pytdscf_amd.compat`` and reproduces the
    Hénon–Heiles pin of the reference's test."""
    import subprocess
    import sys as _sys

    script = tmp_path / "hh_reference_style.py"
    script.write_text('''
from discvar import HarmonicOscillator as HO
from pytdscf import units
from pytdscf.dvr_operator_cls import construct_kinetic_mpo, construct_nMR_recursive
from pytdscf.model_cls import Model
from pytdscf.simulator_cls import Simulator

w, lam, f, N, m, dt = 2000, 1.0e-03, 2, 5, 4, 0.001
dvr_prims = [HO(N, w) for _ in range(f)]
w_au = w / units.au_in_cm1
func = {(0,): lambda Q1: pow(w_au, 2) / 2 * Q1**2,
        (0, 1): lambda Q1, Q2: lam * pow(w_au, 3 / 2) * (Q1**2 * Q2),
        (1,): lambda Qf: pow(w_au, 2) / 2 * Qf**2 - lam * pow(w_au, 3 / 2) / 3 * Qf**3}
operators = {"potential": construct_nMR_recursive(dvr_prims, nMR=2, func=func, rate=0.99999999999),
             "kinetic": construct_kinetic_mpo(dvr_prims)}
model = Model(dvr_prims, operators=operators, bond_dim=m)
model.init_weight_VIBSTATE = [[[0.0, 1.0] + [0.0] * (N - 2)] + [[1.0] + [0.0] * (N - 1)] * (f - 1)]
ener, wf = Simulator(jobname="henon_heiles", model=model, backend="numpy").propagate(maxstep=3, stepsize=dt)
print(f"ENERGY {ener:.15f}")
''')
    r = subprocess.run([_sys.executable, "-m", "pytdscf_amd.compat", str(script)], capture_output=True, text=True, timeout=300,
                       cwd=str(tmp_path), env=dict(os.environ, PYTHONPATH=ROOT_DIR + os.pathsep + os.environ.get("PYTHONPATH", "")))
    assert r.returncode == 0, r.stdout + r.stderr
    ener = float([l for l in r.stdout.splitlines() if l.startswith("ENERGY")][0].split()[1])
    assert ener == pytest.approx(0.018225341011652626)
