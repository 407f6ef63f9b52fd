#!/usr/bin/env python3
"""Development-container only: run the reference (through the same third-party stand-ins as
make_golden.py) and the NumPy oracle side by side on feature COMBINATIONS that no committed
fixture covers, and print the differences.  Nothing here is imported by the tests; it documents
how the oracle was checked beyond the fixtures.

    python tests/golden/crosscheck_reference.py
"""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import make_golden as mg  # noqa: E402


def main():
    if not os.path.isdir(mg.REF):
        sys.exit("reference not present")
    tmp = tempfile.mkdtemp(prefix="xchk_")
    stubs = os.path.join(tmp, "stubs")
    mg.make_third_party_stubs(stubs)
    sys.path.insert(0, mg.REF)
    sys.path.insert(0, stubs)
    os.chdir(tmp)
    import numpy as np
    import pytdscf  # noqa: F401
    from pytdscf import Model, Simulator, units
    from pytdscf import _helper as helper
    from pytdscf.basis import Exciton
    from pytdscf.dvr_operator_cls import TensorOperator
    from pytdscf.hamiltonian_cls import TensorHamiltonian

    from oracle import tdvp_oracle as orc
    from pytdscf_amd.mps import product_state_cores
    from pytdscf_amd.operators import merge_operator_terms

    gold = lambda f: np.load(os.path.join(HERE, f))  # noqa: E731

    def ham_of(mpo, cj=0.0):
        n = len(mpo)
        h = TensorHamiltonian(n, potential=[[{tuple((i, i) for i in range(n)): TensorOperator(mpo=[w.copy() for w in mpo])}]],
                              kinetic=None, backend="numpy")
        h.coupleJ = [[cj]]
        return h

    def report(tag, *diffs):
        print(f"{tag:42s}", "  ".join(f"{d:.2e}" for d in diffs), flush=True)

    g = gold("chain_lanczos.npz")
    n = int(g["nsite"])
    mpo = [g[f"mpo{i}"] for i in range(n)]
    init = [g[f"init{i}"] for i in range(n)]
    dt = float(g["dt_au"])
    basis = [Exciton(nstate=3) for _ in range(n)]

    # scalar term in real time, relaxation and improved relaxation
    m = Model(basis, operators={"hamiltonian": ham_of(mpo, 0.3)}, bond_dim=6)
    m.init_HartreeProduct = [[np.array(c) for c in init]]
    ener, wf = Simulator("x", m, backend="numpy", verbose=0).propagate(stepsize=0.05, maxstep=3)
    st = orc.OracleMPS(orc.canonicalize_site0(init), mpo, shift=0.3)
    for _ in range(3):
        e = st.expectation()
        st.propagate(dt)
    report("propagate + coupleJ: energy, autocorr", abs(ener - e.real), abs(wf._ints_wf_ovlp_mpssm(wf.ci_coef, conj=False) - st.autocorr()))
    for improved in (False, True):
        m = Model(basis, operators={"hamiltonian": ham_of(mpo, 0.4)}, bond_dim=6)
        m.init_HartreeProduct = [[np.array(c) for c in init]]
        ener, wf = Simulator("x", m, backend="numpy", verbose=0).relax(stepsize=0.2, maxstep=3, improved=improved)
        st = orc.OracleMPS(orc.canonicalize_site0(init), mpo, shift=0.4, relax="improved" if improved else True)
        for _ in range(3):
            e = st.expectation()
            st.propagate(0.2 / units.au_in_fs)
        ref = [np.array(s.data) for s in wf.ci_coef.superblock_states[0]]
        report(f"relax(improved={improved}) + coupleJ: energy, 1-|ovlp|", abs(ener - e.real), abs(abs(orc.overlap(ref, st.cores)) - 1))

    # observables in Hilbert space: one-site 4-leg, two-site with a gap and a diagonal core
    rng = np.random.default_rng(3)
    A = rng.standard_normal((3, 3)) + 1j * rng.standard_normal((3, 3))
    A = A + A.conj().T
    Bd = rng.standard_normal(3)
    C = rng.standard_normal((3, 3))
    C = C + C.T
    o1 = TensorHamiltonian(n, potential=[[{((2, 2),): TensorOperator(mpo=[A[None, :, :, None]], legs=(2, 2))}]], kinetic=None, backend="numpy")
    o2 = TensorHamiltonian(n, potential=[[{((1, 1), 4): TensorOperator(mpo=[C[None, :, :, None], Bd[None, :, None]], legs=(1, 1, 4))}]],
                           kinetic=None, backend="numpy")
    m = Model(basis, operators={"hamiltonian": [w.copy() for w in mpo], "o1": o1, "o2": o2}, bond_dim=6)
    m.init_HartreeProduct = [[np.array(c) for c in init]]
    _, wf = Simulator("x", m, backend="numpy", verbose=0).propagate(stepsize=0.05, maxstep=2)
    st = orc.OracleMPS(orc.canonicalize_site0(init), mpo)
    for _ in range(2):
        st.propagate(dt)
    m1 = merge_operator_terms([([A[None, :, :, None]], [2])], [3] * n)
    m2 = merge_operator_terms([([C[None, :, :, None], Bd[None, :, None]], [1, 4])], [3] * n)
    report("observables (1-site, 2-site with gap)", abs(wf.expectation(m.observables["o1"]) - st.expectation(m1).real),
           abs(wf.expectation(m.observables["o2"]) - st.expectation(m2).real))

    # adaptive + coupleJ (real / complex with Arnoldi), gates + adaptive
    ga, gg = gold("adaptive_chain.npz"), gold("gate_chain.npz")
    na = int(ga["nsite"])
    mpoa = [ga[f"mpo{i}"] for i in range(na)]
    inita = [ga[f"init{i}"] for i in range(na)]
    akw = dict(adaptive=True, adaptive_Dmax=7, adaptive_dD=1, adaptive_p_proj=1e-8)
    okw = dict(adaptive=True, Dmax=7, dD=1, p_proj=1e-8)
    for cj in (0.7, 0.7 - 0.2j):
        cplx = isinstance(cj, complex)
        m = Model(basis, operators={"hamiltonian": ham_of(mpoa, cj)}, bond_dim=2)
        m.init_HartreeProduct = [[np.array(c) for c in inita]]
        helper._Debug.niter_krylov.clear()
        kw = dict(integrator="arnoldi", conserve_norm=False) if cplx else {}
        ener, wf = Simulator("x", m, backend="numpy", verbose=0).propagate(stepsize=0.05, maxstep=3, **akw, **kw)
        st = orc.OracleMPS(orc.canonicalize_site0(inita), mpoa, shift=cj, integrator="arnoldi" if cplx else "lanczos",
                           conserve_norm=not cplx, **okw)
        for _ in range(3):
            e = st.expectation()
            st.propagate(float(ga["dt_au"]))
        bd = [s.data.shape[2] for s in wf.ci_coef.superblock_states[0][:-1]]
        assert bd == [c.shape[2] for c in st.cores[:-1]], (bd, [c.shape[2] for c in st.cores[:-1]])
        report(f"adaptive + coupleJ={cj}: energy, autocorr", abs(ener - e.real), abs(wf._ints_wf_ovlp_mpssm(wf.ci_coef, conj=False) - st.autocorr()))
    gate = TensorHamiltonian(ndof=na, potential=[[{((1, 1),): TensorOperator(mpo=[gg["U1"][None, :, :, None]], legs=(1, 1)),
                                                   (4,): TensorOperator(mpo=[gg["U4"][None, :, None]], legs=(4,))}]], kinetic=None, backend="numpy")
    m = Model(basis, operators={"hamiltonian": ham_of(mpoa)}, bond_dim=2, one_gate_to_apply=gate)
    m.init_HartreeProduct = [[np.array(c) for c in inita]]
    helper._Debug.niter_krylov.clear()
    ener, wf = Simulator("x", m, backend="numpy", verbose=0).propagate(stepsize=0.05, maxstep=3, **akw)
    st = orc.OracleMPS(orc.canonicalize_site0(inita), mpoa, gates={1: gg["U1"], 4: gg["U4"]}, **okw)
    for _ in range(3):
        e = st.expectation()
        st.propagate(float(ga["dt_au"]))
    assert [s.data.shape[2] for s in wf.ci_coef.superblock_states[0][:-1]] == [c.shape[2] for c in st.cores[:-1]]
    report("gates + adaptive: energy, autocorr", abs(ener - e.real), abs(wf._ints_wf_ovlp_mpssm(wf.ci_coef, conj=False) - st.autocorr()))

    # Liouville space + adaptive
    gl = gold("chain_liouville.npz")
    nl = int(gl["nsite"])
    lmpo = [gl[f"mpo{i}"] for i in range(nl)]
    rhos = [gl[f"rho{i}"] for i in range(nl)]
    m = Model([Exciton(nstate=4) for _ in range(nl)], operators={"hamiltonian": [w.copy() for w in lmpo]}, bond_dim=2, space="liouville")
    m.init_HartreeProduct = [rhos]
    helper._Debug.niter_krylov.clear()
    _, wf = Simulator("x", m, backend="numpy", verbose=0).propagate(stepsize=0.02, maxstep=3, integrator="arnoldi", autocorr=False, energy=False,
                                                                     adaptive=True, adaptive_Dmax=8, adaptive_dD=2, adaptive_p_proj=1e-9)
    cores = orc.canonicalize_site0(product_state_cores(rhos, 2, space="liouville"), scale=None)
    st = orc.OracleMPS(cores, lmpo, integrator="arnoldi", conserve_norm=False, adaptive=True, Dmax=8, dD=2, p_proj=1e-9)
    for _ in range(3):
        st.propagate(float(gl["dt_au"]))
    assert [s.data.shape[2] for s in wf.ci_coef.superblock_states[0][:-1]] == [c.shape[2] for c in st.cores[:-1]]
    pt_ref = np.array(wf.get_reduced_densities((0, 0, 2))[0])
    report("Liouville + adaptive: norm, partial trace", abs(wf.norm() - st.norm()), np.abs(pt_ref - orc.liouville_partial_trace(st.cores, (0, 0, 2))).max())

    # several electronic states with SHARED operator keys and off-diagonal scalar terms.  The
    # reference's NumPy backend caches its contraction expressions by operator key only
    # (_contraction.py:1050-1055, :1165-1172; multiplyK likewise), so state pairs that share a key,
    # a "summed" block or a non-identity "ovlp" block reuse the first pair's constants -- its JAX
    # path contracts without that cache.  DIAGNOSTIC: the two cached wrappers are called with
    # key=None here (the un-cached branch of the same functions) to compare the intended
    # arithmetic; the committed fixture (multistate_chain.npz) needs no such step.
    import math

    from pytdscf import _contraction as C_

    gm = gold("multistate_chain.npz")
    nm, dm = int(gm["nsite"]), 3
    blk = {(i, j): [gm[f"mpo{i}{j}_{p}"] for p in range(nm)] for i in range(2) for j in range(2)}
    raw = [[gm[f"init{s_}_{p}"] for p in range(nm)] for s_ in range(2)]
    cjm = [[0.02, 0.1 - 0.03j], [0.1 + 0.03j, 0.05]]
    key4 = tuple((i, i) for i in range(nm))
    pot = [[{key4: TensorOperator(mpo=[w.copy() for w in blk[(i, j)]]), (): cjm[i][j]} for j in range(2)] for i in range(2)]
    h2 = TensorHamiltonian(nm, potential=pot, kinetic=None, backend="numpy")
    m = Model([[Exciton(nstate=dm) for _ in range(nm)] for _ in range(2)], operators={"hamiltonian": h2}, bond_dim=int(gm["bond_dim"]))
    m.init_HartreeProduct = [[np.array(c) for c in st_] for st_ in raw]
    m.init_weight_ESTATE = [0.8, 0.6]
    h_, k_ = C_.multiplyH_MPS_direct_MPO._op_lcr_dot, C_.multiplyK_MPS_direct_MPO._op_lr_dot
    C_.multiplyH_MPS_direct_MPO._op_lcr_dot = lambda self, a, b, c, t, key=None: h_(self, a, b, c, t, key=None)
    C_.multiplyK_MPS_direct_MPO._op_lr_dot = lambda self, a, b, t, key=None: k_(self, a, b, t, key=None)
    try:
        helper._Debug.niter_krylov.clear()
        ener, wf = Simulator("x", m, backend="numpy", verbose=0).propagate(stepsize=0.05, maxstep=3)
    finally:
        C_.multiplyH_MPS_direct_MPO._op_lcr_dot, C_.multiplyK_MPS_direct_MPO._op_lr_dot = h_, k_
    w_ = np.array([0.8, 0.6]) / 1.4
    st = orc.OracleMultiMPS([orc.canonicalize_site0(raw[s_], math.sqrt(w_[s_])) for s_ in range(2)],
                            [[blk[(0, 0)], blk[(0, 1)]], [blk[(1, 0)], blk[(1, 1)]]], cjm)
    for _ in range(3):
        e = st.expectation()
        st.propagate(float(gm["dt_au"]))
    fin = [[np.array(c.data) for c in s_] for s_ in wf.ci_coef.superblock_states]
    report("two states, shared keys + off-diagonal coupleJ (cache off): energy, pops, tensors", abs(ener - e.real),
           np.abs(np.array(wf.pop_states()) - np.array(st.pop_states())).max(),
           max(np.abs(a - b).max() for x, y in zip(fin, st.cores) for a, b in zip(x, y)))
    # wire format: a checkpoint written by pytdscf_amd (host-side snapshot classes), converted on the reference's side
    # (pytdscf_amd.checkpoint.to_reference_checkpoint) and RESTARTED by the reference (simulator_cls.py:501-507): its next
    # step against the oracle's
    import dill

    from pytdscf_amd import checkpoint as ck

    st = orc.OracleMPS(orc.canonicalize_site0(init), mpo)
    for _ in range(2):
        st.propagate(dt)
    snap = ck.SavedWFunc(ck.SavedMPSCoef([[ck.SavedSiteCoef(c, "Psi" if i == 0 else "B", i) for i, c in enumerate(st.cores)]], "hilbert"))
    with open("wf_amd.pkl", "wb") as f:
        dill.dump(snap, f)
    m = Model(basis, operators={"hamiltonian": [w.copy() for w in mpo]}, bond_dim=6)
    m.init_HartreeProduct = [[np.array(c) for c in init]]
    sim = Simulator("x", m, backend="numpy", verbose=0)
    ck.to_reference_checkpoint("wf_amd.pkl", "wf_x_amd.pkl", m)
    ener, wf = sim.propagate(stepsize=0.05, maxstep=1, restart=True, loadfile_ext="_amd")
    e = st.expectation()
    st.propagate(dt)
    ref = [np.array(s.data) for s in wf.ci_coef.superblock_states[0]]
    report("restart of the reference from a pytdscf_amd checkpoint: energy, 1-|ovlp|, autocorr", abs(ener - e.real),
           abs(abs(orc.overlap(ref, st.cores)) - 1), abs(wf._ints_wf_ovlp_mpssm(wf.ci_coef, conj=False) - st.autocorr()))
    print("all differences should be at rounding level (<= 1e-12)")


if __name__ == "__main__":
    main()
