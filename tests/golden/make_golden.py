#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE.

Run only in the development container (``/root/reference`` must exist):

    python tests/golden/make_golden.py

The reference (PyTDSCF v1.3.3) is imported from ``/root/reference`` with its
NumPy backend.  Its *third-party* imports that are absent from this image
(jax, loguru, discvar, opt_einsum, polars, netCDF4 -- none of them part of the
reference's own source) are replaced by throw-away stand-ins written to a
temporary directory; see SURVEY.md section 8(c).  No reference source is copied:
the fixtures hold inputs (tensors, time step) and the reference's outputs.

Every fixture is a small ``.npz``; the parity tests load them without the
reference being present.
"""

from __future__ import annotations

import os
import sys
import tempfile
import textwrap

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def _write(path, text):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(textwrap.dedent(text))


def make_third_party_stubs(root: str) -> None:
    """Stand-ins for third-party packages only (never for reference code)."""
    _write(
        f"{root}/jax/__init__.py",
        """
        class Array:  # never instantiated on the NumPy path
            pass
        def jit(f=None, **kw):
            if f is None:
                return lambda g: g
            return f
        class _Cfg:
            def update(self, *a, **k): pass
        config = _Cfg()
        from . import numpy, scipy  # noqa
        """,
    )
    _write(
        f"{root}/jax/numpy/__init__.py",
        """
        import numpy as _np
        from jax import Array as ndarray
        complex128 = _np.complex128
        float64 = _np.float64
        """,
    )
    _write(f"{root}/jax/scipy/__init__.py", "from . import linalg\n")
    _write(f"{root}/jax/scipy/linalg.py", "")
    _write(
        f"{root}/loguru/__init__.py",
        """
        class _L:
            def bind(self, **k): return self
            def add(self, *a, **k): return 0
            def remove(self, *a, **k): pass
            def debug(self, *a, **k): pass
            def info(self, *a, **k): pass
            def warning(self, *a, **k): pass
            def error(self, *a, **k): pass
            def critical(self, *a, **k): pass
        logger = _L()
        """,
    )
    # discvar v0.0.2 is the upstream home of the HO-DVR classes; the reference
    # ships in-tree twins (pytdscf/basis/__init__.py:1-4) which we re-export.
    _write(
        f"{root}/discvar/__init__.py",
        """
        from pytdscf.basis.ho import HarmonicOscillator, PrimBas_HO
        from pytdscf.basis.abc import DVRPrimitivesMixin
        from pytdscf.basis import ho
        from pytdscf.basis import abc
        """,
    )
    _write(f"{root}/discvar/abc.py", "from pytdscf.basis.abc import DVRPrimitivesMixin\n")
    _write(
        f"{root}/opt_einsum/__init__.py",
        """
        import numpy as np
        _cache = {}
        def _path(sub, ops):
            key = (sub, tuple(o.shape for o in ops))
            if key not in _cache:
                _cache[key] = np.einsum_path(sub, *ops, optimize=("optimal", 2**44))[0]
            return _cache[key]
        def contract(sub, *ops, **kw):
            ops = [np.asarray(o) for o in ops]
            return np.einsum(sub, *ops, optimize=_path(sub, ops))
        class _Expr:
            def __init__(self, sub, args, constants):
                self.sub = sub; self.args = list(args); self.constants = list(constants)
            def __call__(self, *ops, **kw):
                full = list(self.args); it = iter(ops)
                for i in range(len(full)):
                    if i not in self.constants:
                        full[i] = next(it)
                return contract(self.sub, *full)
        def contract_expression(sub, *args, constants=(), **kw):
            return _Expr(sub, args, constants)
        """,
    )
    _write(f"{root}/polars/__init__.py", "")
    _write(
        f"{root}/netCDF4/__init__.py",
        """
        class Dataset:
            def __init__(self, *a, **k): raise RuntimeError("netCDF4 absent")
        """,
    )
    _write(
        f"{root}/pytdscf-1.3.3.dist-info/METADATA",
        "Metadata-Version: 2.1\nName: pytdscf\nVersion: 1.3.3\n",
    )


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not present; fixtures can only be regenerated in the dev container")
    tmp = tempfile.mkdtemp(prefix="golden_")
    stubs = os.path.join(tmp, "stubs")
    make_third_party_stubs(stubs)
    sys.path.insert(0, REF)
    sys.path.insert(0, stubs)
    sys.path.insert(0, REPO)
    os.chdir(tmp)  # the reference writes {jobname}_prop/ and wf_*.pkl into cwd

    import numpy as np
    import pytdscf  # noqa: F401  (must come before anything imports discvar)
    from pytdscf import Model, Simulator, units
    from pytdscf import _helper as helper
    from pytdscf._const_cls import const
    from pytdscf._contraction import (
        SplitStack,
        contract_with_site_mpo,
        multiplyH_MPS_direct_MPO,
        multiplyK_MPS_direct_MPO,
    )
    from pytdscf import _integrator
    from pytdscf._mpo_cls import OperatorCore
    from pytdscf._site_cls import SiteCoef
    from pytdscf.basis import Exciton
    from pytdscf.dvr_operator_cls import (
        TensorOperator,
        construct_kinetic_mpo,
        construct_nMR_recursive,
    )
    from pytdscf.hamiltonian_cls import TensorHamiltonian
    from discvar import HarmonicOscillator as HO

    from oracle import tdvp_oracle as orc  # only for the synthetic input builders

    au_in_fs = float(units.au_in_fs)
    rng = np.random.default_rng(20260503)

    def crandn(*shape):
        return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)

    def save(name, **kw):
        np.savez_compressed(os.path.join(HERE, name), **kw)
        print("wrote", name, {k: np.shape(v) for k, v in kw.items()})

    # ------------------------------------------------------------------ unit
    const.use_jax = False  # the global run-type flags the unit-level calls read
    const.conserve_norm = True
    const.verbose = 0
    # a3: environment updates through contract_with_site_mpo
    Dl, d, Dr, Ml, Mr = 5, 3, 4, 3, 2
    A = crandn(Dl, d, Dr)
    Lb = crandn(Dl, Ml, Dl)
    W = crandn(Ml, d, d, Mr)
    core = OperatorCore([0, 1, 2], ((0, 0), (1, 1), (2, 2)), 1, W, "numpy")
    outA = contract_with_site_mpo(SiteCoef(A, "A", 1), SiteCoef(A, "A", 1), Lb, core)
    B = crandn(Dr, d, Dl)  # (D_l', d, D_r') with D_r' = Dl to reuse Lb as right block
    Rb = crandn(Dl, Mr, Dl)
    outB = contract_with_site_mpo(SiteCoef(B, "B", 1), SiteCoef(B, "B", 1), Rb, core)
    save("unit_env.npz", A=A, L=Lb, W=W, outA=outA, B=B, R=Rb, outB=outB)

    # a4 / a5: H_eff and K_eff applies through _op_lcr_dot / _op_lr_dot
    psi = crandn(Dl, d, Dr)
    R3 = crandn(Dr, Mr, Dr)
    mh = multiplyH_MPS_direct_MPO.__new__(multiplyH_MPS_direct_MPO)
    mh._op_lcr_dot_cached = {}
    sig = mh._op_lcr_dot(Lb, core, R3, psi, key="k")
    sv = crandn(Dl, Dr)
    Rk = crandn(Dr, Ml, Dr)
    mk = multiplyK_MPS_direct_MPO.__new__(multiplyK_MPS_direct_MPO)
    mk._op_lr_dot_cached = {}
    sigk = mk._op_lr_dot(Lb, Rk, sv, key="k")
    save("unit_apply.npz", L=Lb, W=W, R=R3, psi=psi, sigma=sig, Rk=Rk, sval=sv, sigma_k=sigk)

    # a6 / a7: local propagators on a dense operator through the SplitStack seam
    class Dense(SplitStack):
        def __init__(self, mat, shape):
            super().__init__(shape)
            self.mat = mat

        def dot(self, states):
            return [(self.mat @ states[0].reshape(-1)).reshape(states[0].shape)]

    n = 36
    G = crandn(n, n)
    Hh = (G + G.conj().T) / 2
    Hn = Hh - 0.3j * np.diag(rng.random(n))  # non-Hermitian (absorbing)
    x0 = crandn(3, 4, 3)
    x0 /= np.linalg.norm(x0)
    out = {}
    for tag, integ, mat, cn, scale in [
        ("lan_dt001", "lanczos", Hh, True, -0.01j),
        ("lan_dt01", "lanczos", Hh, True, -0.1j),
        ("lan_real", "lanczos", Hh, False, -0.05),
        ("arn_dt001", "arnoldi", Hn, False, -0.01j),
        ("arn_dt01", "arnoldi", Hn, False, -0.1j),
    ]:
        const.conserve_norm = cn
        const.use_jax = False
        helper._Debug.niter_krylov.clear()
        helper._Debug.site_now = 0
        fn = (
            _integrator.short_iterative_lanczos
            if integ == "lanczos"
            else _integrator.short_iterative_arnoldi
        )
        xin = x0 * (1.0 if cn else 1.7)
        y1 = fn(scale, Dense(mat, x0.shape), [xin.copy()], 1e-9)[0]
        k1 = helper._Debug.niter_krylov[0]
        # second call: warm-up from the stored iteration count
        y2 = fn(scale, Dense(mat, x0.shape), [y1.copy()], 1e-9)[0]
        k2 = helper._Debug.niter_krylov[0]
        out[tag + "_in"] = xin
        out[tag + "_y1"] = y1
        out[tag + "_y2"] = y2
        out[tag + "_k"] = np.array([k1, k2])
        out[tag + "_scale"] = np.array(scale, dtype=np.complex128)
        out[tag + "_cn"] = np.array(cn)
    save("unit_krylov.npz", Hh=Hh, Hn=Hn, **out)

    # a8: gauge moves
    psi_g = crandn(4, 3, 5)
    a_site, sA = SiteCoef(psi_g.copy(), "Psi", 1).gauge_trf("Psi2Asigma")
    b_site, sB = SiteCoef(psi_g.copy(), "Psi", 1).gauge_trf("Psi2sigmaB")
    save(
        "unit_gauge.npz", psi=psi_g, A=np.array(a_site.data), sigA=sA, B=np.array(b_site.data), sigB=sB
    )

    # bond truncation by SVD: truncate_sigvec (_site_cls.py:586-690)
    from pytdscf._site_cls import truncate_sigvec

    rng_t = np.random.default_rng(77)  # own stream: the fixtures below keep their inputs

    def crandn_t(*shape):
        return rng_t.standard_normal(shape) + 1j * rng_t.standard_normal(shape)

    psi_t = crandn_t(5, 3, 7)
    a_t, sig_t = SiteCoef(psi_t.copy(), "Psi", 1).gauge_trf("Psi2Asigma")
    # a decaying spectrum so that the cumulative-weight criterion actually truncates
    sig_t = sig_t * (0.5 ** np.arange(7))[None, :]
    q_t, _ = np.linalg.qr(crandn_t(12, 7))
    b_t = np.ascontiguousarray(q_t.T.reshape(7, 3, 4))
    out_t = {}
    for tag, pp in (("p1e-2", 1e-2), ("p1e-6", 1e-6), ("p0", 0.0)):
        A2, s2, B2 = truncate_sigvec(SiteCoef(np.array(a_t.data), "A", 1), sig_t.copy(), SiteCoef(b_t.copy(), "B", 2), pp)
        out_t[f"{tag}_sig"] = np.array(s2)
        out_t[f"{tag}_two_site"] = np.einsum("ajk,kl,lmr->ajmr", np.array(A2.data), np.array(s2), np.array(B2.data))
    save("unit_truncate.npz", A=np.array(a_t.data), sigma=sig_t, B=b_t, **out_t)

    # ------------------------------------------------------------ end-to-end
    def run_ref(basis, operators, cores, D, dt_fs, nstep, **kw):
        """Reference Simulator.propagate from explicit (full-rank) cores."""
        model = Model(basis, operators=operators, bond_dim=D)
        model.init_HartreeProduct = [[np.array(c) for c in cores]]
        helper._Debug.niter_krylov.clear()
        sim = Simulator("gold", model, backend="numpy", verbose=0)
        ener, wf = sim.propagate(stepsize=dt_fs, maxstep=nstep, **kw)
        fin = [np.array(s.data) for s in wf.ci_coef.superblock_states[0]]
        return dict(
            energy_last=np.array(ener),
            final=fin,
            autocorr=np.array(wf._ints_wf_ovlp_mpssm(wf.ci_coef, conj=False)),
            norm=np.array(wf.norm()),
            energy_final=np.array(wf.expectation(model.hamiltonian)),
            krylov=np.array(
                [helper._Debug.niter_krylov[i] for i in range(len(cores))]
            ),
        )

    def pack(prefix, res):
        o = {}
        for k, v in res.items():
            if k == "final":
                for i, c in enumerate(v):
                    o[f"{prefix}_final{i}"] = c
            else:
                o[f"{prefix}_{k}"] = v
        return o

    # (i) synthetic Hermitian chain, Lanczos, full-rank start (SURVEY 8d inputs)
    L, d, M, D = 6, 3, 4, 6
    mpo = orc.synthetic_mpo(L, d, M, seed=0)
    bd = orc.bond_dims([d] * L, D)
    cores = [crandn(dl, d, dr) for (dl, dr) in bd]
    basis = [Exciton(nstate=d) for _ in range(L)]
    dt = 0.05
    o = {f"mpo{i}": w for i, w in enumerate(mpo)}
    o.update({f"init{i}": c for i, c in enumerate(cores)})
    for n in (1, 4):
        o.update(pack(f"n{n}", run_ref(basis, {"hamiltonian": [w.copy() for w in mpo]}, cores, D, dt, n)))
    save("chain_lanczos.npz", dt_au=np.array(dt / au_in_fs), nsite=np.array(L), **o)

    # (ii) non-Hermitian chain, Arnoldi, conserve_norm=False
    mpo_n = [w.copy() for w in mpo]
    for p in range(L):
        g = 0.02 * rng.random(d)
        # -i * gamma_j |j><j| on the "local" slot of the MPO
        row = 0
        col = mpo_n[p].shape[3] - 1
        mpo_n[p][row, :, :, col] += -1j * np.diag(g)
    o = {f"mpo{i}": w for i, w in enumerate(mpo_n)}
    o.update({f"init{i}": c for i, c in enumerate(cores)})
    for n in (1, 3):
        o.update(
            pack(
                f"n{n}",
                run_ref(
                    basis,
                    {"hamiltonian": [w.copy() for w in mpo_n]},
                    cores,
                    D,
                    dt,
                    n,
                    integrator="arnoldi",
                    conserve_norm=False,
                ),
            )
        )
    save("chain_arnoldi.npz", dt_au=np.array(dt / au_in_fs), nsite=np.array(L), **o)

    # (ii-b) imaginary-time relaxation (doRelax=True, exp(-H dt/2) + renormalisation,
    # _mps_cls.py:1086-1094, :1162-1169) on the Hermitian chain
    model = Model(basis, operators={"hamiltonian": [w.copy() for w in mpo]}, bond_dim=D)
    model.init_HartreeProduct = [[np.array(c) for c in cores]]
    o = {f"mpo{i}": w for i, w in enumerate(mpo)}
    o.update({f"init{i}": c for i, c in enumerate(cores)})
    for n in (1, 5):
        helper._Debug.niter_krylov.clear()
        sim = Simulator("gold_relax", model, backend="numpy", verbose=0)
        ener, wf = sim.relax(stepsize=0.2, maxstep=n, improved=False)
        o[f"n{n}_energy_last"] = np.array(ener)
        o[f"n{n}_energy_final"] = np.array(wf.expectation(model.hamiltonian))
        o[f"n{n}_norm"] = np.array(wf.norm())
        o[f"n{n}_krylov"] = np.array([helper._Debug.niter_krylov[i] for i in range(L)])
        for i, s in enumerate(wf.ci_coef.superblock_states[0]):
            o[f"n{n}_final{i}"] = np.array(s.data)
    save("chain_relax.npz", dt_au=np.array(0.2 / au_in_fs), nsite=np.array(L), **o)

    # (ii-c) improved relaxation (doRelax="improved": matrix_diagonalize_lanczos per site,
    # no bond propagation, _mps_cls.py:1078-1084, :1159-1160)
    o = {f"mpo{i}": w for i, w in enumerate(mpo)}
    o.update({f"init{i}": c for i, c in enumerate(cores)})
    for n in (1, 3):
        helper._Debug.niter_krylov.clear()
        sim = Simulator("gold_irelax", model, backend="numpy", verbose=0)
        ener, wf = sim.relax(stepsize=0.2, maxstep=n, improved=True)
        o[f"n{n}_energy_last"] = np.array(ener)
        o[f"n{n}_energy_final"] = np.array(wf.expectation(model.hamiltonian))
        o[f"n{n}_norm"] = np.array(wf.norm())
        o[f"n{n}_krylov"] = np.array([helper._Debug.niter_krylov[i] for i in range(L)])
        for i, s in enumerate(wf.ci_coef.superblock_states[0]):
            o[f"n{n}_final{i}"] = np.array(s.data)
    save("chain_improved_relax.npz", nsite=np.array(L), **o)

    # (ii-d) Liouville space (space="Liouville": vectorised density matrix, d = n^2 per
    # site, trace-normalised product start, Arnoldi, conserve_norm forced False,
    # _const_cls.py:219-224): trace expectation (_exp_liouville, _mps_cls.py:3769-3838)
    # and partial traces (get_partial_trace, :1438-1510)
    Ll, Dl_ = 5, 8
    lmpo = orc.synthetic_liouvillian_mpo(Ll, 16, seed=0, gamma=0.05)
    lbasis = [Exciton(nstate=4) for _ in range(Ll)]
    sz = np.diag([1.0, -1.0]).reshape(1, 2, 2, 1).astype(np.complex128)
    sx = np.array([[0.0, 1.0], [1.0, 0.0]]).reshape(1, 2, 2, 1).astype(np.complex128)
    # two-site observable sz_1 sx_3 (identity fill-in on site 2) and a one-site sz_2
    obs1 = TensorHamiltonian(ndof=Ll, potential=[[{((2, 2),): TensorOperator(mpo=[sz], legs=(2, 2))}]], kinetic=None, backend="numpy")
    obs2 = TensorHamiltonian(ndof=Ll, potential=[[{((1, 1), (3, 3)): TensorOperator(mpo=[sz, sx], legs=(1, 1, 3, 3))}]], kinetic=None, backend="numpy")
    lmodel = Model(lbasis, operators={"hamiltonian": [w.copy() for w in lmpo], "sz2": obs1, "sz1sx3": obs2}, bond_dim=Dl_, space="liouville")
    rhos = []
    for p in range(Ll):
        G = crandn(2, 2)
        rho = G @ G.conj().T
        rhos.append((rho / np.trace(rho)).reshape(-1))
    lmodel.init_HartreeProduct = [rhos]
    o = {f"mpo{i}": w for i, w in enumerate(lmpo)}
    o.update({f"rho{i}": r for i, r in enumerate(rhos)})
    o["sz"] = sz
    o["sx"] = sx
    for n in (1, 3):
        helper._Debug.niter_krylov.clear()
        sim = Simulator("gold_liou", lmodel, backend="numpy", verbose=0)
        _, wf = sim.propagate(stepsize=0.02, maxstep=n, integrator="arnoldi", autocorr=False, energy=False)
        o[f"n{n}_norm"] = np.array(wf.norm())
        o[f"n{n}_sz2"] = np.array(wf.expectation(lmodel.observables["sz2"]))
        o[f"n{n}_sz1sx3"] = np.array(wf.expectation(lmodel.observables["sz1sx3"]))
        for tag, legs in (("pt2", (0, 0, 2)), ("pt04", (2, 0, 0, 0, 2)), ("pt1d", (0, 1)), ("pt13", (0, 2, 0, 1))):
            o[f"n{n}_{tag}"] = np.array(wf.get_reduced_densities(legs)[0])
        o[f"n{n}_krylov"] = np.array([helper._Debug.niter_krylov[i] for i in range(Ll)])
        for i, s_ in enumerate(wf.ci_coef.superblock_states[0]):
            o[f"n{n}_final{i}"] = np.array(s_.data)
    save("chain_liouville.npz", dt_au=np.array(0.02 / au_in_fs), nsite=np.array(Ll), bond_dim=np.array(Dl_), **o)

    # (iii) the reference's own exciton pin (tests/test_exiciton_propagate.py):
    # potential = diagonal 3-leg cores + one 4-leg core, kinetic on sites 0-2 only.
    au_in_cm1 = float(units.au_in_cm1)
    freqs = [1000, 2000, 3000]
    omega2 = [(f / au_in_cm1) ** 2 for f in freqs]
    nprim = 8
    prim = [HO(nprim, f, units="cm-1") for f in freqs] + [Exciton(nstate=2, names=["S0", "S1"])]
    dE, J, lamb, kappa = 0.01, 0.001, 0.0001, 0.0001
    W0 = np.zeros((1, nprim, 3), dtype=np.complex128)
    W1 = np.zeros((3, nprim, 4), dtype=np.complex128)
    W2 = np.zeros((4, nprim, 3), dtype=np.complex128)
    W3 = np.zeros((3, 2, 2, 1), dtype=np.complex128)
    q1 = [np.array(ho.get_grids()) for ho in prim[:3]]
    q2 = [q * q for q in q1]
    one = [np.ones_like(q) for q in q1]
    a = prim[3].get_annihilation_matrix()
    ad = prim[3].get_creation_matrix()
    W0[0, :, 0] = one[0]
    W0[0, :, 1] = q1[0]
    W0[0, :, 2] = omega2[0] / 2 * q2[0]
    W1[0, :, 0] = J * one[1] + lamb * q1[1]
    W1[0, :, 1] = one[1]
    W1[0, :, 2] = kappa * q1[1] + omega2[1] ** 2 / 2 * q2[1]
    W1[0, :, 3] = omega2[1] / 2 * q2[1]
    W1[1, :, 0] = lamb * one[1]
    W1[1, :, 2] = kappa * one[1]
    W1[2, :, 2] = one[1]
    W1[2, :, 3] = one[1]
    W2[0, :, 2] = one[2]
    W2[1, :, 0] = dE * one[2] + kappa * q1[2] + omega2[2] / 2 * q2[2]
    W2[1, :, 1] = omega2[2] / 2 * q2[2]
    W2[1, :, 2] = lamb * q1[2]
    W2[2, :, 0] = one[2]
    W2[3, :, 1] = one[2]
    W3[0, :, :, 0] = ad @ a
    W3[1, :, :, 0] = a @ ad
    W3[2, :, :, 0] = ad + a
    pot = [W0, W1, W2, W3]
    kin = []
    for idof in range(3):
        t = prim[idof].get_2nd_derivative_matrix_dvr() / 2
        if idof == 0:
            c = np.zeros((1, nprim, nprim, 2), dtype=np.complex128)
            c[0, :, :, 0] = t
            c[0, :, :, 1] = np.eye(nprim)
        elif idof == 2:
            c = np.zeros((2, nprim, nprim, 1), dtype=np.complex128)
            c[0, :, :, 0] = np.eye(nprim)
            c[1, :, :, 0] = t
        else:
            c = np.zeros((2, nprim, nprim, 2), dtype=np.complex128)
            c[0, :, :, 0] = np.eye(nprim)
            c[1, :, :, 1] = np.eye(nprim)
            c[0, :, :, 1] = t
        kin.append(c)

    def exciton_model():
        ham = TensorHamiltonian(
            ndof=4,
            potential=[[{(0, 1, 2, (3, 3)): TensorOperator(mpo=[w.copy() for w in pot], legs=(0, 1, 2, 3, 3))}]],
            kinetic=[[{((0, 0), (1, 1), (2, 2)): TensorOperator(mpo=[w.copy() for w in kin], legs=(0, 0, 1, 1, 2, 2))}]],
            backend="numpy",
        )
        m = Model(prim, {"hamiltonian": ham}, bond_dim=2)
        m.init_HartreeProduct = [
            [ho.get_unitary()[0].tolist() for ho in prim[:3]] + [np.array([0.0, 1.0]).tolist()]
        ]
        return m

    o = {f"pot{i}": w for i, w in enumerate(pot)}
    o.update({f"kin{i}": w for i, w in enumerate(kin)})
    o.update({f"w{i}": np.array(ho.get_unitary()[0]) for i, ho in enumerate(prim[:3])})
    for n in (19, 20):
        helper._Debug.niter_krylov.clear()
        sim = Simulator("gold_exc", exciton_model(), backend="numpy", verbose=0)
        ener, wf = sim.propagate(stepsize=0.1, maxstep=n)
        o[f"n{n}_energy_last"] = np.array(ener)
        # key (3, 3) -> remain_legs (0, 0, 0, 2), properties.py:69-82
        rd = wf.get_reduced_densities((0, 0, 0, 2))
        o[f"n{n}_rdm33"] = np.array(rd[0])
        # further keys of the reference test: (0, 0) -> (2,), (0, 0, 3, 3) -> (2, 0, 0, 2),
        # and diagonal-only forms (one leg per site)
        for tag, legs in (("rdm00", (2,)), ("rdm0033", (2, 0, 0, 2)), ("rdm1", (0, 1)), ("rdm013", (1, 2, 0, 1))):
            o[f"n{n}_{tag}"] = np.array(wf.get_reduced_densities(legs)[0])
        for i, s in enumerate(wf.ci_coef.superblock_states[0]):
            o[f"n{n}_final{i}"] = np.array(s.data)
    # the reference's own known answers (tests/test_exiciton_propagate.py:178-184)
    assert abs(o["n20_energy_last"] - 0.010000180312707298) < 1e-6 * 0.01
    pin = np.array(
        [
            [1.86417721e-02 + 1.60379680e-20j, 2.87367863e-02 - 6.91095824e-02j],
            [2.87367863e-02 + 6.91095824e-02j, 9.81358228e-01 - 7.40721885e-18j],
        ]
    )
    np.testing.assert_allclose(o["n19_rdm33"], pin, atol=1e-9)
    save("exciton.npz", dt_au=np.array(0.1 / au_in_fs), ref_pin_rdm33=pin,
         ref_pin_energy=np.array(0.010000180312707298), **o)

    # (iv) the reference's Henon-Heiles NumPy pin (tests/test_henon_heiles.py:23)
    w_cm, lam, f, N, m, dt_h = 2000, 1.0e-03, 2, 5, 4, 0.001
    dvr = [HO(N, w_cm) for _ in range(f)]
    w_au = w_cm / au_in_cm1
    func = {
        (0,): lambda Q1: pow(w_au, 2) / 2 * Q1**2,
        (0, 1): lambda Q1, Q2: lam * pow(w_au, 3 / 2) * (Q1**2 * Q2),
        (1,): lambda Qf: pow(w_au, 2) / 2 * Qf**2 - lam * pow(w_au, 3 / 2) / 3 * Qf**3,
    }
    pmpo = construct_nMR_recursive(dvr, nMR=2, func=func, rate=0.99999999999)
    kmpo = construct_kinetic_mpo(dvr)
    o = {f"pot{i}": np.array(w) for i, w in enumerate(pmpo)}
    o.update({f"kin{i}": np.array(w) for i, w in enumerate(kmpo)})
    o.update({f"unitary{i}": np.array(h.get_unitary()) for i, h in enumerate(dvr)})
    model = Model(dvr, operators={"potential": [np.array(w) for w in pmpo], "kinetic": [np.array(w) for w in kmpo]}, bond_dim=m)
    model.init_weight_VIBSTATE = [[[0.0, 1.0] + [0.0] * (N - 2)] + [[1.0] + [0.0] * (N - 1)] * (f - 1)]
    helper._Debug.niter_krylov.clear()
    sim = Simulator("gold_hh", model, backend="numpy", verbose=0)
    ener, wf = sim.propagate(maxstep=3, stepsize=dt_h)
    assert abs(ener - 0.018225341011652626) < 1e-6 * 0.02
    o["n3_energy_last"] = np.array(ener)
    for i, s in enumerate(wf.ci_coef.superblock_states[0]):
        o[f"n3_final{i}"] = np.array(s.data)
    save("henon_heiles.npz", dt_au=np.array(dt_h / au_in_fs), ref_pin_energy=np.array(0.018225341011652626),
         au_in_fs=np.array(au_in_fs), au_in_cm1=np.array(au_in_cm1), **o)

    # (v) adaptive bond dimension (a1TDVP): Simulator.propagate(adaptive=True, ...)
    # (_mps_cls.py:863-987, :1921-2286; the reference's own run is tests/test_a1tdvp.py,
    # which only checks that it executes).  Own RNG stream: earlier fixtures keep their inputs.
    rng_a = np.random.default_rng(4242)

    def crandn_a(*shape):
        return rng_a.standard_normal(shape) + 1j * rng_a.standard_normal(shape)

    def run_adaptive(model, nsite, n, dt_fs, **kw):
        helper._Debug.niter_krylov.clear()
        sim = Simulator("gold_adapt", model, backend="numpy", verbose=0)
        ener, wf = sim.propagate(stepsize=dt_fs, maxstep=n, adaptive=True, **kw)
        res = {
            f"n{n}_energy_last": np.array(ener),
            f"n{n}_energy_final": np.array(wf.expectation(model.hamiltonian)),
            f"n{n}_norm": np.array(wf.norm()),
            f"n{n}_autocorr": np.array(wf._ints_wf_ovlp_mpssm(wf.ci_coef, conj=False)),
            f"n{n}_krylov": np.array([helper._Debug.niter_krylov[i] for i in range(nsite)]),
            f"n{n}_bonddim": np.array([s.data.shape[2] for s in wf.ci_coef.superblock_states[0][:-1]]),
        }
        for i, s_ in enumerate(wf.ci_coef.superblock_states[0]):
            res[f"n{n}_final{i}"] = np.array(s_.data)
        return res

    # (v-a) synthetic chain from low-rank random cores
    La, da, Ma, Da0 = 6, 3, 4, 2
    mpo_a = orc.synthetic_mpo(La, da, Ma, seed=3)
    bd_a = orc.bond_dims([da] * La, Da0)
    cores_a = [crandn_a(dl, da, dr) for (dl, dr) in bd_a]
    basis_a = [Exciton(nstate=da) for _ in range(La)]
    model_a = Model(basis_a, operators={"hamiltonian": [w.copy() for w in mpo_a]}, bond_dim=Da0)
    model_a.init_HartreeProduct = [[np.array(c) for c in cores_a]]
    o = {f"mpo{i}": w for i, w in enumerate(mpo_a)}
    o.update({f"init{i}": c for i, c in enumerate(cores_a)})
    akw = dict(adaptive_Dmax=7, adaptive_dD=1, adaptive_p_proj=1.0e-8)
    for n in (1, 3):
        o.update(run_adaptive(model_a, La, n, 0.05, **akw))
    save("adaptive_chain.npz", dt_au=np.array(0.05 / au_in_fs), nsite=np.array(La), bond_dim0=np.array(Da0),
         Dmax=np.array(7), dD=np.array(1), p_proj=np.array(1.0e-8), **o)

    # (v-b) the model of tests/test_a1tdvp.py (exciton + 3 modes, lambda = 1e-3) from the
    # bond-dimension-1 Hartree product
    pot_b = [w.copy() for w in pot]
    lamb_b = 0.001
    pot_b[1][0, :, 0] = J * one[1] + lamb_b * q1[1]
    pot_b[1][1, :, 0] = lamb_b * one[1]
    pot_b[2][1, :, 2] = lamb_b * q1[2]

    def a1_model():
        ham = TensorHamiltonian(
            ndof=4,
            potential=[[{(0, 1, 2, (3, 3)): TensorOperator(mpo=[w.copy() for w in pot_b], legs=(0, 1, 2, 3, 3))}]],
            kinetic=[[{((0, 0), (1, 1), (2, 2)): TensorOperator(mpo=[w.copy() for w in kin], legs=(0, 0, 1, 1, 2, 2))}]],
            backend="numpy",
        )
        m_ = Model(prim, {"hamiltonian": ham}, bond_dim=1)
        m_.init_HartreeProduct = [
            [ho.get_unitary()[0].tolist() for ho in prim[:3]] + [np.array([0.0, 1.0]).tolist()]
        ]
        return m_

    o = {f"pot{i}": w for i, w in enumerate(pot_b)}
    o.update({f"kin{i}": w for i, w in enumerate(kin)})
    o.update({f"w{i}": np.array(ho.get_unitary()[0]) for i, ho in enumerate(prim[:3])})
    bkw = dict(adaptive_Dmax=12, adaptive_dD=4, adaptive_p_proj=1.0e-5)
    for n in (2, 10):
        o.update(run_adaptive(a1_model(), 4, n, 0.1, **bkw))
    save("adaptive_exciton.npz", dt_au=np.array(0.1 / au_in_fs), Dmax=np.array(12), dD=np.array(4),
         p_proj=np.array(1.0e-5), **o)

    # (vi) one-site gates between the half-sweeps: Model(one_gate_to_apply=...) ->
    # MPSCoef.propagate -> apply_one_gate(reorth_center=nsite-1) (_mps_cls.py:489-490, :2314-2373)
    rng_g = np.random.default_rng(991)

    def crandn_g(*shape):
        return rng_g.standard_normal(shape) + 1j * rng_g.standard_normal(shape)

    # (vi-a) Hilbert space: unitary kicks, a full (4-leg) gate on site 1 and a diagonal (3-leg) one on site 4
    Lg, dg, Mg, Dg = 6, 3, 4, 6
    mpo_g = orc.synthetic_mpo(Lg, dg, Mg, seed=5)
    cores_g = [crandn_g(dl, dg, dr) for (dl, dr) in orc.bond_dims([dg] * Lg, Dg)]
    Hk = crandn_g(dg, dg)
    from scipy.linalg import expm as _expm
    U1 = _expm(-0.3j * (Hk + Hk.conj().T))
    U4 = np.exp(1j * rng_g.standard_normal(dg))
    gate_h = TensorHamiltonian(
        ndof=Lg,
        potential=[[{((1, 1),): TensorOperator(mpo=[U1[None, :, :, None]], legs=(1, 1)),
                     (4,): TensorOperator(mpo=[U4[None, :, None]], legs=(4,))}]],
        kinetic=None, backend="numpy",
    )
    model_g = Model([Exciton(nstate=dg) for _ in range(Lg)], operators={"hamiltonian": [w.copy() for w in mpo_g]},
                    bond_dim=Dg, one_gate_to_apply=gate_h)
    model_g.init_HartreeProduct = [[np.array(c) for c in cores_g]]
    o = {f"mpo{i}": w for i, w in enumerate(mpo_g)}
    o.update({f"init{i}": c for i, c in enumerate(cores_g)})
    o["U1"] = U1
    o["U4"] = U4
    for n in (1, 3):
        helper._Debug.niter_krylov.clear()
        sim = Simulator("gold_gate", model_g, backend="numpy", verbose=0)
        ener, wf = sim.propagate(stepsize=0.05, maxstep=n)
        o[f"n{n}_energy_last"] = np.array(ener)
        o[f"n{n}_energy_final"] = np.array(wf.expectation(model_g.hamiltonian))
        o[f"n{n}_norm"] = np.array(wf.norm())
        o[f"n{n}_autocorr"] = np.array(wf._ints_wf_ovlp_mpssm(wf.ci_coef, conj=False))
        o[f"n{n}_krylov"] = np.array([helper._Debug.niter_krylov[i] for i in range(Lg)])
        for i, s_ in enumerate(wf.ci_coef.superblock_states[0]):
            o[f"n{n}_final{i}"] = np.array(s_.data)
    save("gate_chain.npz", dt_au=np.array(0.05 / au_in_fs), nsite=np.array(Lg), **o)

    # (vi-b) Liouville space: a one-site super-gate exp(D dt) of a Lindblad dissipator on site 2
    # (tests/test_mixedstate.py:373-388 pattern) on top of the Liouvillian chain of (ii-d)
    Lm = crandn_g(2, 2) * 0.3
    E2_ = np.eye(2)
    Dsup = np.kron(Lm, Lm.conj()) - 0.5 * (np.kron(Lm.conj().T @ Lm, E2_) + np.kron(E2_, Lm.T @ Lm.conj()))
    Gsup = _expm(Dsup * 0.5)
    gate_l = TensorHamiltonian(
        ndof=Ll, potential=[[{((2, 2),): TensorOperator(mpo=[Gsup[None, :, :, None]], legs=(2, 2))}]], kinetic=None, backend="numpy",
    )
    lmodel_g = Model(lbasis, operators={"hamiltonian": [w.copy() for w in lmpo], "sz2": obs1, "sz1sx3": obs2}, bond_dim=Dl_,
                     space="liouville", one_gate_to_apply=gate_l)
    lmodel_g.init_HartreeProduct = [rhos]
    o = {f"mpo{i}": w for i, w in enumerate(lmpo)}
    o.update({f"rho{i}": r for i, r in enumerate(rhos)})
    o["sz"] = sz
    o["sx"] = sx
    o["G2"] = Gsup
    for n in (1, 3):
        helper._Debug.niter_krylov.clear()
        sim = Simulator("gold_lgate", lmodel_g, backend="numpy", verbose=0)
        _, wf = sim.propagate(stepsize=0.02, maxstep=n, integrator="arnoldi", autocorr=False, energy=False)
        o[f"n{n}_norm"] = np.array(wf.norm())
        o[f"n{n}_sz2"] = np.array(wf.expectation(lmodel_g.observables["sz2"]))
        o[f"n{n}_sz1sx3"] = np.array(wf.expectation(lmodel_g.observables["sz1sx3"]))
        for tag, legs in (("pt2", (0, 0, 2)), ("pt13", (0, 2, 0, 1))):
            o[f"n{n}_{tag}"] = np.array(wf.get_reduced_densities(legs)[0])
        o[f"n{n}_krylov"] = np.array([helper._Debug.niter_krylov[i] for i in range(Ll)])
        for i, s_ in enumerate(wf.ci_coef.superblock_states[0]):
            o[f"n{n}_final{i}"] = np.array(s_.data)
    save("gate_liouville.npz", dt_au=np.array(0.02 / au_in_fs), nsite=np.array(Ll), bond_dim=np.array(Dl_), **o)

    # (vii) Kraus maps on purified states: Model(kraus_op=...) -> apply_kraus after the forward
    # half-sweep (_mps_cls.py:491-492, :2375-2418; kraus.py).  tests/test_mixedstate.py:560-700
    # pattern with a synthetic chain: (a) the system site carries the ancilla index (d*K),
    # (b) a separate ancilla site right of the system site and a two-site map.
    from pytdscf.kraus import lindblad_to_kraus, trace_kraus_dim

    rng_k = np.random.default_rng(31337)

    def crandn_k(*shape):
        return rng_k.standard_normal(shape) + 1j * rng_k.standard_normal(shape)

    dk, Kk, Lk, Mk, Dk = 3, 4, 4, 4, 8
    # real Lindblad operators like the reference's tests (its Kraus self-check assumes them)
    Lops = [0.4 * rng_k.standard_normal((dk, dk)), 0.3 * rng_k.standard_normal((dk, dk))]
    dt_k = 0.05
    Bk = np.array(lindblad_to_kraus([x.copy() for x in Lops], 0.5))
    base = orc.synthetic_mpo(Lk, dk, Mk, seed=9)
    for p_ in range(Lk):  # weak non-Hermitian part so that Arnoldi / conserve_norm=False matter
        base[p_][0, :, :, base[p_].shape[3] - 1] += -0.01j * np.eye(dk)
    # (a) single-site map on site 1 whose physical index is (system, ancilla)
    mpo_k = [w.copy() for w in base]
    mpo_k[1] = np.einsum("aijb,kl->aikjlb", base[1], np.eye(Kk)).reshape(base[1].shape[0], dk * Kk, dk * Kk, base[1].shape[3])
    dims_k = [dk, dk * Kk, dk, dk]
    # full-rank random cores (a product start leaves rank-deficient bonds whose rounding noise the
    # bond propagation feeds back at ~1e-6, in the reference itself: no parity target)
    w_k = [crandn_k(a_, dd_, b_) for dd_, (a_, b_) in zip(dims_k, orc.bond_dims(dims_k, Dk))]
    model_k = Model([Exciton(nstate=dd_) for dd_ in dims_k], operators={"hamiltonian": [w.copy() for w in mpo_k]},
                    kraus_op={(1,): Bk.copy()}, bond_dim=Dk)
    model_k.init_HartreeProduct = [[w.copy() for w in w_k]]
    o = {f"mpo{i}": w for i, w in enumerate(mpo_k)}
    o.update({f"w{i}": w for i, w in enumerate(w_k)})
    o["B"] = Bk
    for n in (1, 4):
        helper._Debug.niter_krylov.clear()
        sim = Simulator("gold_kraus1", model_k, backend="numpy", verbose=0)
        _, wf = sim.propagate(stepsize=dt_k, maxstep=n, integrator="arnoldi", conserve_norm=False, autocorr=False, energy=False)
        o[f"n{n}_norm"] = np.array(wf.norm())
        o[f"n{n}_rdm1"] = trace_kraus_dim(np.array(wf.get_reduced_densities((0, 2))[0]), dk)
        o[f"n{n}_rdm2"] = np.array(wf.get_reduced_densities((0, 0, 2))[0])
        o[f"n{n}_krylov"] = np.array([helper._Debug.niter_krylov[i] for i in range(Lk)])
    save("kraus_single.npz", dt_au=np.array(dt_k / au_in_fs), bond_dim=np.array(Dk), d=np.array(dk), K=np.array(Kk), **o)

    # (b) two-site map: system site 1 (d), ancilla site 2 (K) with identity Hamiltonian cores
    Mb = base[1].shape[3]
    anc = np.einsum("ab,ij->aijb", np.eye(Mb), np.eye(Kk)).astype(np.complex128)
    mpo_k2 = [base[0].copy(), base[1].copy(), anc, base[2].copy(), base[3].copy()]
    dims_k2 = [dk, dk, Kk, dk, dk]
    w_k2 = [crandn_k(a_, dd_, b_) for dd_, (a_, b_) in zip(dims_k2, orc.bond_dims(dims_k2, Dk))]
    model_k2 = Model([Exciton(nstate=dd_) for dd_ in dims_k2], operators={"hamiltonian": [w.copy() for w in mpo_k2]},
                     kraus_op={(1, 2): Bk.copy()}, bond_dim=Dk)
    model_k2.init_HartreeProduct = [[w.copy() for w in w_k2]]
    o = {f"mpo{i}": w for i, w in enumerate(mpo_k2)}
    o.update({f"w{i}": w for i, w in enumerate(w_k2)})
    o["B"] = Bk
    for n in (1, 4):
        helper._Debug.niter_krylov.clear()
        sim = Simulator("gold_kraus2", model_k2, backend="numpy", verbose=0)
        _, wf = sim.propagate(stepsize=dt_k, maxstep=n, integrator="arnoldi", conserve_norm=False, autocorr=False, energy=False)
        o[f"n{n}_norm"] = np.array(wf.norm())
        o[f"n{n}_rdm1"] = np.array(wf.get_reduced_densities((0, 2))[0])
        o[f"n{n}_rdm3"] = np.array(wf.get_reduced_densities((0, 0, 0, 2))[0])
        o[f"n{n}_krylov"] = np.array([helper._Debug.niter_krylov[i] for i in range(Lk + 1)])
    save("kraus_two_site.npz", dt_au=np.array(dt_k / au_in_fs), bond_dim=np.array(Dk), d=np.array(dk), K=np.array(Kk), **o)

    # (viii) Simulator.operate: variational application of an operator (the Model's "hamiltonian"
    # entry, simulator_cls.py:356-360) to the state -> WFunc.apply_dipole (wavefunction.py:303-351)
    rng_o = np.random.default_rng(60606)

    def crandn_o(*shape):
        return rng_o.standard_normal(shape) + 1j * rng_o.standard_normal(shape)

    Lo, do, Mo, Do = 6, 3, 3, 5
    dip = orc.synthetic_mpo(Lo, do, Mo, seed=21)
    for p_ in range(Lo):  # a "dipole": make the local terms O(1) so that O|psi> is far from |psi>
        dip[p_][0, :, :, dip[p_].shape[3] - 1] += crandn_o(do, do)
    cores_o = [crandn_o(dl, do, dr) for (dl, dr) in orc.bond_dims([do] * Lo, Do)]
    model_o = Model([Exciton(nstate=do) for _ in range(Lo)], operators={"hamiltonian": [w.copy() for w in dip]}, bond_dim=Do)
    model_o.init_HartreeProduct = [[np.array(c) for c in cores_o]]
    o = {f"mpo{i}": w for i, w in enumerate(dip)}
    o.update({f"init{i}": c for i, c in enumerate(cores_o)})
    for n in (1, 10):
        sim = Simulator("gold_operate", model_o, backend="numpy", verbose=0)
        nrm, wf = sim.operate(maxstep=n)
        o[f"n{n}_norm"] = np.array(nrm)
        for i, s_ in enumerate(wf.ci_coef.superblock_states[0]):
            o[f"n{n}_final{i}"] = np.array(s_.data)
    save("operate_chain.npz", nsite=np.array(Lo), bond_dim=np.array(Do), **o)

    # (ix) adaptive bond dimension with a scalar term (coupleJ): the ovlp term enters the H_eff /
    # K_eff applies of the rank-selection functional through the <widened|thin> overlap blocks
    ham_s = TensorHamiltonian(La, potential=[[{tuple((i, i) for i in range(La)): TensorOperator(mpo=[w.copy() for w in mpo_a])}]],
                              kinetic=None, backend="numpy")
    ham_s.coupleJ = [[0.7]]
    model_s = Model(basis_a, operators={"hamiltonian": ham_s}, bond_dim=Da0)
    model_s.init_HartreeProduct = [[np.array(c) for c in cores_a]]
    o = {f"mpo{i}": w for i, w in enumerate(mpo_a)}
    o.update({f"init{i}": c for i, c in enumerate(cores_a)})
    for n in (1, 3):
        o.update(run_adaptive(model_s, La, n, 0.05, **akw))
    save("adaptive_chain_shift.npz", dt_au=np.array(0.05 / au_in_fs), nsite=np.array(La), bond_dim0=np.array(Da0),
         Dmax=np.array(7), dD=np.array(1), p_proj=np.array(1.0e-8), coupleJ=np.array(0.7), **o)

    # (x) several electronic states (MPS-SM, nstate = 2): one MPS per state, Hamiltonian blocks
    # H[i][j] (TensorHamiltonian(potential=[[..]]), hamiltonian_cls.py:628-752), the centre tensors
    # of all states stacked for the local solves.  NOTE the reference's NumPy backend caches its
    # contraction expressions by OPERATOR KEY only (_contraction.py:1050-1055, :1165-1172), so two
    # state pairs that use the same key (or both carry a "summed" / non-identity "ovlp" block) share
    # the first pair's blocks -- its JAX path has no such cache.  The fixture therefore gives every
    # pair a distinct full-chain key (mixing 4-leg and diagonal 3-leg cores) and keeps the
    # off-diagonal scalar terms zero, which makes the NumPy path exact.
    rng_m = np.random.default_rng(70707)

    def crandn_m(*shape):
        return rng_m.standard_normal(shape) + 1j * rng_m.standard_normal(shape)

    Lm, dm, Mm, Dm = 5, 3, 4, 5
    eye_m = np.eye(dm)
    adj = lambda w: np.ascontiguousarray(np.conj(w.transpose(0, 2, 1, 3)))  # noqa: E731
    full = lambda w: np.einsum("cit,ij->cijt", w, eye_m)  # noqa: E731
    h00 = orc.synthetic_mpo(Lm, dm, Mm, seed=31)
    d11 = [np.ascontiguousarray(np.einsum("ciit->cit", w).real) + 0j for w in orc.synthetic_mpo(Lm, dm, Mm, seed=32)]
    # coupling block: sites 0-2 four-leg, sites 3-4 diagonal; its adjoint is written with site 3 as a 4-leg core
    bm = [1, 3, 3, 2, 2, 1]
    h01 = [0.12 * crandn_m(bm[i], dm, dm, bm[i + 1]) for i in range(3)] + [0.5 * crandn_m(bm[3], dm, bm[4]), 0.5 * crandn_m(bm[4], dm, bm[5])]
    h10 = [adj(w) for w in h01[:3]] + [adj(full(h01[3])), np.conj(h01[4])]
    keys = {
        (0, 0): (tuple((i, i) for i in range(Lm)), None),
        (1, 1): (tuple(range(Lm)), tuple(range(Lm))),
        (0, 1): (((0, 0), (1, 1), (2, 2), 3, 4), (0, 0, 1, 1, 2, 2, 3, 4)),
        (1, 0): (((0, 0), (1, 1), (2, 2), (3, 3), 4), (0, 0, 1, 1, 2, 2, 3, 3, 4)),
    }
    blocks = {(0, 0): h00, (1, 1): d11, (0, 1): h01, (1, 0): h10}
    cj_m = [[0.0, 0.0], [0.0, 0.05]]

    def ham_m():
        pot = [[None, None], [None, None]]
        for (i_, j_), cores_ in blocks.items():
            key_, legs_ = keys[(i_, j_)]
            pot[i_][j_] = {key_: TensorOperator(mpo=[w.copy() for w in cores_], legs=legs_)}
            if cj_m[i_][j_] != 0.0:
                pot[i_][j_][()] = cj_m[i_][j_]
        return TensorHamiltonian(Lm, potential=pot, kinetic=None, backend="numpy")

    basis_m = [[Exciton(nstate=dm) for _ in range(Lm)] for _ in range(2)]
    bd_m = orc.bond_dims([dm] * Lm, Dm)
    init_m = [[crandn_m(a, dm, b) for a, b in bd_m] for _ in range(2)]
    weights_m = [0.8, 0.6]
    o = {"coupleJ": np.array(cj_m), "weights": np.array(weights_m)}
    for (i_, j_), cores_ in blocks.items():  # stored as full-chain 4-leg cores
        for p_, w in enumerate(cores_):
            o[f"mpo{i_}{j_}_{p_}"] = w if w.ndim == 4 else full(w)
    for s_ in range(2):
        o.update({f"init{s_}_{p_}": c for p_, c in enumerate(init_m[s_])})
    for tag, kw in (("", {}), ("relax_", {"relax": True}), ("improved_", {"relax": "improved"})):
        for n in (1, 3):
            model_m = Model(basis_m, operators={"hamiltonian": ham_m()}, bond_dim=Dm)
            model_m.init_HartreeProduct = [[np.array(c) for c in st_] for st_ in init_m]
            model_m.init_weight_ESTATE = list(weights_m)
            helper._Debug.niter_krylov.clear()
            sim = Simulator("gold_multistate", model_m, backend="numpy", verbose=0)
            if kw:
                ener, wf = sim.relax(stepsize=0.2, maxstep=n, improved=kw["relax"] == "improved")
            else:
                ener, wf = sim.propagate(stepsize=0.05, maxstep=n)
            pre = f"{tag}n{n}"
            o[f"{pre}_energy_last"] = np.array(ener)
            o[f"{pre}_energy_final"] = np.array(wf.expectation(model_m.hamiltonian))
            o[f"{pre}_pops"] = np.array(wf.pop_states())
            o[f"{pre}_norm"] = np.array(wf.norm())
            o[f"{pre}_autocorr"] = np.array(wf._ints_wf_ovlp_mpssm(wf.ci_coef, conj=False))
            o[f"{pre}_krylov"] = np.array([helper._Debug.niter_krylov[i] for i in range(Lm)])
            for s_ in range(2):
                for p_, c in enumerate(wf.ci_coef.superblock_states[s_]):
                    o[f"{pre}_final{s_}_{p_}"] = np.array(c.data)
    # Simulator.operate with two states (same blocks used as the "dipole")
    for n in (1, 10):
        model_m = Model(basis_m, operators={"hamiltonian": ham_m()}, bond_dim=Dm)
        model_m.init_HartreeProduct = [[np.array(c) for c in st_] for st_ in init_m]
        model_m.init_weight_ESTATE = list(weights_m)
        sim = Simulator("gold_multistate_op", model_m, backend="numpy", verbose=0)
        nrm, wf = sim.operate(maxstep=n)
        o[f"operate_n{n}_norm"] = np.array(nrm)
        o[f"operate_n{n}_pops"] = np.array(wf.pop_states())
        for s_ in range(2):
            for p_, c in enumerate(wf.ci_coef.superblock_states[s_]):
                o[f"operate_n{n}_final{s_}_{p_}"] = np.array(c.data)
    save("multistate_chain.npz", dt_au=np.array(0.05 / au_in_fs), dt_relax_au=np.array(0.2 / au_in_fs),
         nsite=np.array(Lm), nstate=np.array(2), bond_dim=np.array(Dm), **o)

    # (xi) the built-in equidistant-grid DVR bases (pytdscf/basis/sin.py, exponential.py): grids,
    # transformation and derivative matrices -- setup-side data of the user surface
    from pytdscf.basis import Exponential as RefExp, Sine as RefSine

    o = {}
    for tag, b in (("sine_t", RefSine(7, 3.0, x0=0.5, units="angstrom", include_terminal=True)),
                   ("sine_n", RefSine(6, 4.0, x0=-1.0, units="bohr", include_terminal=False)),
                   ("exp", RefExp(7, 2.0 * np.pi, x0=0.1))):
        o[f"{tag}_grids"] = np.array(b.get_grids())
        o[f"{tag}_unitary"] = np.array(b.get_unitary())
        o[f"{tag}_sqrt_weights"] = np.array(b.get_sqrt_weights())
        o[f"{tag}_d1_dvr"] = np.array(b.get_1st_derivative_matrix_dvr())
        o[f"{tag}_d2_dvr"] = np.array(b.get_2nd_derivative_matrix_dvr())
        o[f"{tag}_d1_fbr"] = np.array(b.get_1st_derivative_matrix_fbr())
        o[f"{tag}_d2_fbr"] = np.array(b.get_2nd_derivative_matrix_fbr())
        o[f"{tag}_fbr2_at_grid"] = np.array([b.fbr_func(2, x) for x in b.get_grids()])
        o[f"{tag}_dvr3_at_grid"] = np.array([b.dvr_func(3, x) for x in b.get_grids()])
        if tag != "exp":
            o[f"{tag}_pos"] = np.array(b.get_pos_rep_matrix())
    save("basis_dvr.npz", **o)

    # (xii) polynomial (sum-of-products) Hamiltonians with the MPS standard method: BASELINE
    # configs[0] (tests/test_harmonic_fbr_sm_propagate_numpy.py) and the anharmonic H2O force
    # field (tests/test_anharmonic_fbr_mpssm_propagate_np.py).  The reference runs them through its
    # SoP contractions (MPSCoefSoP); the engine runs the exact MPO of the same operator.
    import math as _math

    from pytdscf.basis._primints_cls import PrimBas_HO as RefPrimHO
    from pytdscf.hamiltonian_cls import PolynomialHamiltonian as RefPoly, read_potential_nMR as ref_read_nMR
    from pytdscf.model_cls import BasInfo as RefBasInfo
    from pytdscf.potentials.h2o_potential import k_orig as h2o_k

    o = {}
    prim_h = [[RefPrimHO(0.0, 1500, 8), RefPrimHO(0.0, 2000, 8)]]
    bi = RefBasInfo(prim_h)
    ham_h = RefPoly(ndof=2)
    ham_h.set_HO_potential(bi)
    ener, _ = Simulator("harmonic_fbr_sm", Model(bi, {"hamiltonian": ham_h}), ci_type="standard-method",
                        backend="numpy", verbose=0).propagate(maxstep=1)
    assert abs(ener - 0.007973586692598029) < 1e-12  # the reference's own pin
    o["harmonic_energy"] = np.array(ener)
    keys = sorted(k for k in h2o_k.keys())
    o["h2o_keys"] = np.array([list(k) + [0] * (4 - len(k)) for k in keys])  # zero padded mode labels
    o["h2o_vals"] = np.array([h2o_k[k] for k in keys])

    def h2o_model(nprim, bond_dim):
        prim = [[RefPrimHO(0.0, _math.sqrt(h2o_k[(i, i)]) * units.au_in_cm1, nprim) for i in (1, 2, 3)]]
        return Model(RefBasInfo(prim), {"hamiltonian": ref_read_nMR(h2o_k)}, bond_dim=bond_dim)

    ener, _ = Simulator("anharmonic_fbr_propagate_sm", h2o_model(6, 4), backend="numpy", verbose=0).propagate(maxstep=2)
    assert abs(ener - 0.021360262338234466) < 1e-12  # the reference's own pin
    o["h2o_energy_pin"] = np.array(ener)
    for n in (1, 5):  # dynamics from |000> and from a vibrationally excited product state
        for tag, wts in (("gs", None), ("ex", [[[0.0, 1.0, 0.0, 0.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0, 0.0, 0.0], [0.6, 0.8, 0.0, 0.0, 0.0, 0.0]]])):
            m_ = h2o_model(6, 4)
            if wts is not None:
                m_.init_weight_VIBSTATE = wts
            ener, wf = Simulator("h2o_dyn", m_, backend="numpy", verbose=0).propagate(stepsize=0.2, maxstep=n)
            o[f"h2o_{tag}_n{n}_energy_last"] = np.array(ener)
            o[f"h2o_{tag}_n{n}_autocorr"] = np.array(wf.autocorr())
            o[f"h2o_{tag}_n{n}_norm"] = np.array(wf.norm())
            for p_, c in enumerate(wf.ci_coef.superblock_states[0]):
                o[f"h2o_{tag}_n{n}_final{p_}"] = np.array(c.data)
    save("polynomial_sm.npz", dt_au=np.array(0.2 / au_in_fs), **o)


if __name__ == "__main__":
    main()
