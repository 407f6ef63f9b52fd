"""The spin-bath model of the reference's strongest physics pin, ``tests/test_mixedstate.py``, and its exact solution.

Model (``tests/test_mixedstate.py:32-74``; parameters are DATA of that test): bath spin 1/2 (site 0) -- system spin 1
(site 1) -- bath spin 1/2 (site 2),

    H = Bx Ix + By Iy + Bz Iz + J01 S0.I1 + J12 I1.S2,   Haberkorn loss -k_H rho,
    Lindblad jumps on the system spin: sqrt(k_amp) |m+1><m|-type lowering, sqrt(k_deph) Iz.

The exact solution (``:104-236``) is the dense propagator exp(L dt) of the vectorised Liouvillian applied step by step
and traced down to the system spin; it is recomputed HERE with ``scipy.linalg.expm`` (nothing of the reference is
imported or stored).  The reference builds its matrix-product operators with PyMPO, a third-party package absent from
the image; the same operators are written out as sums of products below and turned into MPOs by the direct sum of
``pytdscf_amd.operators`` (exact; the dense matrix of every MPO is checked against the dense operator in the tests).

Five set-ups of the reference's test, each as ``case_*() -> dict`` consumed by the oracle driver and by the shell:
trajectories (``:239-318``), vectorised density matrix with / without the dissipator as a super-gate (``:321-457``),
purified state (``:460-558``), Kraus maps on one / two sites (``:561-811``)."""

from __future__ import annotations

import numpy as np
from scipy.linalg import expm

J01, J12 = 1.0, 0.5
BX = BY = BZ = 1.0
K_HAB = 0.1
K_AMP, K_DEPH = 6.0, 9.0
DT, NSTEPS = 0.1, 11

SX = np.array([[0, 1], [1, 0]], dtype=complex) / 2
SY = np.array([[0, -1j], [1j, 0]]) / 2
SZ = np.array([[1, 0], [0, -1]], dtype=complex) / 2
IP = np.array([[0, 1, 0], [0, 0, 1], [0, 0, 0]], dtype=complex) * np.sqrt(2) / 2
IM = IP.T.copy()
IX, IY = 0.5 * (IP + IM), -0.5j * (IP - IM)
IZ = np.diag([1.0, 0.0, -1.0]).astype(complex) / 2  # the reference's spin-1 "Iz" (test_mixedstate.py:45)
E2, E3 = np.eye(2, dtype=complex), np.eye(3, dtype=complex)
L_AMP = np.array([[0, 1, 0], [0, 0, 1], [0, 0, 0]], dtype=complex) * np.sqrt(K_AMP)
L_DEPH = IZ * np.sqrt(K_DEPH)
JUMPS = (L_AMP, L_DEPH)


def k3(a, b, c):
    return np.kron(np.kron(a, b), c)


def hamiltonian_dense():
    H = BX * k3(E2, IX, E2) + BY * k3(E2, IY, E2) + BZ * k3(E2, IZ, E2)
    for s, i in ((SX, IX), (SY, IY), (SZ, IZ)):
        H = H + J01 * k3(s, i, E2) + J12 * k3(E2, i, s)
    return H


def dissipator(jumps, eye):
    """sum_j L (x) conj(L) - (L^+ L (x) 1 + 1 (x) L^T conj(L)) / 2, row-major vectorisation"""
    D = 0
    for L in jumps:
        LdL = L.conj().T @ L
        D = D + np.kron(L, L.conj()) - 0.5 * (np.kron(LdL, eye) + np.kron(eye, LdL.T))
    return D


def exact_rdms(ini_diag=(0, 0, 1), lindblad=True, dt=DT, nsteps=NSTEPS):
    """System-spin density matrix at t = 0, dt, ..., (nsteps-1) dt from the dense propagator."""
    n = 12
    H = hamiltonian_dense()
    eye = np.eye(n)
    Lv = (np.kron(H, eye) - np.kron(eye, H.T)) / 1j - K_HAB * np.eye(n * n)
    if lindblad:
        Lv = Lv + dissipator([k3(E2, L, E2) for L in JUMPS], eye)
    P = expm(Lv * dt)
    p0 = np.diag(np.asarray(ini_diag, dtype=complex))
    rho = k3(E2 / 2, p0 / np.trace(p0), E2 / 2).reshape(-1)
    out = []
    for _ in range(nsteps):
        out.append(np.einsum("abcadc->bd", rho.reshape(2, 3, 2, 2, 3, 2)))
        rho = P @ rho
    return np.array(out)


# --------------------------------------------------------------------------- operators as sums of products -> MPO
def sop_mpo(terms, dims):
    """[(coefficient, {site: matrix}), ...] -> one full-chain MPO: every product is a bond-1 chain (identities on the
    sites it skips), the sum is their direct sum, rounded losslessly."""
    from pytdscf_amd.operators import compress_mpo, merge_operator_terms

    chains = []
    for coef, ops in terms:
        sites = sorted(ops)
        cores = [np.asarray(ops[s], dtype=complex)[None, :, :, None] for s in sites]
        cores[0] = cores[0] * coef
        chains.append((cores, sites))
    return compress_mpo(merge_operator_terms(chains, dims))


def hilbert_terms(s0, s1, s2, pad=lambda m: m):
    """H - i k_H / 2 on (bath, system, bath) = sites (s0, s1, s2); ``pad`` widens the system-spin operators
    (Kraus ancilla inside the physical index)."""
    t = [(BX, {s1: pad(IX)}), (BY, {s1: pad(IY)}), (BZ, {s1: pad(IZ)}), (-0.5j * K_HAB, {s1: pad(E3)})]
    for s, i in ((SX, IX), (SY, IY), (SZ, IZ)):
        t.append((J01, {s0: s, s1: pad(i)}))
        t.append((J12, {s1: pad(i), s2: s}))
    return t


def liouville_terms(with_dissipator):
    """The super-operator i L = H (x) 1 - 1 (x) H^T - i k_H (+ i D) on sites of dimension 4, 9, 4"""
    def le(a, e):  # acts on the ket leg
        return np.kron(a, e)

    def ri(a, e):  # acts on the bra leg
        return np.kron(e, a.T)

    t = []
    for c, i in ((BX, IX), (BY, IY), (BZ, IZ)):
        t += [(c, {1: le(i, E3)}), (-c, {1: ri(i, E3)})]
    for s, i in ((SX, IX), (SY, IY), (SZ, IZ)):
        t += [(J01, {0: le(s, E2), 1: le(i, E3)}), (-J01, {0: ri(s, E2), 1: ri(i, E3)})]
        t += [(J12, {1: le(i, E3), 2: le(s, E2)}), (-J12, {1: ri(i, E3), 2: ri(s, E2)})]
    t.append((-1j * K_HAB, {1: np.kron(E3, E3)}))
    if with_dissipator:
        t.append((1j, {1: dissipator(JUMPS, E3)}))
    return t


def _pair(n, first=True):
    """maximally entangled ancilla-physical pair as two cores: (1, n, n) then (n, n, 1)"""
    a = np.zeros((1, n, n)) if first else np.zeros((n, n, 1))
    for k in range(n):
        if first:
            a[0, k, k] = 1
        else:
            a[k, k, 0] = 1
    return a


def case_trajectories():
    dims = [2, 3, 2]
    starts = [[[1, 0], [0, 0, 1], [1, 0]], [[1, 0], [0, 0, 1], [0, 1]], [[0, 1], [0, 0, 1], [1, 0]], [[0, 1], [0, 0, 1], [0, 1]]]
    return dict(dims=dims, mpo=sop_mpo(hilbert_terms(0, 1, 2), dims), starts=starts, space="hilbert", key=(1, 1), scale=1,
                exact=dict(ini_diag=(0, 0, 1), lindblad=False), system_dim=3, ancilla=1)


def case_liouville(supergate, scale=1):
    dims = [4, 9, 4]
    p0 = np.diag([0.0, 0.0, 1.0])
    c = dict(dims=dims, mpo=sop_mpo(liouville_terms(not supergate), dims), starts=[[E2.reshape(-1), p0.reshape(-1), E2.reshape(-1)]],
             space="liouville", key=(1, 1), scale=scale, exact=dict(ini_diag=(0, 0, 1), lindblad=True), system_dim=3, ancilla=1)
    if supergate:
        c["gate"] = {1: expm(dissipator(JUMPS, E3) * DT / scale)}
    return c


def case_purified():
    dims = [2, 2, 3, 2, 2]
    sys_ = np.zeros((1, 3, 1))
    sys_[0, 2, 0] = 1
    start = [_pair(2), _pair(2, False), sys_, _pair(2), _pair(2, False)]
    return dict(dims=dims, mpo=sop_mpo(hilbert_terms(1, 2, 3), dims), starts=[start], space="hilbert", key=(2, 2), scale=1,
                exact=dict(ini_diag=(0, 0, 1), lindblad=False), system_dim=3, ancilla=1)


def case_kraus_single(scale=2, kdim=24):
    from pytdscf_amd.kraus import lindblad_to_kraus

    dims = [2, 2, 3 * kdim, 2, 2]
    ek = np.eye(kdim)
    sys_ = np.zeros((1, 3 * kdim, 1))
    sys_[0, 2 * kdim, 0] = 1
    start = [_pair(2), _pair(2, False), sys_, _pair(2), _pair(2, False)]
    return dict(dims=dims, mpo=sop_mpo(hilbert_terms(1, 2, 3, pad=lambda m: np.kron(m, ek)), dims), starts=[start], space="hilbert",
                key=(2, 2), scale=scale, kraus={(2,): lindblad_to_kraus(list(JUMPS), DT / scale)},
                exact=dict(ini_diag=(0, 0, 1), lindblad=True), system_dim=3, ancilla=kdim)


def case_kraus_two_site(scale=2, kdim=32):
    from pytdscf_amd.kraus import lindblad_to_kraus

    dims = [2, 2, 3, kdim, 2, 2]
    sys_ = np.zeros((1, 3, 2))
    sys_[0, 2, 0] = 1
    sys_[0, 1, 1] = 1
    anc = np.zeros((2, kdim, 1))
    anc[0, 0, 0] = 1
    anc[1, 1, 0] = 1
    start = [_pair(2), _pair(2, False), sys_, anc, _pair(2), _pair(2, False)]
    return dict(dims=dims, mpo=sop_mpo(hilbert_terms(1, 2, 4), dims), starts=[start], space="hilbert", key=(2, 2), scale=scale,
                kraus={(2, 3): lindblad_to_kraus(list(JUMPS), DT / scale)}, exact=dict(ini_diag=(0, 1, 1), lindblad=True),
                system_dim=3, ancilla=1)


def system_rdm(rho, case):
    """the ancilla inside the physical index (single-site Kraus maps) traced out (kraus.py:434-455)"""
    n, k = case["system_dim"], case["ancilla"]
    rho = np.asarray(rho)
    return rho if k == 1 else np.einsum("aKbK->ab", rho.reshape(n, k, n, k))


def legs_of(key, nsite):
    legs = [0] * nsite
    for s in key:
        legs[s] += 1
    return legs[: max(key) + 1]


def run_oracle(case):
    """The NumPy oracle through the reference's loop (observables before the step, simulator_cls.py:419-454): the
    system-spin density at every step, averaged over the start states."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd.mps import product_state_cores

    liou = case["space"] == "liouville"
    nst = NSTEPS * case["scale"]
    acc = 0
    for start in case["starts"]:
        cores = orc.canonicalize_site0(product_state_cores(start, 64, space=case["space"]), scale=None if liou else 1.0)
        st = orc.OracleMPS(cores, case["mpo"], integrator="arnoldi", conserve_norm=False, gates=case.get("gate"),
                           kraus=case.get("kraus"))
        legs = legs_of(case["key"], len(case["dims"]))
        out = []
        for _ in range(nst):
            rho = orc.liouville_partial_trace(st.cores, legs) if liou else orc.reduced_density(st.cores, legs)
            out.append(system_rdm(rho, case))
            st.propagate(DT / case["scale"])
        acc = acc + np.array(out)
    return acc / len(case["starts"])


def run_shell(case, jobname):
    """The same run through the PyTDSCF-shaped shell on the GPU, as the reference's test drives it: Model / Simulator
    .propagate(reduced_density=..., integrator="arnoldi", conserve_norm=False), densities read back from
    {jobname}_prop/reduced_density.nc."""
    from pytdscf_amd import Exciton, Model, Simulator, TensorHamiltonian, TensorOperator, units
    from pytdscf_amd.util import read_nc

    liou = case["space"] == "liouville"
    kw = {}
    if "gate" in case:
        (site, U), = case["gate"].items()
        kw["one_gate_to_apply"] = TensorHamiltonian(len(case["dims"]), potential=[[{((site, site),): TensorOperator(
            mpo=[U[None, :, :, None]], legs=(site, site))}]], kinetic=None, backend="hip")
    if "kraus" in case:
        kw["kraus_op"] = case["kraus"]
    if liou:
        kw["space"] = "Liouville"
    acc = 0
    for i, start in enumerate(case["starts"]):
        model = Model([Exciton(nstate=d) for d in case["dims"]], operators={"hamiltonian": case["mpo"]}, bond_dim=64, **kw)
        model.init_HartreeProduct = [start]
        sim = Simulator(jobname=f"{jobname}_{i}", model=model, backend="hip", verbose=0)
        sim.propagate(reduced_density=([case["key"]], 1), maxstep=NSTEPS * case["scale"], stepsize=DT * units.au_in_fs / case["scale"],
                      autocorr=False, energy=False, norm=False, populations=False, conserve_norm=False, integrator="arnoldi")
        data = read_nc(f"{jobname}_{i}_prop/reduced_density.nc", [case["key"]])
        acc = acc + np.array([system_rdm(r, case) for r in data[case["key"]]])
    return acc / len(case["starts"])
