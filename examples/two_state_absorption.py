#!/usr/bin/env python3
"""Vibronic absorption spectrum of a two-state linear-vibronic-coupling model with several
electronic states kept as separate MPS (the multi-state mode):

    relax on S0  ->  operate the transition dipole (S0 -> S1)  ->  propagate  ->  FFT of <Psi(0)|Psi(t)>

Runs on one MI355X.  Everything below is the PyTDSCF-shaped surface of ``pytdscf_amd``.

    python examples/two_state_absorption.py [nsteps]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytdscf_amd as pytdscf  # noqa: E402
from pytdscf_amd import HarmonicOscillator as HO, Model, Simulator, TensorHamiltonian, TensorOperator, units  # noqa: E402

freqs_cm1 = [800.0, 1200.0, 1600.0, 3000.0]
kappa = [0.6, -0.4, 0.3, 0.2]  # dimensionless displacements of S1 along each mode
delta_e = 0.02  # vertical gap (a.u.)
lam = 0.002  # S0 / S1 coupling through mode 0 (a.u.)
N, D = 10, 8
prims = [HO(N, w, units="cm-1") for w in freqs_cm1]
nmode = len(prims)


def harmonic_terms(shift):
    """{key: TensorOperator}: kinetic energy per mode (4-leg, DVR second derivative) and the
    harmonic wells, displaced by kappa for the excited state (diagonal in the DVR)."""
    terms = {}
    for i, p in enumerate(prims):
        q = np.asarray(p.get_grids())
        w = p.omega
        v = 0.5 * w**2 * q**2 + (shift * kappa[i] * w**1.5 * q if shift else 0.0)
        terms[(i,)] = TensorOperator(mpo=[np.asarray(v, dtype=complex).reshape(1, N, 1)], legs=(i,))
        terms[((i, i),)] = TensorOperator(mpo=[(-0.5 * p.get_2nd_derivative_matrix_dvr()).astype(complex).reshape(1, N, N, 1)], legs=(i, i))
    return terms


q0 = np.asarray(prims[0].get_grids())
coupling = {(0,): TensorOperator(mpo=[(lam * np.sqrt(prims[0].omega) * q0).astype(complex).reshape(1, N, 1)], legs=(0,))}
ham = TensorHamiltonian(nmode, potential=[[harmonic_terms(0), coupling], [coupling, {**harmonic_terms(1), (): delta_e}]])
# transition dipole: constant (Condon), S0 <-> S1
dip = TensorHamiltonian(nmode, potential=[[{}, {(): 1.0}], [{(): 1.0}, {}]], name="dipole")
basis = [prims, prims]  # the same primitive basis for both electronic states


def model_for(op):
    m = Model(basis, operators={"hamiltonian": op}, bond_dim=D)
    m.init_weight_ESTATE = [1.0, 0.0]  # everything in S0, vibrational ground state of the DVR basis
    return m


def main():
    nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    job = "lvc2"
    e_gs, _ = Simulator(job, model_for(ham)).relax(stepsize=1.0, maxstep=30, improved=False, savefile_ext="_gs")
    print(f"S0 ground state energy {e_gs * units.au_in_cm1:10.2f} cm-1 (sum of zero-point energies "
          f"{0.5 * sum(freqs_cm1):.2f})")
    nrm, wf = Simulator(job, model_for(dip)).operate(maxstep=10, restart=True, loadfile_ext="_gs", savefile_ext="_dip")
    print(f"|mu Psi| = {nrm:.6f}, populations after the dipole {np.round(wf.pop_states(), 6)}")
    sim = Simulator(job, model_for(ham))
    e, wf = sim.propagate(stepsize=0.1, maxstep=nsteps, restart=True, loadfile_ext="_dip", autocorr=True, energy=True)
    print(f"<H> on S1 {e * units.au_in_eV:8.4f} eV, populations {np.round(wf.pop_states(), 4)}, norm {wf.norm():.12f}")
    t, ac = pytdscf.spectra.load_autocorr(f"{job}_prop/autocorr.dat")
    wn, inten = pytdscf.spectra.ifft_autocorr(t, ac, E_shift=e_gs * units.au_in_eV)
    pytdscf.spectra.export_spectrum(wn, inten, f"{job}_prop/spectrum.dat")
    sel = (wn > 0) & (wn < 12000)
    print(f"strongest line at {wn[sel][np.argmax(inten[sel])]:.0f} cm-1 above the S0 zero-point level; "
          f"spectrum written to {job}_prop/spectrum.dat")


if __name__ == "__main__":
    main()
