#!/usr/bin/env python3
"""Hénon–Heiles chain (f coupled anharmonic oscillators) on a DVR grid: potential from mode
functions (n-mode representation -> diagonal MPO), kinetic energy MPO, real-time propagation with
an adaptive bond dimension.

    python examples/henon_heiles.py [f N steps]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytdscf_amd import HarmonicOscillator as HO, Model, Simulator, units  # noqa: E402
from pytdscf_amd.dvr_operator_cls import construct_kinetic_mpo, construct_nMR_recursive  # noqa: E402


def main():
    f, N, steps = [int(x) for x in sys.argv[1:4]] + [6, 10, 50][len(sys.argv) - 1:]
    w_cm1, lam = 2000.0, 2.0e-3
    w = w_cm1 / units.au_in_cm1
    prims = [HO(N, w_cm1) for _ in range(f)]
    func = {}
    for i in range(f):
        cubic = 0.0 if i == 0 else -lam * w**1.5 / 3.0
        func[(i,)] = lambda q, c=cubic: 0.5 * w**2 * q**2 + c * q**3
        if i + 1 < f:
            func[(i, i + 1)] = lambda q1, q2: lam * w**1.5 * q1**2 * q2
    pot = construct_nMR_recursive(prims, nMR=2, func=func)
    kin = construct_kinetic_mpo(prims)
    model = Model(prims, {"potential": pot, "kinetic": kin}, bond_dim=2)
    # displaced start: first excited DVR-basis function on every second mode
    model.init_weight_VIBSTATE = [[[0.0, 1.0] + [0.0] * (N - 2) if i % 2 == 0 else [1.0] + [0.0] * (N - 1) for i in range(f)]]
    sim = Simulator("hh", model)
    ener, wf = sim.propagate(stepsize=0.05, maxstep=steps, adaptive=True, adaptive_Dmax=24, adaptive_dD=4, adaptive_p_proj=1.0e-6)
    print(f"E = {ener * units.au_in_cm1:.3f} cm-1, norm = {wf.norm():.12f}, bond dimensions {wf.bonddim()}")


if __name__ == "__main__":
    main()
