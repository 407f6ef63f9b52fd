"""A few QRs of one site shape (for rocprofv3 --kernel-trace: the launch sequence of one factorisation, durations in order).
  python tools/qr_one.py D d"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytdscf_amd import engine as E

D, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 32)
rng = np.random.default_rng(0)
psi = rng.standard_normal((D, d, D)) + 1j * rng.standard_normal((D, d, D))
for _ in range(4):
    E.gauge_trf(psi, "Psi2Asigma")
