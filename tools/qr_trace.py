"""Phase timeline of the one-workgroup QR kernels (thread 0, s_memrealtime stamps per column step):
  MITDVP_QR_TRACE=1 python tools/qr_trace.py [D d]
phases: totals + zlarfg | update | stage + barrier | publish | barrier  (the stamps serialise wave 0: read the step total)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytdscf_amd import engine as E

D, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 10)
rng = np.random.default_rng(0)
psi = rng.standard_normal((D, d, D)) + 1j * rng.standard_normal((D, d, D))
for _ in range(3):
    E.gauge_trf(psi, "Psi2Asigma")
