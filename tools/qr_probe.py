"""Standalone gauge move (QR) at a given site shape, for rocprofv3."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytdscf_amd import engine as E

dl, d, dr = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (1024, 16, 1024)
rng = np.random.default_rng(0)
psi = rng.standard_normal((dl, d, dr)) + 1j * rng.standard_normal((dl, d, dr))
for key in ("Psi2Asigma", "Psi2sigmaB"):
    t0 = time.perf_counter()
    site, sig = E.gauge_trf(psi, key)
    print(key, "wall incl. H2D/D2H", round(time.perf_counter() - t0, 3), "s", flush=True)
A = site.reshape(dl, d * dr)
print("orth err", abs(A @ A.conj().T - np.eye(dl)).max())
