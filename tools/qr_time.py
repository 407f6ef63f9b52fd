"""Device time per gauge-move QR (HIP events around the QR phase) at several site shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytdscf_amd import TDVPEngine
from pytdscf_amd import engine as E

if len(sys.argv) > 1:  # python tools/qr_time.py 0|1 : panel algorithm (0 per-column Householder, 1 CholeskyQR2 + reconstruction)
    E.set_qr_fast(bool(int(sys.argv[1])))
print("qr_fast", E.get_qr_fast())

shapes = [("C4i", 16, 1024, 8), ("C5", 4, 512, 12), ("C3", 32, 128, 6), ("C2", 10, 32, 10), ("C4", 16, 1024, 5), ("mid", 8, 256, 10)]
for name, d, D, L in shapes:
    eng = TDVPEngine(L)
    eng.init_random([d] * L, D, seed=1)   # warm-up (allocations)
    eng.set_profiling(True)
    eng.counters_reset()
    eng.init_random([d] * L, D, seed=2)   # L-1 QRs of the canonicalisation
    c = eng.counters()
    full = [i for i in range(L) if eng.get_site_shape(i)[0] == D and eng.get_site_shape(i)[2] == D]
    print(name, "d", d, "D", D, "n_qr", c["n_qr"], "qr ms total", round(c["qr_ms"], 3), "ms/QR", round(c["qr_ms"] / max(c["n_qr"], 1), 3),
          "interior sites", len(full), "launches", c["n_launch"], flush=True)
    eng.close()
