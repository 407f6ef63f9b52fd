"""A/B of the QR panel algorithms at the gauge-move shapes of the BASELINE configs (device time, launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytdscf_amd import engine as E

shapes = [("C5 2048x512", 2048, 512), ("C3 4096x128", 4096, 128), ("C4 16384x1024", 16384, 1024), ("mid 2048x256", 2048, 256),
          ("C2-like 640x64", 640, 64)]
for name, m, n in shapes:
    row = []
    for fast in (False, True):
        E.set_qr_fast(fast)
        ms, nl = E.bench_qr(m, n, reps=5 if m * n > 4e6 else 20)
        row.append((ms, nl))
    E.set_qr_fast(True)
    print(f"{name:16s} per-column {row[0][0]:8.3f} ms {row[0][1]:5d} launches | CholQR2+reconstruct {row[1][0]:8.3f} ms {row[1][1]:5d} launches | x{row[0][0] / row[1][0]:.2f}", flush=True)
