"""H_eff apply rate over a range of site shapes (device-resident operands, HIP-event timing)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytdscf_amd import engine as E

for (D, d, M) in [(32, 10, 6), (64, 16, 8), (64, 8, 16), (128, 8, 16), (128, 16, 16), (128, 32, 16), (256, 8, 16), (256, 16, 16),
                  (256, 16, 32), (512, 4, 16), (512, 16, 32), (1024, 16, 32)]:
    ms = E.bench_heff(D, d, D, M, M, reps=5 if D < 1024 else 2, warmup=2 if D < 1024 else 1)
    fl = 8.0 * (D * D * M * d * D * 2 + D * D * M * M * d * d)
    print(f"D={D:5d} d={d:3d} M={M:3d}  {ms * 1e3:10.1f} us  {fl / ms / 1e9:7.2f} TFLOP/s", flush=True)
