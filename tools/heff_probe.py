"""One H_eff apply stream at a given site shape on device-resident random operands
(for rocprofv3 kernel-trace / PMC passes): python tools/heff_probe.py D d M reps"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytdscf_amd import engine as E

D, d, M, reps = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (1024, 16, 32, 3)
if len(sys.argv) > 5:
    E.set_gemm_mode(sys.argv[5])
ms = E.bench_heff(D, d, D, M, M, reps=reps, warmup=1)
f = 8.0 * (D * D * M * d * D + D * D * M * M * d * d + D * D * D * M * d)
print(f"H_eff apply ({D},{d},{D}) M={M} mode={E.get_gemm_mode()}: {ms:.2f} ms = {f / ms / 1e9:.2f} TFLOP/s algorithmic")
