"""H_eff applies at the C4 interior shape with the bench's finite-state-machine MPO core (block-sparse W stage)
on random environments / centre tensor, through the fine seam (mitdvp_heff_apply, reps applies on the device):
for rocprofv3 kernel-trace / PMC passes.   python tools/heff_fsm_probe.py [D d M reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytdscf_amd import engine as E, synthetic as syn

D, d, M, reps = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (1024, 16, 32, 3)
rng = np.random.default_rng(0)
def crandn(*s):
    a = rng.standard_normal(s + (2,))
    return a.view(np.complex128).reshape(s)
W = syn.synthetic_mpo(4, d, M, seed=0)[1]  # an interior core (M, d, d, M)
Lb, Rb, psi = crandn(D, M, D), crandn(D, M, D), crandn(D, d, D)
out, ms = E.heff_apply(Lb, W, Rb, psi, reps=reps)  # ms: average per apply
f = 8.0 * (D * D * M * d * D + D * D * M * M * d * d + D * D * D * M * d)
print(f"H_eff apply ({D},{d},{D}) M={M} FSM core, mode={E.get_gemm_mode()}: {ms:.2f} ms per apply = {f / ms / 1e9:.2f} TFLOP/s algorithmic", flush=True)
