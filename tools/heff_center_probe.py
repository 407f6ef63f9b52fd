"""H_eff applies at the centre of a CANONICAL chain under the bench's generators (identity blocks in the environments, the
finite-state-machine / direct-sum MPO cores): exactly the kernels a local exponential issues (mitdvp_heff_apply_center),
reps times after one warm-up, for rocprofv3 kernel-trace / PMC passes.
    [MITDVP_EDGE_APPLY=0|1] python tools/heff_center_probe.py C3|C5|C4 [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytdscf_amd import TDVPEngine, synthetic as syn

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg = {"C3": (6, 32, 128, 16, False), "C5": (14, 4, 512, 16, True), "C4": (7, 16, 1024, 32, False)}[name]
L, d, D, M, liou = cfg
mpo = syn.synthetic_liouvillian_mpo(L, M, seed=0, gamma=0.002) if liou else syn.synthetic_mpo(L, d, M, seed=0)
eng = TDVPEngine(L, integrator="arnoldi" if liou else "lanczos", conserve_norm=not liou)
eng.set_mpo(mpo)
eng.init_random([d] * L, D, seed=1)
c = L // 2
eng.build_envs(1)
for _ in range(c):
    eng.split_center(True)
    eng.absorb_bond(True)
shape = eng.get_site_shape(c)[:3]
x = eng.get_site(c)
_, flags = eng.heff_apply_center(x)  # warm-up (also builds the cached reduced cores of the edge form)
print(f"PROBE_BEGIN {name} site {c} shape {shape} flags {flags} reps {reps}", flush=True)
t0 = time.perf_counter()
for _ in range(reps):
    eng.heff_apply_center(x)
print(f"PROBE_END {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per call (with host copies)", flush=True)
eng.close()
