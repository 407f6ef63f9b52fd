"""C2-like run for rocprofv3 --kernel-trace: wall time of N steps next to the kernel-time sum."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytdscf_amd import synthetic as orc  # product-side synthetic inputs (oracle/ is test infrastructure)
from pytdscf_amd import TDVPEngine
L, d, D, M = 10, 10, 32, 6
eng = TDVPEngine(L)
eng.set_mpo(orc.synthetic_mpo(L, d, M, seed=0))
eng.init_random([d] * L, D, seed=1)
for _ in range(2):
    eng.propagate(2.0)
eng.norm()
t0 = time.perf_counter()
for _ in range(10):
    eng.propagate(2.0)
eng.norm()
print("wall ms for 20 sweeps", 1e3 * (time.perf_counter() - t0), flush=True)
