"""Phase timeline of the small-site kernel (MITDVP_SS_TRACE=1): a few C2 sweeps, one line per launch on stderr."""
import os, sys
os.environ.setdefault("MITDVP_SS_TRACE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytdscf_amd import TDVPEngine, synthetic as syn

L, d, D, M = 10, 10, 32, 6
eng = TDVPEngine(L)
eng.set_mpo(syn.synthetic_mpo(L, d, M, seed=0))
eng.init_random([d] * L, D, seed=1)
for i in range(3):
    eng.sweep(2.0, i % 2 == 0)
print("norm", eng.norm(), "k", eng.krylov_stats())
