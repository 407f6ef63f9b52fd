"""Small-bond regime (C2-like) timing with and without per-phase HIP-event profiling."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytdscf_amd import synthetic as orc  # product-side synthetic inputs (oracle/ is test infrastructure)
from pytdscf_amd import TDVPEngine

L, d, D, M = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (10, 10, 32, 6)
dt = float(sys.argv[5]) if len(sys.argv) > 5 else 2.0
mpo = orc.synthetic_mpo(L, d, M, seed=0)
for prof in (False, True, False):
    eng = TDVPEngine(L)
    eng.set_mpo(mpo)
    eng.init_random([d] * L, D, seed=1)
    eng.set_profiling(prof)
    for _ in range(2):
        eng.propagate(dt)
    eng.norm()
    eng.counters_reset()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        eng.propagate(dt)
    eng.norm()
    el = time.perf_counter() - t0
    c = eng.counters()
    print(f"profiling={prof}: {2 * n / el:.1f} sweeps/s, {1e3 * el / (2 * n):.2f} ms/sweep, launches/sweep {c['n_launch'] / (2 * n):.0f}, "
          f"us/launch {1e6 * el / c['n_launch']:.2f}, heff applies/sweep {c['n_heff'] / (2 * n):.0f}", flush=True)
    eng.close()
