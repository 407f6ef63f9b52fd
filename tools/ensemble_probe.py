"""Small-bond regime: several independent engines (one HIP stream each) stepped from host threads
share the GPU -- aggregate throughput of an ensemble of trajectories vs a single one."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytdscf_amd import synthetic as orc  # product-side synthetic inputs (oracle/ is test infrastructure)
from pytdscf_amd import TDVPEngine

L, d, D, M = 10, 10, 32, 6
mpo = orc.synthetic_mpo(L, d, M, seed=0)
nstep = 10


def make(seed):
    e = TDVPEngine(L)
    e.set_mpo(mpo)
    e.init_random([d] * L, D, seed=seed)
    for _ in range(2):
        e.propagate(2.0)
    e.norm()
    return e


def run(e):
    for _ in range(nstep):
        e.propagate(2.0)
    e.norm()


for nt in (1, 2, 4, 8):
    engs = [make(s + 1) for s in range(nt)]
    th = [threading.Thread(target=run, args=(e,)) for e in engs]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    el = time.perf_counter() - t0
    print(f"{nt} engines: {nt * 2 * nstep / el:.1f} sweeps/s aggregate, {2 * nstep / el:.1f} per engine, norms {[round(e.norm(), 12) for e in engs][:2]}", flush=True)
    for e in engs:
        e.close()
