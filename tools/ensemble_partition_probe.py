"""Small-bond regime, ensemble mode: B replicas on disjoint compute-unit ranges (TDVPEnsemble, one library call per batch of
time steps) against one engine on the whole chip and one engine on a slice.   python tools/ensemble_partition_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytdscf_amd import synthetic as syn
from pytdscf_amd import TDVPEngine, TDVPEnsemble

L, d, D, M, dt = 10, 10, 32, 6, 2.0
mpo = syn.synthetic_mpo(L, d, M, seed=0)
nstep = 20

e = TDVPEngine(L)
e.set_mpo(mpo); e.init_random([d] * L, D, seed=1)
for _ in range(3): e.propagate(dt)
e.norm(); t0 = time.perf_counter()
for _ in range(nstep): e.propagate(dt)
e.norm(); el = time.perf_counter() - t0
print(f"one engine, whole chip: {2 * nstep / el:.1f} sweeps/s", flush=True)
e.close()

for B in (1, 2, 4, 8, 16):
    try:
        ens = TDVPEnsemble(B, L)
    except Exception as ex:
        print(B, "replicas:", ex); continue
    ens.set_mpo(mpo)
    for r, g in enumerate(ens.engines):
        g.init_random([d] * L, D, seed=1 + r)
    ens.propagate(dt, 3)
    t0 = time.perf_counter()
    ens.propagate(dt, nstep)
    el = time.perf_counter() - t0
    c = ens[0].counters()
    print(f"{B} replicas x {ens.cu_per_replica} CUs: {B * 2 * nstep / el:.1f} sweeps/s aggregate, {2 * nstep / el:.1f} per replica; "
          f"norms {[round(g.norm(), 12) for g in ens.engines][:2]}; replica 0: {c['n_launch'] / (2 * (nstep + 3)):.0f} launches/sweep, host waits {c['n_host_waits']:.0f}", flush=True)
    ens.close()
