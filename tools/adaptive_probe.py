"""Adaptive bond dimension (a1TDVP) timing: ranks grow from D0 towards Dmax on a synthetic
chain; prints the bond dimensions and seconds per time step on the GPU and, with --cpu,
for the NumPy oracle on the host cores of the same box (the checker, timed as a baseline).

    python tools/adaptive_probe.py L d M D0 Dmax dD steps [--cpu]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pytdscf_amd import TDVPEngine, synthetic

args = [a for a in sys.argv[1:] if not a.startswith("--")]
L, d, M, D0, Dmax, dD, steps = (int(x) for x in args) if len(args) == 7 else (16, 8, 16, 32, 256, 32, 4)
dt, p = 0.5, 1e-10
mpo = synthetic.synthetic_mpo(L, d, M, seed=0)
eng = TDVPEngine(L)
eng.set_mpo(mpo)
eng.init_random([d] * L, D0, seed=1)
init = eng.get_mps()
eng.set_adaptive(True, Dmax=Dmax, dD=dD, p_proj=p)
out = {"config": dict(L=L, d=d, M=M, D0=D0, Dmax=Dmax, dD=dD, dt=dt, p_proj=p), "gpu": [], "cpu": []}
eng.counters_reset()
for s in range(steps):
    t0 = time.perf_counter()
    eng.propagate(dt)
    nrm = eng.norm()  # synchronises
    out["gpu"].append(dict(step=s, s=round(time.perf_counter() - t0, 4), max_bond=max(eng.bond_dims()), norm=nrm))
    print("gpu", out["gpu"][-1], flush=True)
c = eng.counters()
out["gpu_counters"] = {k: c[k] for k in ("n_heff", "n_env", "n_qr", "n_launch") if k in c}
out["bond_dims_gpu"] = eng.bond_dims()
if "--cpu" in sys.argv:  # CPU baseline leg: the only place this tool touches oracle/
    from oracle import tdvp_oracle as orc

    st = orc.OracleMPS([x.copy() for x in init], mpo, adaptive=True, Dmax=Dmax, dD=dD, p_proj=p)
    for s in range(steps):
        t0 = time.perf_counter()
        st.propagate(dt)
        out["cpu"].append(dict(step=s, s=round(time.perf_counter() - t0, 3), max_bond=max(c.shape[2] for c in st.cores[:-1])))
        print("cpu", out["cpu"][-1], flush=True)
    out["bond_dims_cpu"] = [c.shape[2] for c in st.cores[:-1]]
    out["cpu_threads"] = os.cpu_count()
print("RESULT " + json.dumps(out))
