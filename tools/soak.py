"""Soak run: thousands of small steps with adaptive growth, gates and observables; prints device
memory in use and step time at intervals (leak / drift check)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pytdscf_amd import synthetic as orc  # product-side synthetic inputs (oracle/ is test infrastructure)
from pytdscf_amd import TDVPEngine

L, d, M = 12, 6, 5
mpo = orc.synthetic_mpo(L, d, M, seed=0)
eng = TDVPEngine(L)
eng.set_mpo(mpo)
eng.init_random([d] * L, 4, seed=1)
eng.set_adaptive(True, Dmax=48, dD=4, p_proj=1e-9)
rng = np.random.default_rng(0)
eng.set_gates({3: np.exp(1j * 0.01 * rng.standard_normal(d))})
nstep = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
t0 = time.perf_counter()
free0 = None
for s in range(nstep):
    eng.propagate(0.3)
    if s % 250 == 249 or s == nstep - 1:
        eng.expectation(); eng.autocorr(); eng.site_rdm(2)
        free, tot = torch.cuda.mem_get_info()
        free0 = free0 or free
        print(f"step {s + 1}: {1e3 * (time.perf_counter() - t0) / (s + 1):.2f} ms/step  norm-1 {eng.norm() - 1:+.1e}  "
              f"device memory in use {(tot - free) / 2**20:.0f} MiB (change since first report {(free0 - free) / 2**20:+.0f})  "
              f"max bond {max(eng.bond_dims())}", flush=True)
eng.close()
