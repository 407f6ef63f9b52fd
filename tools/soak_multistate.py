"""Soak run of the multi-state mode: many steps with three coupled states; prints device memory in
use and step time at intervals (leak / drift check)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pytdscf_amd import MultiStateEngine, synthetic

L, d, M, D, S = 10, 5, 4, 12, 3
rng = np.random.default_rng(0)
crandn = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)  # noqa: E731
mpo = [[None] * S for _ in range(S)]
for i in range(S):
    mpo[i][i] = synthetic.synthetic_mpo(L, d, M, seed=i)
    for j in range(i + 1, S):
        w = [0.05 * crandn(a, d, d, b) for a, b in zip([1] + [2] * (L - 1), [2] * (L - 1) + [1])]
        mpo[i][j], mpo[j][i] = w, [np.ascontiguousarray(np.conj(c.transpose(0, 2, 1, 3))) for c in w]
eng = MultiStateEngine(L, S)
eng.set_hamiltonian(mpo, [[0.0, 0.01, 0.0], [0.01, 0.1, 0.02j], [0.0, -0.02j, 0.2]])
eng.set_states([synthetic.random_mps_cores([d] * L, D, seed=s) for s in range(S)], weights=[1.0, 0.0, 0.0])
nstep = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
t0 = time.perf_counter()
free0 = None
e0 = eng.expectation().real
for s in range(nstep):
    eng.propagate(0.3)
    if s % 200 == 199 or s == nstep - 1:
        free, tot = torch.cuda.mem_get_info()
        free0 = free0 or free
        print(f"step {s + 1}: {1e3 * (time.perf_counter() - t0) / (s + 1):.2f} ms/step  norm-1 {eng.norm() - 1:+.1e}  "
              f"dE {eng.expectation().real - e0:+.1e}  pops {np.round(eng.pop_states(), 4)}  "
              f"device memory in use {(tot - free) / 2**20:.0f} MiB (change since first report {(free0 - free) / 2**20:+.0f})", flush=True)
eng.close()
