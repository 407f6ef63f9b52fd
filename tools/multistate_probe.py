#!/usr/bin/env python3
"""Time per step of the multi-state mode against the single-state engine on the same chain
(S states, all pairs coupled: S*S operator chains per apply).

    python tools/multistate_probe.py [L d D M S steps]
"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pytdscf_amd import synthetic as orc  # product-side synthetic inputs  # noqa: E402
from pytdscf_amd import MultiStateEngine, TDVPEngine  # noqa: E402


def main():
    a = [int(x) for x in sys.argv[1:7]] + [12, 16, 128, 8, 2, 6][len(sys.argv) - 1:]
    L, d, D, M, S, steps = a
    rng = np.random.default_rng(0)
    crandn = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)  # noqa: E731
    raw = [[crandn(x, d, y) for x, y in orc.bond_dims([d] * L, D)] for _ in range(S)]
    diag = [orc.synthetic_mpo(L, d, M, seed=s) for s in range(S)]
    mpo = [[None] * S for _ in range(S)]
    for i in range(S):
        mpo[i][i] = diag[i]
        for j in range(i + 1, S):
            w = [0.05 * crandn(x, d, d, y) for x, y in zip([1] + [2] * (L - 1), [2] * (L - 1) + [1])]
            mpo[i][j] = w
            mpo[j][i] = [np.ascontiguousarray(np.conj(c.transpose(0, 2, 1, 3))) for c in w]
    dt = 0.2
    one = TDVPEngine(L)
    one.set_mpo(diag[0])
    one.set_mps(raw[0], canonicalize=True)
    one.propagate(dt)
    t0 = time.perf_counter()
    for _ in range(steps):
        one.propagate(dt)
    one.norm()
    t_one = (time.perf_counter() - t0) / steps
    k_one = float(np.mean(one.krylov_stats()))
    ms = MultiStateEngine(L, S)
    ms.set_hamiltonian(mpo, [[0.0] * S for _ in range(S)])
    ms.set_states(raw, weights=[1.0] * S)
    ms.propagate(dt)
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        ms.propagate(dt)
        ms.norm()
        ts.append(time.perf_counter() - t0)
    print(json.dumps({"L": L, "d": d, "D": D, "M": M, "S": S, "s_per_step_single": t_one, "krylov_single": k_one,
                      "s_per_step_multi": float(np.mean(ts)), "s_per_step_multi_first_last": [ts[0], ts[-1]],
                      "krylov_multi": float(np.mean(ms.krylov_stats())), "norm": ms.norm(), "pops": ms.pop_states()}))


if __name__ == "__main__":
    main()
