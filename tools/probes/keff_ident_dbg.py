"""K_eff identity-state shortcut on the Liouvillian chain: where do the two forms differ, and which is closer to the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import tdvp_oracle as orc
from pytdscf_amd import TDVPEngine
from pytdscf_amd import synthetic as syn

L, d, M, D = 8, 4, 16, 256
for gamma in (0.002, 0.02):
    mpo = syn.synthetic_liouvillian_mpo(L, M, seed=0, gamma=gamma)
    mps = orc.synthetic_mps([d] * L, D, seed=1)
    ref = orc.OracleMPS([c.copy() for c in mps], mpo, integrator="arnoldi", conserve_norm=False)
    ref.propagate(0.3)
    out = {}
    for flag in ("1", "0", "0b"):
        os.environ["MITDVP_KEFF_IDENT"] = flag[0]
        eng = TDVPEngine(L, integrator="arnoldi", conserve_norm=False)
        eng.set_mpo(mpo); eng.set_mps(mps)
        eng.propagate(0.3)
        out[flag] = eng.get_mps(); ks = eng.krylov_stats()
        eng.close()
        print(gamma, flag, "krylov", ks, "norm", abs(orc.overlap(out[flag], out[flag])))
    for x, y in (("1", "0"), ("0", "0b")):
        print(gamma, x, y, "per-site max diff", ["%.1e" % np.abs(p - q).max() for p, q in zip(out[x], out[y])])
    for x in ("1", "0"):
        print(gamma, x, "vs oracle", ["%.1e" % np.abs(p - q).max() for p, q in zip(out[x], ref.cores)],
              "max elem", max(np.abs(p).max() for p in out[x]))
