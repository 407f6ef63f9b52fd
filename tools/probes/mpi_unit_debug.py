"""debug: the reference's tests/test_mpi.py state (zero-padded product start, H = 2) on 1 / 2 ranks, adaptive on / off:
norm after every step and half step"""
import os, sys, json
os.environ["MITDVP_SMALL_KERNELS"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import tdvp_oracle as orc
from pytdscf_amd import mps as M
from pytdscf_amd.dist import Comm
from pytdscf_amd.parallel_sites import SiteShardedTDVP
comm = Comm()
split = {1: [(0, 11)], 2: [(0, 5), (6, 11)]}[comm.world]
wv = [[1.0, 0.0, 0.0, 0.0], [1.0, 1.0, 0.0, 0.0], [1.0, 1.0, 1.0, 0.0]] + [[1.0, 1.0, 1.0, 1.0]] * 9
mpo = [np.eye(4, dtype=complex).reshape(1, 4, 4, 1) * (2.0 if i == 0 else 1.0) for i in range(12)]
for bond in (10,):
    for adaptive in (False, True):
        for reg, ps in ((True, 1e-7), (True, None)):
            start = orc.canonicalize_site0(M.product_state_cores(wv, bond_dim=bond))
            ad = dict(Dmax=30, dD=30, p_proj=1e-4) if adaptive else None
            eng = SiteShardedTDVP(comm, mpo, cores=start, split=split, regularize=reg, p_svd=ps, adaptive=ad)
            norms = [eng.norm()]
            for k in range(2):
                eng.step(0.1)
                norms.append(eng.norm())
            x = eng.X if comm.rank < comm.world - 1 else None
            if comm.rank == 0:
                print("bond", bond, "adaptive", adaptive, "reg", reg, "p_svd", ps, "norms", norms, "bonds", eng.bond_dims(),
                      "sv(X)", None if x is None else np.linalg.svd(x, compute_uv=False)[:3], flush=True)
            comm.barrier()
            eng.close()
comm.close()
