import sys; sys.path.insert(0,'.')
import numpy as np
from oracle import tdvp_oracle as orc
from pytdscf_amd import TDVPEngine
for (L,d,M,D) in [(16,4,16,64),(128,4,16,8),(64,4,16,8),(100,4,16,8),(128,4,16,64)]:
    eng=TDVPEngine(L); eng.set_mpo(orc.synthetic_mpo(L,d,M,seed=0)); eng.init_random([d]*L,D,seed=1)
    cores=eng.get_mps()
    bad=[i for i,c in enumerate(cores) if not np.isfinite(c).all()]
    print(L,d,M,D,'nan sites',bad[:10], 'norm', eng.norm(), 'E', eng.expectation(), 'E oracle', orc.OracleMPS(cores, orc.synthetic_mpo(L,d,M,seed=0)).expectation() if not bad else None, flush=True)
    eng.close()
