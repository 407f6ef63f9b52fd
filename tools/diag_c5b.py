import sys; sys.path.insert(0,'.')
import numpy as np
from oracle import tdvp_oracle as orc
from pytdscf_amd import TDVPEngine
L,d,M = 128,4,16
for D,dt in [(64,0.5),(256,0.5),(512,0.1),(512,0.5)]:
    eng=TDVPEngine(L); eng.set_mpo(orc.synthetic_mpo(L,d,M,seed=0)); eng.init_random([d]*L,D,seed=1)
    e0=eng.expectation()
    try:
        eng.sweep(dt,True); eng.sweep(dt,False)
        print(D,dt,'ok k',sorted(set(eng.krylov_stats())),'E',e0,eng.expectation(),'norm',eng.norm(),flush=True)
    except ValueError as e:
        print(D,dt,'FAIL',e,'k',eng.krylov_stats()[:40],'E0',e0,flush=True)
    eng.close()
