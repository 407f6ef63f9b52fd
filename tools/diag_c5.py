import sys; sys.path.insert(0,'.')
import numpy as np
from oracle import tdvp_oracle as orc
from pytdscf_amd import TDVPEngine
from pytdscf_amd import engine as E
for mode in ("3m","4m"):
    E.set_gemm_mode(mode)
    for (L,d,M,D,dt) in [(16,4,16,64,0.5),(24,4,16,128,0.5)]:
        mpo=orc.synthetic_mpo(L,d,M,seed=0); mps=orc.synthetic_mps([d]*L,D,seed=1)
        st=orc.OracleMPS([c.copy() for c in mps],mpo); eng=TDVPEngine(L); eng.set_mpo(mpo); eng.set_mps(mps)
        try:
            for i in range(2):
                st.propagate(dt); eng.propagate(dt)
            print(mode,L,d,M,D,'k',sorted(set(st.kprev.values())), sorted(set(eng.krylov_stats())), abs(st.expectation()-eng.expectation()), abs(abs(orc.overlap(st.cores,eng.get_mps()))-1), flush=True)
        except ValueError as e:
            print(mode,L,d,M,D,'FAIL',e, eng.krylov_stats(), flush=True)
