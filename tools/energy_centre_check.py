import sys; sys.path.insert(0,'.')
import bench, numpy as np
from pytdscf_amd import TDVPEngine, synthetic as syn
for (L,d,D,M,integ) in ((10,10,32,6,'lanczos'),(6,32,128,16,'lanczos'),(14,4,64,16,'arnoldi')):
    liou = integ=='arnoldi'
    mpo = syn.synthetic_liouvillian_mpo(L, M, seed=0, gamma=0.002) if liou else syn.synthetic_mpo(L,d,M,seed=0)
    e = TDVPEngine(L, integrator=integ, conserve_norm=not liou)
    e.set_mpo(mpo); e.init_random([d]*L, D, seed=1)
    e.sweep(0.5, True); e.sweep(0.5, False)
    a = e.expectation(); b = bench.energy_at_centre(e)
    e.sweep(0.5, True)
    c = bench.energy_at_centre(e)
    e.sweep(0.5, False)
    print(L,d,D, 'expect', a, 'centre(site0)', b, 'centre(last)', c, 'expect after', e.expectation())
    e.close()
