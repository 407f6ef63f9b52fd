import sys; sys.path.insert(0,'.')
import numpy as np
from oracle import tdvp_oracle as orc
from pytdscf_amd import TDVPEngine
L,d,M,D=128,4,16,64
rng=np.random.default_rng(3)
bd=orc.bond_dims([d]*L,D)
cores=[rng.standard_normal((a,d,b))+1j*rng.standard_normal((a,d,b)) for a,b in bd]
eng=TDVPEngine(L); eng.set_mpo(orc.synthetic_mpo(L,d,M,seed=0)); eng.set_mps(cores, canonicalize=True)
out=eng.get_mps()
bad=[i for i,c in enumerate(out) if not np.isfinite(c).all()]
print('host-random + device canonicalize: nan sites', bad[:5], bad[-5:], 'norm', eng.norm())
ref=orc.canonicalize_site0(cores)
if not bad: print('fidelity', abs(orc.overlap(ref,out)))
# device RNG only: L sites but D small so that canonicalize is cheap -> fetch raw? use init_random then check
eng2=TDVPEngine(L); eng2.set_mpo(orc.synthetic_mpo(L,d,M,seed=0))
for seed in (1,2,3):
    eng2.init_random([d]*L,D,seed=seed)
    o=eng2.get_mps(); bad=[i for i,c in enumerate(o) if not np.isfinite(c).all()]
    print('seed',seed,'nan sites',bad[:3],bad[-3:])
