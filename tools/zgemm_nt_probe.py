import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from pytdscf_amd import engine as E
rng=np.random.default_rng(0)
for (m,n,k) in [(512,512,8192),(1024,1024,16384),(256,256,4096),(128,128,2048),(512,512,2048),(2048,512,8192)]:
    A=rng.standard_normal((m,k))+1j*rng.standard_normal((m,k)); B=rng.standard_normal((n,k))+1j*rng.standard_normal((n,k))
    for cfg in (-1,1,2):
        out,ms=E.zgemm(A,B,transB=True,tile_cfg=cfg,reps=5)
        print(m,n,k,"cfg",cfg,round(ms*1e3,1),"us",round(8*m*n*k/ms/1e9,1),"TF", flush=True)
