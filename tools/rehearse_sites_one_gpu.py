"""Two site-sharded ranks on ONE GPU (gloo) at the C4 interior shape (D = 1024, d = 16, M = 32, L = 8): one time
step next to the serial engine on the same inputs -- exercises the 1024^2 pseudo-inverse, the full-size junction
update and the host-staged halo messages.  Start with:  python tools/rehearse_sites_one_gpu.py  (spawns its ranks).
RS_WORLD ranks (default 2, at most 6 on one GPU), RS_L / RS_D / RS_STEPS; RS_SETUP_ONLY=1: the set-up alone (seconds per
rank, device memory in use by all ranks together afterwards) -- the full C4 chain is RS_L=64."""
import json, os, socket, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if "WORLD_SIZE" not in os.environ:
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=os.environ.get("RS_WORLD", "2"), MITDVP_DIST_BACKEND="gloo",
               MITDVP_SMALL_KERNELS="0")
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(env, RANK=str(r), LOCAL_RANK="0")) for r in range(int(os.environ.get("RS_WORLD", "2")))]
    rc = 0
    for p in ps:
        rc = rc or p.wait()
    sys.exit(rc)

import numpy as np
from pytdscf_amd import TDVPEngine, synthetic as syn
from pytdscf_amd.dist import Comm
from pytdscf_amd.parallel_sites import SiteShardedTDVP

L, d, D, M, dt = int(os.environ.get("RS_L", 8)), 16, int(os.environ.get("RS_D", 1024)), 32, 0.5
comm = Comm()
mpo = syn.synthetic_mpo(L, d, M, seed=0)
t0 = time.time()
eng = SiteShardedTDVP(comm, mpo, dims=[d] * L, bond_dim=D, seed=1)
assert eng.selftest()
t_setup = time.time() - t0
if os.environ.get("RS_SETUP_ONLY"):
    import torch

    comm.barrier()
    free, total = torch.cuda.mem_get_info(0)
    allt = comm.max_over_ranks(t_setup)
    print(json.dumps(dict(rank=comm.rank, world=comm.world, L=L, D=D, setup_mode=eng.setup_mode, setup_s=t_setup, setup_s_max=allt,
                          device_GB_in_use_all_ranks=(total - free) / 1e9, norm=eng.norm(), energy=eng.expectation().real)), flush=True)
    comm.barrier()
    eng.close()
    comm.close()
    sys.exit(0)
nsteps = int(os.environ.get("RS_STEPS", 1))
trace = [(0, eng.norm(), eng.expectation().real)]  # folded rank by rank on the device, no gather
t0 = time.time()
for k in range(nsteps):
    eng.step(dt)
    if nsteps > 1:
        trace.append((k + 1, eng.norm(), eng.expectation().real))
comm.barrier()
t_step = (time.time() - t0) / nsteps
g = eng.gather()
if comm.rank == 0:
    ser = TDVPEngine(L)
    ser.set_mpo(mpo)
    ser.init_random([d] * L, D, seed=1)
    t0 = time.time()
    for _ in range(nsteps):
        ser.propagate(dt)
    n = ser.norm()
    t_ser = time.time() - t0
    ref = ser.get_mps()
    # <ref|g> and <g|g> with plain transfer contractions (bond 1024: a few seconds of NumPy)
    def ov(a, b):
        e = np.ones((1, 1), complex)
        for x, y in zip(a, b):
            e = np.einsum("ab,aic,bid->cd", e, x.conj(), y, optimize=True)
        return e[0, 0]
    gg = ov(g, g).real
    print(json.dumps(dict(L=L, D=D, steps=nsteps, norm_energy_trace=trace, serial_energy=ser.expectation().real, setup_s=t_setup, step_s=t_step, serial_step_s=t_ser, norm=float(np.sqrt(gg)),
                          infidelity=float(1 - abs(ov(ref, g)) / np.sqrt(gg) / n), halo_GB=eng.traffic()[0] / 1e9)), flush=True)
comm.barrier()
eng.close()
comm.close()
