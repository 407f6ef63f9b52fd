"""Device time of one H_eff apply at every site of a workload's chain (HIP events of the engine's own phase timers),
as the local exponentials issue them:  python tools/heff_per_site.py C3 [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytdscf_amd import TDVPEngine, synthetic as syn

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
L, d, D, M, liou = {"C3": (6, 32, 128, 16, False), "C5": (12, 4, 512, 16, True), "C2": (10, 10, 32, 6, False)}[name]
mpo = syn.synthetic_liouvillian_mpo(L, M, seed=0, gamma=0.002) if liou else syn.synthetic_mpo(L, d, M, seed=0)
eng = TDVPEngine(L, integrator="arnoldi" if liou else "lanczos", conserve_norm=not liou)
eng.set_mpo(mpo)
eng.init_random([d] * L, D, seed=1)
eng.build_envs(1)
for c in range(L):
    shape = eng.get_site_shape(c)[:3]
    x = eng.get_site(c)
    _, flags = eng.heff_apply_center(x)
    eng.counters_reset()
    eng.set_profiling(True)
    for _ in range(reps):
        eng.heff_apply_center(x)
    eng.norm()
    k = eng.counters()
    eng.set_profiling(False)
    n = max(k["n_heff"], 1)
    print(f"site {c} {shape} flags {flags:#x}: {k['heff_ms'] / n * 1e3:8.1f} us per apply, stages "
          f"{[round(v / n * 1e3, 1) for v in k['heff_stage_ms']]}, algorithmic {k['heff_flops'] / n / 1e9:.2f} GF, "
          f"executed {(k['heff_flops'] - k['heff_flops_skipped']) / n / 1e9:.2f} GF", flush=True)
    if c + 1 < L:
        eng.split_center(True)
        eng.absorb_bond(True)
eng.close()
