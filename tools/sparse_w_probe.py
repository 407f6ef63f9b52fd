"""Stage times of the H_eff apply at the C4 interior shape with the block-sparse W stage on / off
(MITDVP_SPARSE_W), on a short chain whose middle sites are 1024 x 16 x 1024 (L = 8)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytdscf_amd import TDVPEngine, synthetic as syn

L, d, D, M = 8, 16, 1024, 32
out = {}
for flag in sys.argv[1:] or ["1", "0"]:
    os.environ["MITDVP_SPARSE_W"] = flag
    eng = TDVPEngine(L)
    eng.set_mpo(syn.synthetic_mpo(L, d, M, seed=0))
    eng.init_random([d] * L, D, seed=1)
    eng.sweep(0.5, True)
    eng.norm()
    eng.counters_reset()
    eng.set_profiling(True)
    eng.sweep(0.5, False)
    eng.norm()
    c = eng.counters()
    out[flag] = dict(n_heff=c["n_heff"], stage_ms_per_apply=[x / c["n_heff"] for x in c["heff_stage_ms"]], heff_ms=c["heff_ms"],
                     env_ms=c["env_ms"], skipped_share=c["heff_flops_skipped"] / c["heff_flops"])
    eng.close()
print(json.dumps(out))
