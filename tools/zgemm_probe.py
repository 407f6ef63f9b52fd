"""Standalone zgemm runs for rocprofv3 (kernel trace / PMC passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytdscf_amd import engine as E

mode = sys.argv[1] if len(sys.argv) > 1 else "3m"
E.set_gemm_mode(mode)
rng = np.random.default_rng(0)
for (m, n, k, cfg) in [(8192, 8192, 1024, 0), (8192, 8192, 1024, 1)]:
    A = rng.standard_normal((m, k)) + 1j * rng.standard_normal((m, k))
    B = rng.standard_normal((k, n)) + 1j * rng.standard_normal((k, n))
    out, ms = E.zgemm(A, B, tile_cfg=cfg, reps=10)
    print(mode, m, n, k, cfg, round(ms, 3), "ms", round(8.0 * m * n * k / ms / 1e9, 2), "TF", flush=True)
