"""K_eff-second-stage-like NT GEMMs (short output, long contraction) under different split-K workgroup targets:
MITDVP_SPLITK_TARGET=384|512|768 python tools/splitk_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytdscf_amd import engine as E
rng = np.random.default_rng(0)
for (m, n, k) in [(512, 512, 8192), (1024, 1024, 32768), (128, 128, 2048), (256, 256, 4096), (2048, 512, 8192), (32, 480, 2048), (32, 992, 16384)]:
    A = rng.standard_normal((m, k)) + 1j * rng.standard_normal((m, k))
    B = rng.standard_normal((n, k)) + 1j * rng.standard_normal((n, k))
    out, ms = E.zgemm(A, B, transB=True, reps=10)
    print(os.environ.get("MITDVP_SPLITK_TARGET", "384"), m, n, k, round(ms * 1e3, 1), "us", round(8 * m * n * k / ms / 1e9, 1), "TF algorithmic", flush=True)
