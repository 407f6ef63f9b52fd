"""Jacobi SVD on the device: time, sweeps and accuracy against LAPACK for square complex matrices.
  python tools/svd_probe.py [n ...]     (MITDVP_SVD_BLOCKED=0 selects the row-pair step, MITDVP_SVD_INNER the inner sweeps)"""
import json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytdscf_amd import engine as E

sizes = [int(a) for a in sys.argv[1:]] or [256, 512, 1024]
rng = np.random.default_rng(3)
E.svd(rng.standard_normal((64, 64)) + 0j)  # load, probe, warm up
for n in sizes:
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    # graded spectrum: singular values over 12 decades, like a bond matrix before truncation
    u, _, vh = np.linalg.svd(A)
    sv = np.logspace(0, -12, n)
    B = (u * sv) @ vh
    out = {"n": n}
    Bs = B[np.argsort(-np.linalg.norm(B, axis=1))]  # rows by descending norm (de Rijk's ordering, done by the caller here)
    for name, X in (("random", A), ("graded", B), ("graded_rows_sorted", Bs)):
        t0 = time.time()
        U, S, Vh, sweeps = E.svd(X)
        dt = time.time() - t0
        Sref = np.linalg.svd(X, compute_uv=False)
        out[name] = dict(seconds=round(dt, 4), sweeps=int(sweeps), s_err=float(np.max(np.abs(S - Sref)) / Sref[0]),
                         s_relerr_max=float(np.max(np.abs(S - Sref) / Sref)),
                         recon=float(np.linalg.norm((U * S) @ Vh - X) / np.linalg.norm(X)),
                         orth=float(np.linalg.norm(U.conj().T @ U - np.eye(n))))
    print(json.dumps(out), flush=True)
