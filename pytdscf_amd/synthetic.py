"""Synthetic benchmark inputs: the Hermitian chain MPO of configurations
C2-C4 and the Lindblad-like generator of C5.  Product-side twin of the builders the oracle keeps for
its own tests (``tests/test_host_logic.py`` asserts they are identical), so that ``bench.py`` and the
tools never touch ``oracle/`` outside the CPU-baseline leg."""

from __future__ import annotations

import numpy as np

from .mps import bond_dims  # noqa: F401  (LatticeInfo.get_bond_dim)


def synthetic_mpo(L: int, d: int, M: int, seed: int = 0, dtype=np.complex128):
    """Hermitian nearest-neighbour-like MPO with bond M = K+2."""
    rng = np.random.default_rng(seed)
    K = M - 2
    cores = []
    for p in range(L):
        W = np.zeros((M, d, d, M), dtype=dtype)
        eye = np.eye(d)

        def herm(scale):
            G = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
            return scale * (G + G.conj().T) / 2

        W[0, :, :, 0] = eye
        W[M - 1, :, :, M - 1] = eye
        for k in range(1, K + 1):
            A = herm(0.01)
            W[0, :, :, k] = A
            W[k, :, :, M - 1] = A
        W[0, :, :, M - 1] = herm(0.05)
        if p == 0:
            W = W[0:1]
        if p == L - 1:
            W = W[:, :, :, M - 1 : M]
        cores.append(np.ascontiguousarray(W))
    return cores


def synthetic_liouvillian_mpo(L: int, M: int = 16, seed: int = 0, gamma: float = 0.02):
    """Non-Hermitian MPO on d = 4 = 2 x 2 sites with the structure of a
    vectorised Lindblad generator:
        H (x) 1  -  1 (x) H^T  -  i * sum_p Gamma_p
    H is a synthetic Hermitian spin-1/2 chain MPO of bond (M-2)/2, Gamma_p a
    random positive single-site damping matrix.  Bond dimension M (even, >= 6)."""
    if M % 2 or M < 6:
        raise ValueError("M must be even and >= 6")
    mh = (M - 2) // 2
    H = synthetic_mpo(L, 2, mh, seed=seed)
    rng = np.random.default_rng(seed + 1000)
    eye = np.eye(2)
    cores = []
    for p in range(L):
        w = H[p]
        a = np.einsum("cijt,kl->cikjlt", w, eye).reshape(w.shape[0], 4, 4, w.shape[3])       # H (x) 1
        b = np.einsum("cijt,kl->ckiljt", w.transpose(0, 2, 1, 3), eye).reshape(w.shape[0], 4, 4, w.shape[3])  # 1 (x) H^T
        if p == 0:
            b = -b
        G = rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4))
        damp = -1j * gamma * (G @ G.conj().T) / 4.0
        # local damping as a bond-2 MPO: [[1, damp], [0, 1]]
        loc = np.zeros((2, 4, 4, 2), dtype=np.complex128)
        loc[0, :, :, 0] = np.eye(4)
        loc[1, :, :, 1] = np.eye(4)
        loc[0, :, :, 1] = damp
        if p == 0:
            loc = loc[0:1]
        if p == L - 1:
            loc = loc[:, :, :, 1:2]
        parts = [a, b, loc]
        ml = 1 if p == 0 else sum(x.shape[0] for x in parts)
        mr = 1 if p == L - 1 else sum(x.shape[3] for x in parts)
        W = np.zeros((ml, 4, 4, mr), dtype=np.complex128)
        ro = co = 0
        for x in parts:
            r0 = 0 if p == 0 else ro
            c0 = 0 if p == L - 1 else co
            W[r0 : r0 + x.shape[0], :, :, c0 : c0 + x.shape[3]] += x
            ro += x.shape[0]
            co += x.shape[3]
        cores.append(W)
    return cores


def random_mps_cores(dims, D: int, seed: int = 1):
    """Full-rank random site tensors with the capped bond dimensions (not canonicalised: hand
    them to ``TDVPEngine.set_mps(..., canonicalize=True)``)."""
    rng = np.random.default_rng(seed)
    cores = []
    for (dl, dr), d in zip(bond_dims(list(dims), D), dims):
        c = rng.standard_normal((dl, d, dr)) + 1j * rng.standard_normal((dl, d, dr))
        cores.append(c / np.sqrt(2.0 * dl * d))
    return cores
