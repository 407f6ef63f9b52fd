"""Setup helper of the Kraus-map path: Lindblad operators -> Kraus tensor for one time step
(``pytdscf/kraus.py:17-123``; Werner et al., PRL 116, 237201 (2016)).  Host-side, runs once per
model; the maps themselves are applied on the device (``mitdvp_set_kraus``)."""

from __future__ import annotations

from math import isqrt

import numpy as np
import scipy.linalg


def lindblad_to_kraus(Lops, dt: float, backend: str = "numpy", tol: float = 1.0e-14) -> np.ndarray:
    """Kraus tensor B (k, d, d) with exp(D dt) = sum_q B_q (x) conj(B_q) for the dissipator
    D = sum_j [ L_j (x) conj(L_j) - (L_j^+ L_j (x) 1 + 1 (x) L_j^T conj(L_j)) / 2 ]
    (row-major vectorisation of the density matrix).  The Kraus operators are the scaled
    eigenvectors of the Choi matrix of the one-step channel with eigenvalue > ``tol``; the set
    is fixed up to a unitary mixing of the index q, which does not change the channel."""
    Lops = [np.asarray(L) for L in Lops]
    if not Lops or any(L.ndim != 2 or L.shape[0] != L.shape[1] for L in Lops):
        raise ValueError("Lindblad operators must be square matrices")
    if not dt > 0:
        raise ValueError("dt must be positive")
    if backend not in ("numpy", "hip", "jax"):
        raise ValueError(f"Invalid backend: {backend}")
    d = Lops[0].shape[0]
    eye = np.eye(d)
    D = np.zeros((d * d, d * d), dtype=np.complex128)
    for L in Lops:
        LdL = L.conj().T @ L
        D += np.kron(L, L.conj()) - 0.5 * (np.kron(LdL, eye) + np.kron(eye, LdL.T))
    G = scipy.linalg.expm(D * dt)  # rho'[a,b] = sum G[(a,b),(m,n)] rho[m,n]
    if isqrt(G.shape[0]) != d:
        raise ValueError("internal: dissipator shape")
    choi = G.reshape(d, d, d, d).transpose(0, 2, 1, 3).reshape(d * d, d * d)  # J[(a,m),(b,n)]
    choi = 0.5 * (choi + choi.conj().T)
    w, V = np.linalg.eigh(choi)
    if w.min() < -1.0e-12:
        raise ValueError(f"the one-step map is not completely positive (Choi eigenvalue {w.min()})")
    keep = [q for q in range(len(w)) if w[q] > tol]
    B = np.stack([np.sqrt(w[q]) * V[:, q].reshape(d, d) for q in keep], axis=0).astype(np.complex128)
    recon = sum(np.kron(b, b.conj()) for b in B)
    if np.abs(recon - G).max() > 1.0e-12:
        raise ValueError("Kraus decomposition does not reproduce exp(D dt)")
    return B
