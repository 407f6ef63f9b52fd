"""Host-side reduction of PyTDSCF operator dictionaries to ONE full-chain MPO.

The reference keeps, per state pair, a dict of operators keyed by the legs they
act on (``TensorHamiltonian.mpo[i][j]``, hamiltonian_cls.py:669-752) with
diagonal 3-leg cores, 4-leg cores, and identity fill-ins for skipped sites
(``_mpo_cls.py:116-163``); its sweep then carries one environment block per
operator key plus a running "summed" block (_mps_mpo.py:535-598).

The MI355X engine instead contracts a single MPO whose bond space is the direct
sum of the terms' bond spaces: one large GEMM chain per apply instead of one
small chain per key.  The sum is exact (block structure, no truncation), so
H_eff, K_eff and every observable are unchanged up to rounding.
"""

from __future__ import annotations

import numpy as np


def as_four_leg(core: np.ndarray) -> np.ndarray:
    """(M_l, d, M_r) diagonal core -> (M_l, d, d, M_r); 4-leg cores pass through
    (OperatorCore.only_diag, _mpo_cls.py:196-198)."""
    core = np.asarray(core)
    if core.ndim == 4:
        return core.astype(np.complex128)
    if core.ndim == 3:
        ml, d, mr = core.shape
        out = np.zeros((ml, d, d, mr), dtype=np.complex128)
        idx = np.arange(d)
        out[:, idx, idx, :] = core
        return out
    raise ValueError(f"Invalid core shape {core.shape}")


def full_chain(cores, sites, dims) -> list[np.ndarray]:
    """Pad one operator term to all ``len(dims)`` sites with identity cores:
    bond 1 outside its span, carried bond inside gaps (_mpo_cls.py:153-163)."""
    sites = list(sites)
    if len(cores) != len(sites):
        raise ValueError("one core per acting site expected")
    if sorted(sites) != sites or len(set(sites)) != len(sites):
        raise ValueError("operator sites must be strictly ascending")
    by_site = {s: as_four_leg(c) for s, c in zip(sites, cores)}
    out = []
    bond = 1
    for p, d in enumerate(dims):
        if p in by_site:
            w = by_site[p]
            if w.shape[1] != d or w.shape[2] != d:
                raise ValueError(f"core at site {p} has physical dim {w.shape[1:3]}, basis has {d}")
            if w.shape[0] != bond:
                raise ValueError(f"MPO bond mismatch at site {p}: {w.shape[0]} vs {bond}")
            bond = w.shape[3]
        else:
            w = np.einsum("ab,ij->aijb", np.eye(bond), np.eye(d)).astype(np.complex128)
        out.append(w)
    if bond != 1:
        raise ValueError("last core of an operator term must close the MPO bond (M_r = 1)")
    return out


def merge_operator_terms(terms, dims) -> list[np.ndarray]:
    """Direct sum of operator terms -> one MPO.

    terms: iterable of ``(cores, sites)``.  Returns 4-leg complex128 cores with
    bond dimensions sum_k M_k (1 at the chain ends).
    """
    chains = [full_chain(c, s, dims) for c, s in terms]
    if not chains:
        raise ValueError("no operator terms")
    n = len(dims)
    if len(chains) == 1:
        return chains[0]
    out = []
    for p, d in enumerate(dims):
        ml = 1 if p == 0 else sum(ch[p].shape[0] for ch in chains)
        mr = 1 if p == n - 1 else sum(ch[p].shape[3] for ch in chains)
        w = np.zeros((ml, d, d, mr), dtype=np.complex128)
        ro = co = 0
        for ch in chains:
            c = ch[p]
            r0 = 0 if p == 0 else ro
            c0 = 0 if p == n - 1 else co
            w[r0 : r0 + c.shape[0], :, :, c0 : c0 + c.shape[3]] += c
            ro += c.shape[0]
            co += c.shape[3]
        out.append(w)
    return out


def compress_mpo(mpo, tol: float = 1.0e-13) -> list[np.ndarray]:
    """Tensor-train rounding of a full-chain 4-leg MPO: right-to-left QR sweep, then
    left-to-right SVDs dropping singular values below ``tol`` times the largest one.  With the
    default tolerance this only removes exact linear dependencies (the direct sum of operator
    terms repeats identity channels), i.e. the operator is unchanged to rounding while the bond
    dimension M -- the cost of an apply grows like M^2 -- shrinks.  The reference's counterpart
    are the SVD sweeps of ``_mpo_cls.py:601-912`` (``sweep_compress_twodot``)."""
    cs = [np.array(w, dtype=np.complex128) for w in mpo]
    shp = [c.shape for c in cs]
    cs = [c.reshape(c.shape[0], c.shape[1] * c.shape[2], c.shape[3]) for c in cs]
    for i in range(len(cs) - 1, 0, -1):
        r, n, rr = cs[i].shape
        q, t = np.linalg.qr(cs[i].reshape(r, n * rr).T)
        cs[i] = q.T.reshape(-1, n, rr)
        cs[i - 1] = np.tensordot(cs[i - 1], t.T, axes=(2, 0))
    for i in range(len(cs) - 1):
        r, n, rr = cs[i].shape
        u, sv, vh = np.linalg.svd(cs[i].reshape(r * n, rr), full_matrices=False)
        k = max(int((sv > tol * sv[0]).sum()), 1) if sv[0] > 0 else 1
        cs[i] = u[:, :k].reshape(r, n, k)
        cs[i + 1] = np.tensordot(sv[:k, None] * vh[:k], cs[i + 1], axes=(1, 0))
    return [np.ascontiguousarray(c.reshape(c.shape[0], s[1], s[2], c.shape[2])) for c, s in zip(cs, shp)]


def mpo_to_dense(mpo) -> np.ndarray:
    """Dense matrix of a (small) MPO -- test helper."""
    t = mpo[0]
    for w in mpo[1:]:
        t = np.tensordot(t, w, axes=(t.ndim - 1, 0))
    t = t[0, ..., 0]  # drop the boundary bonds
    n = len(mpo)
    perm = [2 * i for i in range(n)] + [2 * i + 1 for i in range(n)]
    t = t.transpose(perm)
    dim = int(np.prod(t.shape[:n]))
    return t.reshape(dim, dim)
