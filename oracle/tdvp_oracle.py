"""CPU oracle for the one-site TDVP sweep hot path (NumPy restatement).

TEST INFRASTRUCTURE -- NOT THE PRODUCT.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module, and only as the checker / the timed CPU baseline.  The
product path (``pytdscf_amd`` -> ``libmitdvp.so``) never routes through it.

Parity status: PINNED.  Every function below is checked in
``tests/test_oracle_golden.py`` against golden vectors that were produced by
running the reference itself (``/root/reference``, PyTDSCF v1.3.3, NumPy
backend) in the development container with ``tests/golden/make_golden.py``.

Beyond the sweep itself (rows a1-a11) the module restates, each pinned the same way: SVD bond
truncation, Liouville-space traces, adaptive bond dimension (a1TDVP), one-site gates and Kraus
maps between the half-sweeps, ``Simulator.operate``, several electronic states
(:class:`OracleMultiMPS`).  ``tests/golden/crosscheck_reference.py``
additionally runs reference and oracle side by side on combinations of these features.

All ``file:line`` citations are relative to ``/root/reference/pytdscf``.

Conventions:
    site tensor  psi[b, j, s]     shape (D_l, d, D_r), complex128, C order
    left env     L[a, c, b]       (bra D_l, MPO bond M_l, ket D_l)
    right env    R[r, t, s]       (bra D_r, MPO bond M_r, ket D_r)
    MPO core     W[c, i, j, t]    (M_l, d_out(bra), d_in(ket), M_r)
The oracle only knows ONE full-chain 4-leg MPO (per pair of electronic states); the shell in
``pytdscf_amd.operators`` reduces the reference's operator dictionaries
(several keys, diagonal 3-leg cores, identity fill-ins, coupleJ) to that form
by an exact MPO direct sum, see DESIGN.md.
"""

from __future__ import annotations

import cmath
import math
from dataclasses import dataclass, field

import numpy as np
import scipy.linalg

EPS = 1e-12  # _integrator.py:20-ish: breakdown threshold "EPS"
MAX_KRYLOV = 20  # _integrator.py:182


# --------------------------------------------------------------------------
# a1: bond dimensions and initial canonical form
# --------------------------------------------------------------------------
def bond_dims(dims: list[int], m_aux_max: int) -> list[tuple[int, int]]:
    """(D_l, D_r) per site, ``LatticeInfo.get_bond_dim`` (_mps_cls.py:2616-2631)."""
    nsite = len(dims)
    out = []
    for isite in range(nsite):
        dim_left = 1 if isite == 0 else min(m_aux_max, math.prod(dims[:isite]))
        dim_right = (
            1 if isite == nsite - 1 else min(m_aux_max, math.prod(dims[isite + 1 :]))
        )
        dc = dims[isite]
        out.append(
            (min(dim_left, dc * dim_right, m_aux_max), min(dim_left * dc, dim_right, m_aux_max))
        )
    return out


def qr_psi2Asigma(psi: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """"Psi2Asigma": psi.reshape(D_l*d, D_r) = Q R; A = Q, sigma = R.

    _site_cls.py:278-291 (scipy.linalg.qr, mode="economic").
    """
    dl, d, dr = psi.shape
    q, r = scipy.linalg.qr(psi.reshape(dl * d, dr), mode="economic")
    return q.reshape(dl, d, -1), r


def qr_psi2sigmaB(psi: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """"Psi2sigmaB": QR of psi.transpose(2,1,0).reshape(D_r*d, D_l).

    sigma = R^T, B = Q.reshape(D_r, d, k).transpose(2,1,0); _site_cls.py:257-273.
    Returns (sigma, B).
    """
    dl, d, dr = psi.shape
    q, r = scipy.linalg.qr(
        np.ascontiguousarray(psi.transpose(2, 1, 0).reshape(dr * d, dl)), mode="economic"
    )
    return r.T, q.reshape(dr, d, -1).transpose(2, 1, 0)


def canonicalize_site0(cores: list[np.ndarray], scale: float | None = 1.0) -> list[np.ndarray]:
    """Right->left QR sweep so sites 1.. are "B" and site 0 is "Psi".

    Tail of ``alloc_superblock_random`` (_mps_cls.py:2684-2699): C2sigmaB on
    every site from the right, sigma absorbed into the left neighbour, site 0
    scaled to ``scale`` (Hilbert space).
    """
    cores = [np.array(c, dtype=np.complex128) for c in cores]
    for isite in range(len(cores) - 1, 0, -1):
        sval, matB = qr_psi2sigmaB(cores[isite])
        cores[isite] = np.ascontiguousarray(matB)
        cores[isite - 1] = np.tensordot(cores[isite - 1], sval, axes=(2, 0))
    if scale is not None:  # Hilbert space; in Liouville space the state keeps its (trace) normalisation, :2695-2699
        cores[0] = cores[0] * (scale / np.linalg.norm(cores[0]))
    return cores


# --------------------------------------------------------------------------
# a3: environment update
# --------------------------------------------------------------------------
def env_update_left(L: np.ndarray, A: np.ndarray, W: np.ndarray, bra: np.ndarray | None = None) -> np.ndarray:
    """L'[i,q,j] = sum conj(bra)[m,r,i] A[n,s,j] L[m,p,n] W[p,r,s,q]  (bra = A unless given).

    "mri,nsj,mpn,prsq->iqj", _contraction.py:286-297 (gauge "A", modes 3,2),
    called from renormalize_op_psite (_mps_mpo.py:554).  GEMM order
    L.A -> W -> conj(A) so that it is also a fair zgemm CPU baseline.  A
    separate (wider) ``bra`` tensor is the adaptive-rank case
    (superblock_states_bra, _mps_cls.py:1950-1963): L is then (bra bond, M, ket bond).
    """
    bra = A if bra is None else bra
    Db, M, Dk = L.shape
    _, d, Dko = A.shape
    Dbo = bra.shape[2]
    Mr = W.shape[3]
    # X[m,p,s,j] = sum_n L[m,p,n] A[n,s,j]
    X = (L.reshape(Db * M, Dk) @ A.reshape(Dk, d * Dko)).reshape(Db, M * d, Dko)
    # Y[m,(r,q),j] = sum_(p,s) W2[(r,q),(p,s)] X[m,(p,s),j]
    W2 = W.transpose(1, 3, 0, 2).reshape(d * Mr, M * d)
    Y = np.matmul(W2, X)  # (Db, d*Mr, Dko)
    # L'[i,(q,j)] = sum_(m,r) conj(bra)[(m,r),i] Y[(m,r),(q,j)]
    out = bra.reshape(Db * d, Dbo).conj().T @ Y.reshape(Db * d, Mr * Dko)
    return out.reshape(Dbo, Mr, Dko)


def env_update_right(R: np.ndarray, B: np.ndarray, W: np.ndarray, bra: np.ndarray | None = None) -> np.ndarray:
    """R'[i,p,j] = sum conj(bra)[i,r,m] B[j,s,n] R[m,q,n] W[p,r,s,q]  (bra = B unless given).

    "irm,jsn,mqn,prsq->ipj", _contraction.py:376-388 (gauge "B" mirror case).
    Evaluated as the mirror image of :func:`env_update_left`.
    """
    Bt = np.ascontiguousarray(B.transpose(2, 1, 0))  # [n,s,j]
    Wt = np.ascontiguousarray(W.transpose(3, 1, 2, 0))  # [q,r,s,p]
    brat = None if bra is None else np.ascontiguousarray(bra.transpose(2, 1, 0))
    return env_update_left(R, Bt, Wt, brat)


# --------------------------------------------------------------------------
# a4 / a5: effective Hamiltonian applies
# --------------------------------------------------------------------------
def heff_apply(L: np.ndarray, W: np.ndarray, R: np.ndarray, psi: np.ndarray) -> np.ndarray:
    """sigma[a,i,r] = sum L[a,c,b] W[c,i,j,t] R[r,t,s] psi[b,j,s].

    "bjs,acb,cijt,rts->air", _contraction.py:1154-1161 via
    multiplyH_MPS_direct_MPO.dot (_contraction.py:1182-1243).  The blocks may be
    rectangular (bra bond != ket bond: adaptive rank, tensor_shapes_out).
    """
    Dlo, Ml, Dli = L.shape
    Dro, Mr, Dri = R.shape
    d = psi.shape[1]
    X = (L.reshape(Dlo * Ml, Dli) @ psi.reshape(Dli, d * Dri)).reshape(Dlo, Ml * d, Dri)
    W2 = W.transpose(1, 3, 0, 2).reshape(d * Mr, Ml * d)
    Y = np.matmul(W2, X)  # [a,(i,t),s]
    out = Y.reshape(Dlo * d, Mr * Dri) @ R.reshape(Dro, Mr * Dri).T
    return out.reshape(Dlo, d, Dro)


def heff_apply_chunked(L, W, R, psi, chunk: int) -> np.ndarray:
    """:func:`heff_apply` evaluated in blocks of ``chunk`` left-bond rows so the
    stage intermediates stay bounded (used for the C4-size CPU baseline)."""
    out = np.empty_like(psi)
    for a0 in range(0, L.shape[0], chunk):
        Lr = L[a0 : a0 + chunk]
        na, Ml, Dl = Lr.shape
        Dr, Mr, _ = R.shape
        d = psi.shape[1]
        X = (Lr.reshape(na * Ml, Dl) @ psi.reshape(Dl, d * Dr)).reshape(na, Ml * d, Dr)
        W2 = W.transpose(1, 3, 0, 2).reshape(d * Mr, Ml * d)
        Y = np.matmul(W2, X)
        out[a0 : a0 + na] = (Y.reshape(na * d, Mr * Dr) @ R.reshape(Dr, Mr * Dr).T).reshape(na, d, Dr)
    return out


def keff_apply(L: np.ndarray, R: np.ndarray, sigma: np.ndarray) -> np.ndarray:
    """sigma'[a,r] = sum L[a,c,b] sigma[b,s] R[r,c,s].

    "bs,acb,rcs->ar", _contraction.py:1339-1352 via multiplyK_MPS_direct_MPO.dot.
    """
    Dlo, M, Dli = L.shape
    Dro, _, Dri = R.shape
    X = L.reshape(Dlo * M, Dli) @ sigma  # [(a,c),s]
    return X.reshape(Dlo, M * Dri) @ R.reshape(Dro, M * Dri).T


# --------------------------------------------------------------------------
# a6 / a7: local propagators
# --------------------------------------------------------------------------
def _n_warmup(size: int, k_prev: int) -> int:
    """_iter_info, _integrator.py:178-186."""
    return min(size, min(max(0, k_prev - 2), 15))


def sil_lanczos(scale, matvec, psi, thresh=1e-9, k_prev=0, conserve_norm=True, size=None):
    """exp(scale*H) psi by the reference's short-iterative Lanczos.

    Follows _integrator.py:453-655 statement by statement, including the
    reference's non-textbook alpha_l = <v0|H|v_l> (``v0_conj`` fixed,
    :535, :556), the warm-up that skips eigen-decompositions
    (:578-579), host ``eigh_tridiagonal`` for real alpha (:617-621), dense
    ``eig``+``solve`` otherwise (:622-633), the successive-approximant
    convergence test (:644-652) and ``_normalize/_rescale`` (:189-213).
    Returns (psi_new, k) where k is the Krylov dimension stored in
    ``_Debug.niter_krylov`` (:641, :649).  Raises ValueError like :653.
    """
    shape = psi.shape
    v0 = np.array(psi, dtype=np.complex128).reshape(-1)
    # _iter_info counts the UNPADDED input tensor (adaptive rank: psi is zero-padded, _integrator.py:178-186)
    size = v0.size if size is None else size
    ndim = min(size, MAX_KRYLOV)
    n_warm = _n_warmup(size, k_prev)
    if conserve_norm:
        beta0 = 1.0
    else:
        beta0 = float(np.linalg.norm(v0))
        if beta0 == 0.0:
            raise ValueError("Initial psi has zero norm.")
        v0 = v0 / beta0
    v0_conj = np.conj(v0)
    V = [v0]
    alpha: list[complex] = []
    beta: list[float] = []
    alpha_is_real = True
    psi_sv = None
    beta_l = 0.0
    for ldim in range(ndim):
        trial = psi if ldim == 0 else V[-1].reshape(shape)
        v_l = np.array(matvec(trial)).reshape(-1)
        if not conserve_norm and ldim == 0:
            v_l = v_l / beta0
        a_l = complex(np.inner(v0_conj, v_l))
        alpha.append(a_l)
        v_l = v_l - V[-1] * a_l
        if ldim > 0:
            v_l = v_l - V[-2] * beta_l
        beta_l = float(np.linalg.norm(v_l))
        beta.append(beta_l)
        if beta_l >= EPS:
            v_l = v_l / beta_l
        V.append(v_l)
        is_converged = beta_l < EPS or ldim + 1 == size
        if alpha_is_real and abs(a_l.imag) > 1e-10:
            alpha_is_real = False
        if ldim < n_warm and not is_converged:
            continue
        if ldim == 0:
            psi_next = v0 * cmath.exp(scale * alpha[-1])
        else:
            if alpha_is_real:
                lam, phi = scipy.linalg.eigh_tridiagonal(np.real(alpha), beta[:-1])
                coef = phi @ (np.exp(scale * lam) * np.conjugate(phi).T[:, 0])
            else:
                mat = (
                    np.diag(alpha, 0)
                    + np.diag(beta[:-1], -1).astype(np.complex128)
                    + np.diag(beta[:-1], 1).astype(np.complex128)
                )
                lam, phi = scipy.linalg.eig(mat)
                e0 = np.zeros(ldim + 1, dtype=mat.dtype)
                e0[0] = 1
                coef = phi @ (np.exp(scale * lam) * np.linalg.solve(phi, e0))
            psi_next = np.dot(coef, np.array(V[:-1]))
        done = is_converged
        if not done:
            if psi_sv is not None and float(np.linalg.norm(psi_next - psi_sv)) < thresh:
                done = True
            psi_sv = psi_next
        if done:
            if conserve_norm:
                psi_next = psi_next / float(np.linalg.norm(psi_next))
            else:
                psi_next = psi_next * beta0
            return psi_next.reshape(shape), ldim + 1
    raise ValueError(
        f"Short Iterative Lanczos is not converged in {ndim} basis. Try shorter time interval."
    )


def sil_arnoldi(scale, matvec, psi, thresh=1e-9, k_prev=0, conserve_norm=True, size=None):
    """exp(scale*H) psi by short-iterative Arnoldi, _integrator.py:287-432.

    Classical Gram-Schmidt against all previous vectors (_orth_step_np,
    :247-260), (k+1) x k Hessenberg, ``eig`` + ``solve(eigvecs, e0)``
    (:401-409), same warm-up / convergence / rescale rules as Lanczos.
    """
    shape = psi.shape
    v0 = np.array(psi, dtype=np.complex128).reshape(-1)
    # _iter_info counts the UNPADDED input tensor (adaptive rank: psi is zero-padded, _integrator.py:178-186)
    size = v0.size if size is None else size
    ndim = min(size, MAX_KRYLOV)
    n_warm = _n_warmup(size, k_prev)
    hessen = np.zeros((ndim + 1, ndim), dtype=np.complex128)
    if conserve_norm:
        beta0 = 1.0
    else:
        beta0 = float(np.linalg.norm(v0))
        if beta0 == 0.0:
            raise ValueError("Initial psi has zero norm.")
        v0 = v0 / beta0
    V = [v0]
    psi_sv = None
    v = v0
    for ldim in range(ndim):
        trial = psi if ldim == 0 else v.reshape(shape)
        v_l = np.array(matvec(trial)).reshape(-1)
        if not conserve_norm and ldim == 0:
            v_l = v_l / beta0
        Vm = np.array(V)
        hcol = np.sum(np.conj(Vm) * v_l[np.newaxis, :], axis=1)
        v_l = v_l - np.sum(hcol[:, np.newaxis] * Vm, axis=0)
        beta = float(np.linalg.norm(v_l))
        hessen[: ldim + 1, ldim] = hcol
        if beta > EPS:
            v_l = v_l / beta
            V.append(v_l)
            if hessen.shape[0] > ldim + 1:
                hessen[ldim + 1, ldim] = beta
        v = v_l
        is_converged = beta < EPS or ldim + 1 == size
        if ldim < n_warm and not is_converged:
            continue
        if ldim == 0:
            psi_next = v0 * cmath.exp(scale * hessen[0, 0])
        else:
            subH = hessen[: ldim + 1, : ldim + 1]
            lam, vec = np.linalg.eig(subH)
            e0 = np.zeros(ldim + 1, dtype=subH.dtype)
            e0[0] = 1
            coef = vec @ (np.exp(scale * lam) * np.linalg.solve(vec, e0))
            psi_next = np.tensordot(coef, np.array(V[: ldim + 1]), axes=(0, 0))
        done = is_converged
        if not done:
            if psi_sv is not None and float(np.linalg.norm(psi_next - psi_sv)) < thresh:
                done = True
            psi_sv = psi_next
        if done:
            if conserve_norm:
                psi_next = psi_next / float(np.linalg.norm(psi_next))
            else:
                psi_next = psi_next * beta0
            return psi_next.reshape(shape), ldim + 1
    raise ValueError("Short Iterative Arnoldi is not converged in 20 basis.")


def lanczos_ground_state(matvec, psi, thresh=1e-9, root=0):
    """Lowest eigenvector of a Hermitian H_eff by Lanczos, the "improved
    relaxation" local solver ``matrix_diagonalize_lanczos`` (_integrator.py:74-138):
    orthodox Lanczos (alpha_l = Re<v_l|H|v_l>, :115), a tridiagonal ``eigh`` after
    every new vector (:123), convergence on the change of the Ritz vector (:131-134).
    Returns (psi_new, k)."""
    shape = psi.shape
    v = np.array(psi, dtype=np.complex128).reshape(-1)
    ndim = v.size
    n_iter = min(ndim, 3000)
    alpha = np.array([], dtype=np.float64)
    beta = np.array([0.0], dtype=np.float64)
    V = [v]
    psi_sv = None
    for i in range(n_iter + 1):
        sig = np.array(matvec(V[-1].reshape(shape))).reshape(-1)
        alpha = np.append(alpha, np.inner(np.conj(V[-1]), sig).real)
        sig = sig - V[-1] * alpha[-1]
        if len(V) >= 2:
            sig = sig - V[-2] * beta[-1]
        beta = np.append(beta, np.linalg.norm(sig))
        sig = sig / beta[-1] if beta[-1] != 0 else sig
        _, vecs = scipy.linalg.eigh_tridiagonal(alpha, beta[1:-1])
        psi_next = (np.array(V).T @ vecs[:, root].reshape(i + 1, 1)).reshape(ndim)
        if abs(beta[-1]) < EPS:
            return psi_next.reshape(shape), i + 1
        if i == 0:
            psi_sv = psi_next
        else:
            if np.linalg.norm(psi_next - psi_sv) < thresh or i == ndim:
                return psi_next.reshape(shape), i + 1
            psi_sv = psi_next
        V.append(sig)
    raise ValueError("Lanczos Diagonalization is not converged in 3000 basis")


# --------------------------------------------------------------------------
# f2: adaptive bond dimension (a1TDVP) helpers
# --------------------------------------------------------------------------
def thin_to_full(core: np.ndarray, gauge: str, delta_rank: int) -> np.ndarray:
    """Isometry + ``delta_rank`` further orthonormal columns (rows) of its
    orthogonal complement, SiteCoef.thin_to_full (_site_cls.py:294-405): the
    trailing columns of LAPACK's full QR of the isometry itself; the leading
    ones are sign-aligned with the input, i.e. they ARE the input."""
    l, c, r = core.shape
    if gauge == "A":
        dr = min(delta_rank, l * c - r)
        mat = core.reshape(l * c, r)
        Q, _ = scipy.linalg.qr(mat, mode="full")
        ip = mat.T.conj() @ Q
        unflip = np.sign(np.sign(np.diag(ip[:r, :r])) + 0.5)
        Q = Q[:, : r + dr].copy()
        Q[:, :r] *= unflip[np.newaxis, :]
        return Q.reshape(l, c, r + dr)
    dl = min(delta_rank, c * r - l)
    mat = np.ascontiguousarray(core.reshape(l, c * r).T)
    Q, _ = scipy.linalg.qr(mat, mode="full")
    ip = mat.T.conj() @ Q
    unflip = np.sign(np.sign(np.diag(ip[:l, :l])) + 0.5)
    Q = Q[:, : l + dl].copy()
    Q[:, :l] *= unflip[np.newaxis, :]
    return np.ascontiguousarray(Q.T).reshape(l + dl, c, r)


def actual_delta_rank(cores: list[np.ndarray], isite: int, gauge: str, delta_rank: int) -> int:
    """get_actual_delta_rank (_mps_cls.py:3723-3755)."""
    nsite = len(cores)
    l1, c1, r1 = cores[isite].shape
    if gauge == "A":
        if isite == nsite - 1:
            return 0
        l2, c2, r2 = cores[isite + 1].shape
        return max(min(delta_rank, min(l1 * c1 - r1, c2 * r2 - l2)), 0)
    if isite == 0:
        return 0
    l2, c2, r2 = cores[isite - 1].shape
    return max(min(delta_rank, min(c1 * r1 - l1, l2 * c2 - r2)), 0)


def superblock_full(cores: list[np.ndarray], center: int, delta_rank: int) -> list[np.ndarray]:
    """get_superblock_full (_mps_cls.py:3699-3720): sites left of the centre are
    "A", right of it "B", the centre is copied."""
    out = []
    for i, c in enumerate(cores):
        if i == center:
            out.append(c.copy())
        else:
            g = "A" if i < center else "B"
            out.append(thin_to_full(c, g, actual_delta_rank(cores, i, g, delta_rank)))
    return out


def select_rank(h_left: np.ndarray, k_sig: np.ndarray, h_right: np.ndarray, dmin: int, dmax: int, p: float):
    """The D loop of get_rank_and_projection_error (_mps_cls.py:2083-2105):
    f(D) = |H psi_left[..., :D]|^2 - |K sigma[:D, :D]|^2 + |H psi_right[:D, ...]|^2,
    stop at the first D whose relative increment falls below p."""
    prev = 0.0
    D = dmin
    for D in range(dmin, dmax + 1):
        a = h_left[:, :, :D].ravel()
        b = h_right[:D, :, :].ravel()
        k = k_sig[:D, :D].ravel()
        tot = float(np.vdot(a, a).real) - float(np.vdot(k, k).real) + float(np.vdot(b, b).real)
        if D > dmin:
            metric = (tot - prev) / tot
            if metric < p:
                return D - 1, metric
        prev = tot
    return max(dmin, D), 0.0


# --------------------------------------------------------------------------
# f3: one-site gates between the half-sweeps (apply_one_gate)
# --------------------------------------------------------------------------
def canonicalize_A(cores: list[np.ndarray], lo: int, hi: int) -> None:
    """canonicalizeA on cores[lo..hi] (_mps_cls.py:3539-3567): A(lo) .. A(hi-1) Psi(hi), in place."""
    sval = None
    for i in range(lo, hi + 1):
        if sval is not None:
            cores[i] = np.tensordot(sval, cores[i], axes=(1, 0))
        if i != hi:
            cores[i], sval = qr_psi2Asigma(cores[i])


def canonicalize_B(cores: list[np.ndarray], lo: int, hi: int) -> None:
    """canonicalizeB on cores[lo..hi] (_mps_cls.py:3570-3598): Psi(lo) B(lo+1) .. B(hi), in place."""
    sval = None
    for i in range(hi, lo - 1, -1):
        if sval is not None:
            cores[i] = np.tensordot(cores[i], sval, axes=(2, 0))
        if i != lo:
            sval, B = qr_psi2sigmaB(cores[i])
            cores[i] = np.ascontiguousarray(B)


def apply_one_gate(cores: list[np.ndarray], center: int, gates: dict, conj: bool = False, power: int = 1) -> None:
    """MPSCoef.apply_one_gate (_mps_cls.py:2314-2373, :2420-2451): U[d', b] on the
    physical leg of the listed sites, then re-orthogonalisation towards ``center``
    (which must be the current "Psi" site) over the span of the touched sites."""
    changed = []
    for isite, U in sorted(gates.items()):
        U = np.asarray(U, dtype=np.complex128)
        if U.ndim == 1:  # diagonal core, key (isite,)
            U = np.diag(U)
        if conj:
            U = U.conj()
        if power != 1:
            U = np.linalg.matrix_power(U, power)
        cores[isite] = np.einsum("abc,db->adc", cores[isite], U)
        if isite != center:
            changed.append(isite)
    if changed:
        if max(changed) > center:
            canonicalize_B(cores, center, max(changed))
        if min(changed) < center:
            canonicalize_A(cores, min(changed), center)


# --------------------------------------------------------------------------
# f3: Kraus maps on purified states between the half-sweeps (apply_kraus)
# --------------------------------------------------------------------------
def kraus_single_site(B: np.ndarray, A: np.ndarray) -> np.ndarray:
    """_kraus_contract_single_site_np (kraus.py:146-228): the site's physical index is
    (d, K) = (system, ancilla); C[(m,n,x),(k,K)] = sum_d B[k,x,d] A[m,d,K,n], the ancilla
    index (k,K) is cut back to K by an SVD (U S kept), result (m, x K, n)."""
    k, x, d = B.shape
    m, dK, n = A.shape
    K = dK // d
    C = np.einsum("kxd,mdKn->mnxkK", B, A.reshape(m, d, K, n)).reshape(m * n * x, k * K)
    U, S, _ = scipy.linalg.svd(C, full_matrices=False)
    out = U[:, :K] * S[np.newaxis, :K]
    return np.ascontiguousarray(out.reshape(m, n, x * K).swapaxes(1, 2))


def kraus_two_site(B: np.ndarray, A1: np.ndarray, A2: np.ndarray):
    """_kraus_contract_two_site_np (kraus.py:281-358): system site A1 (m,d,l), ancilla
    site A2 (l,K,n); the Kraus index is absorbed into the ancilla by one SVD, the two
    sites are split again by a second one (bond dimension l kept)."""
    k, x, d = B.shape
    m, _, l = A1.shape
    _, K, n = A2.shape
    C = np.einsum("kxd,mdl,lKn->mxnkK", B, A1, A2).reshape(m * x * n, k * K)
    U, S, _ = scipy.linalg.svd(C, full_matrices=False)
    U = U[:, :K] * S[np.newaxis, :K]
    C = np.ascontiguousarray(U.reshape(m, x, n, K).swapaxes(2, 3)).reshape(m * x, K * n)
    U, S, Vh = scipy.linalg.svd(C, full_matrices=False)
    return (U[:, :l] * S[np.newaxis, :l]).reshape(m, x, -1), Vh[:l].reshape(-1, K, n)


def apply_kraus(cores: list[np.ndarray], center: int, kraus: dict) -> None:
    """MPSCoef.apply_kraus (_mps_cls.py:2375-2418): keys (site,) or (site, site + 1)."""
    lo, hi = len(cores), -1
    for sites, B in kraus.items():
        B = np.asarray(B, dtype=np.complex128)
        if len(sites) == 1:
            cores[sites[0]] = kraus_single_site(B, cores[sites[0]])
        elif len(sites) == 2 and sites[0] + 1 == sites[1]:
            cores[sites[0]], cores[sites[1]] = kraus_two_site(B, cores[sites[0]], cores[sites[1]])
        else:
            raise ValueError(f"site_inds={sites} is not yet implemented")
        lo, hi = min(lo, sites[0]), max(hi, sites[-1])
    if hi > center:
        canonicalize_B(cores, center, hi)
    if lo < center:
        canonicalize_A(cores, lo, center)


# --------------------------------------------------------------------------
# a8-a10: sweep
# --------------------------------------------------------------------------
@dataclass
class OracleMPS:
    """Site-0-centred MPS + MPO + the environment cache handed between sweeps.

    ``cores[0]`` is "Psi", the rest "B" (_mps_cls.py:2684-2703).  ``envs`` is
    the reference's ``op_sys_sites`` (_mps_cls.py:848-861, :1006-1012) stored
    by bond: ``left[p]`` acts on bond (p-1|p), ``right[p]`` on bond (p|p+1).
    ``kprev`` mirrors ``_Debug.niter_krylov`` keyed by site (_helper.py:29).
    """

    cores: list[np.ndarray]
    mpo: list[np.ndarray]
    integrator: str = "lanczos"
    thresh: float = 1e-9
    conserve_norm: bool = True
    shift: complex = 0.0  # coupleJ[0][0] * ovlp term, _contraction.py:1200-1216
    relax: bool | str = False  # const.doRelax: True = exp(-H dt/2) / exp(+K dt/2) + renormalise
    #   (_mps_cls.py:1086-1094); "improved" = Lanczos ground state of H_eff, bond step skipped (:1078-1084, :1159-1160)
    gates: dict | None = None  # Model(one_gate_to_apply=...): {site: U (d x d) or diagonal (d,)}, applied between the half-sweeps
    kraus: dict | None = None  # Model(kraus_op=...): {(site,) | (site, site+1): B (k, d, d)}, after the gates
    adaptive: bool = False  # const.adaptive (_const_cls.py:120-124, :212-216)
    Dmax: int = 100
    dD: int = 10
    p_proj: float = 1e-4
    left: dict = field(default_factory=dict)
    right: dict = field(default_factory=dict)
    kprev: dict = field(default_factory=dict)
    n_apply: int = 0
    center: int = 0

    def __post_init__(self):
        self.cores = [np.array(c, dtype=np.complex128) for c in self.cores]
        self.mpo = [np.array(w, dtype=np.complex128) for w in self.mpo]
        self.nsite = len(self.cores)
        one = np.ones((1, 1, 1), dtype=np.complex128)  # construct_op_zerosite, _mps_mpo.py:364
        self.left[0] = one
        self.right[self.nsite - 1] = one

    # ---- environment construction (construct_op_sites, _mps_cls.py:1738-1796)
    def build_right_envs(self):
        for p in range(self.nsite - 1, 0, -1):
            self.right[p - 1] = env_update_right(self.right[p], self.cores[p], self.mpo[p])

    def _exp(self, scale, matvec, x, site, size=None):
        fn = sil_lanczos if self.integrator == "lanczos" else sil_arnoldi
        out, k = fn(
            scale, matvec, x, self.thresh, self.kprev.get(site, 0), self.conserve_norm, size
        )
        self.kprev[site] = k
        return out

    def _heff(self, p):
        L, W, R = self.left[p], self.mpo[p], self.right[p]

        def mv(x):
            self.n_apply += 1
            y = heff_apply(L, W, R, x)
            if self.shift != 0.0:
                y = y + self.shift * x
            return y

        return mv

    def _keff(self, L, R):
        def mv(x):
            y = keff_apply(L, R, x)
            if self.shift != 0.0:
                y = y + self.shift * x
            return y

        return mv

    def sweep(self, dt: float, forward: bool):
        """One half-sweep, propagate_along_sweep (_mps_cls.py:798-1014)."""
        n = self.nsite
        sites = range(0, n) if forward else range(n - 1, -1, -1)
        end = n - 1 if forward else 0
        full = superblock_full(self.cores, 0 if forward else n - 1, self.dD) if self.adaptive else None
        for p in sites:
            if self.adaptive and p != end and self._adaptive_site(p, dt, forward, full):
                continue
            # exp_superH_propagation_direct, _mps_cls.py:1016-1100 (:1070)
            zs = -1.0 if self.relax else -1.0j  # _mps_cls.py:1070 vs :1088
            if self.relax == "improved":
                new, k = lanczos_ground_state(self._heff(p), self.cores[p], self.thresh)
                self.kprev[p] = k
                self.cores[p] = new / np.linalg.norm(new)
            else:
                self.cores[p] = self._exp(zs * dt / 2, self._heff(p), self.cores[p], p)
            if p == end:
                break
            if forward:
                # trans_next_psite_AsigmaB, _mps_cls.py:1798-1850
                A, sval = qr_psi2Asigma(self.cores[p])
                self.cores[p] = A
                self.left[p + 1] = env_update_left(self.left[p], A, self.mpo[p])
                # exp_superK_propagation_direct, _mps_cls.py:1102-1170 (:1151)
                if self.relax != "improved":
                    sval = self._exp(
                        -zs * dt / 2, self._keff(self.left[p + 1], self.right[p]), sval, p
                    )
                # trans_next_psite_APsiB, _mps_cls.py:1172-1206
                self.cores[p + 1] = np.tensordot(sval, self.cores[p + 1], axes=(1, 0))
            else:
                sval, B = qr_psi2sigmaB(self.cores[p])
                self.cores[p] = np.ascontiguousarray(B)
                self.right[p - 1] = env_update_right(self.right[p], self.cores[p], self.mpo[p])
                if self.relax != "improved":
                    sval = self._exp(
                        -zs * dt / 2, self._keff(self.left[p], self.right[p - 1]), sval, p
                    )
                self.cores[p - 1] = np.tensordot(self.cores[p - 1], sval, axes=(2, 0))
        self.center = end

    # ---- f2: adaptive bond dimension ---------------------------------------
    def _adaptive_site(self, p: int, dt: float, forward: bool, full: list[np.ndarray]) -> bool:
        """One site of an adaptive half-sweep (const.adaptive branches of
        propagate_along_sweep, _mps_cls.py:863-987; get_adaptive_rank_and_block,
        :2152-2286).  Returns False when the bond is already at maximal rank
        (is_max_rank, :3757-3766): the caller then does the plain step."""
        if self.relax or self.integrator not in ("lanczos", "arnoldi"):
            raise NotImplementedError
        l, c, r = self.cores[p].shape
        W = self.mpo
        if forward:
            if l * c <= r or r >= self.Dmax:
                return False
            q = p + 1
            env_prev = self.right[q]  # thin block right of site p+1
            # get_op_block_full, modes "braket" and "bra" (:2111-2150)
            env_braket = env_update_right(env_prev, full[q], W[q])
            env_bra = env_update_right(env_prev, self.cores[q], W[q], bra=full[q])
            dmax = min(self.Dmax, full[q].shape[0])
            # get_psi_sigvec_psi_fullblock (:1921-1983)
            A, sig = qr_psi2Asigma(self.cores[p])
            psi_prime = np.tensordot(sig, self.cores[q], axes=(1, 0))
            A_full = thin_to_full(A, "A", dmax - r)
            sys_bra = env_update_left(self.left[p], A, W[p], bra=A_full)
            # get_rank_and_projection_error (:1985-2105)
            dmax = min(dmax, l * c, psi_prime.shape[1] * psi_prime.shape[2])
            newD = r
            if r != dmax:
                h_left = heff_apply(self.left[p], W[p], env_bra[:dmax], self.cores[p])
                h_right = heff_apply(sys_bra[:dmax], W[q], env_prev, psi_prime)
                k_sig = keff_apply(sys_bra[:dmax], env_bra[:dmax], sig)
                if self.shift != 0.0:  # coupleJ * ovlp: the overlap blocks <full|thin> are [1; 0] embeddings
                    h_left[:, :, :r] += self.shift * self.cores[p]
                    h_right[:r] += self.shift * psi_prime
                    k_sig[:r, :r] += self.shift * sig
                newD, _ = select_rank(h_left, k_sig, h_right, r, dmax, self.p_proj)
            env_D_bra = env_bra[:newD]
            env_D_braket = env_braket[:newD, :, :newD]
            self.cores[q] = np.ascontiguousarray(full[q][:newD])
            # exp_superH_propagation_direct with tensor_shapes_out=(l, c, newD): every apply
            # sees the vector truncated to the old shape (SplitStack.split(truncate=True),
            # _contraction.py:593-610; _integrator.py:368,382 / :524,543)
            Lb = self.left[p]

            def mv(v):
                x = v[:, :, :r]
                y = heff_apply(Lb, W[p], env_D_bra, x)
                if self.shift != 0.0:
                    y[:, :, :r] += self.shift * x
                return y

            x0 = np.zeros((l, c, newD), dtype=np.complex128)
            x0[:, :, :r] = self.cores[p]
            self.cores[p] = self._exp(-1.0j * dt / 2, mv, x0, p, l * c * r)
            hook = getattr(self, "site_hook", None)  # the junction update's regularisation (_mps_parallel.py:362-370)
            if hook is not None:
                self.cores[p] = hook(self.cores[p])
            A, sval = qr_psi2Asigma(self.cores[p])
            self.cores[p] = A
            self.left[q] = env_update_left(self.left[p], A, W[p])
            sval = self._exp(+1.0j * dt / 2, self._keff(self.left[q], env_D_braket), sval, p)
            self.cores[q] = np.tensordot(sval, self.cores[q], axes=(1, 0))
            return True
        if l >= c * r or l >= self.Dmax:
            return False
        q = p - 1
        env_prev = self.left[q]  # thin block left of site p-1
        env_braket = env_update_left(env_prev, full[q], W[q])
        env_bra = env_update_left(env_prev, self.cores[q], W[q], bra=full[q])
        dmax = min(self.Dmax, full[q].shape[2])
        sig, B = qr_psi2sigmaB(self.cores[p])
        B = np.ascontiguousarray(B)
        psi_prime = np.tensordot(self.cores[q], sig, axes=(2, 0))
        B_full = thin_to_full(B, "B", dmax - l)
        sys_bra = env_update_right(self.right[p], B, W[p], bra=B_full)
        dmax = min(dmax, psi_prime.shape[0] * psi_prime.shape[1], c * r)
        newD = l
        if l != dmax:
            h_left = heff_apply(env_prev, W[q], sys_bra[:dmax], psi_prime)
            h_right = heff_apply(env_bra[:dmax], W[p], self.right[p], self.cores[p])
            k_sig = keff_apply(env_bra[:dmax], sys_bra[:dmax], sig)
            if self.shift != 0.0:
                h_left[:, :, :l] += self.shift * psi_prime
                h_right[:l] += self.shift * self.cores[p]
                k_sig[:l, :l] += self.shift * sig
            newD, _ = select_rank(h_left, k_sig, h_right, l, dmax, self.p_proj)
        env_D_bra = env_bra[:newD]
        env_D_braket = env_braket[:newD, :, :newD]
        self.cores[q] = np.ascontiguousarray(full[q][:, :, :newD])
        Rb = self.right[p]

        def mv(v):
            x = v[:l]
            y = heff_apply(env_D_bra, W[p], Rb, x)
            if self.shift != 0.0:
                y[:l] += self.shift * x
            return y

        x0 = np.zeros((newD, c, r), dtype=np.complex128)
        x0[:l] = self.cores[p]
        self.cores[p] = self._exp(-1.0j * dt / 2, mv, x0, p, l * c * r)
        sval, B = qr_psi2sigmaB(self.cores[p])
        self.cores[p] = np.ascontiguousarray(B)
        self.right[q] = env_update_right(self.right[p], self.cores[p], W[p])
        sval = self._exp(+1.0j * dt / 2, self._keff(env_D_braket, self.right[q]), sval, p)
        self.cores[q] = np.tensordot(self.cores[q], sval, axes=(2, 0))
        return True

    def propagate(self, dt: float):
        """One time step = forward + backward half-sweep, MPSCoef.propagate
        (_mps_cls.py:452-503).  The first call builds all right environments
        (:835-843)."""
        if len(self.right) < self.nsite:
            self.build_right_envs()
        self.sweep(dt, True)
        if self.gates or self.kraus:  # _mps_cls.py:489-492; op_sys_sites = None -> all left blocks are rebuilt (:2370)
            if self.gates:
                apply_one_gate(self.cores, self.nsite - 1, self.gates)
            if self.kraus:
                apply_kraus(self.cores, self.nsite - 1, self.kraus)
            for p in range(self.nsite - 1):
                self.left[p + 1] = env_update_left(self.left[p], self.cores[p], self.mpo[p])
        self.sweep(dt, False)

    # ---- a11: observables -------------------------------------------------
    def norm(self) -> float:
        """pop_states / norm (_mps_cls.py:682-716): ||Psi(site 0)||."""
        return float(np.linalg.norm(self.cores[0]))

    def expectation(self, mpo: list[np.ndarray] | None = None) -> complex:
        """<Psi|O|Psi> at site 0 with fresh right environments
        (MPSCoef.expectation, _mps_cls.py:540-612; expectation_Op,
        _integrator.py:41-71)."""
        mpo = self.mpo if mpo is None else [np.asarray(w, dtype=np.complex128) for w in mpo]
        R = np.ones((1, 1, 1), dtype=np.complex128)
        for p in range(self.nsite - 1, 0, -1):
            R = env_update_right(R, self.cores[p], mpo[p])
        L = np.ones((1, 1, 1), dtype=np.complex128)
        sig = heff_apply(L, mpo[0], R, self.cores[0])
        if mpo is self.mpo and self.shift != 0.0:
            sig = sig + self.shift * self.cores[0]
        return complex(np.vdot(self.cores[0].reshape(-1), sig.reshape(-1)))

    def autocorr(self) -> complex:
        """<Psi^*|Psi> (no conjugation of the bra), the t/2 trick
        (wavefunction.py:226-257, properties.py:211-220)."""
        return overlap(self.cores, self.cores, conj_bra=False)


def overlap(bra: list[np.ndarray], ket: list[np.ndarray], conj_bra: bool = True) -> complex:
    """Transfer-matrix overlap of two MPS, left to right."""
    T = np.ones((1, 1), dtype=np.complex128)
    for b, k in zip(bra, ket):
        bb = np.conj(b) if conj_bra else b
        # T'[i,j] = sum_{m,n,s} bb[m,s,i] T[m,n] k[n,s,j]
        tmp = np.tensordot(T, k, axes=(1, 0))  # [m,s,j]
        T = np.tensordot(bb, tmp, axes=([0, 1], [0, 1]))
    return complex(T[0, 0])


# --------------------------------------------------------------------------
# several electronic states: one MPS per state, Hamiltonian blocks H[i][j]
# --------------------------------------------------------------------------
@dataclass
class OracleMultiMPS:
    """MPS-SM with ``nstate > 1``: ``cores[i]`` is the site-0-centred MPS of state i,
    ``mpo[i][j]`` the full-chain MPO between bra state i and ket state j (or None),
    ``coupleJ[i][j]`` the scalar term (it multiplies the bra/ket overlap blocks,
    _contraction.py:1200-1216; for i == j those are the identity).

    The local problems act on the states' centre tensors stacked into one vector
    (SplitStack.stack/split, _contraction.py:479-608): sigma_i = sum_j H_eff[i][j] psi_j
    with the blocks of the state pair (i, j) (multiplyH_MPS_direct_MPO.dot,
    _contraction.py:1182-1243; operators_for_superH, _mps_mpo.py:698-858).  Environment
    blocks are kept per state pair (renormalize_op_psite, _mps_mpo.py:421-696), the gauge
    move is a separate QR per state (trans_next_psite_AsigmaB, _mps_cls.py:1798-1850) and
    ``kprev`` is shared by all states (one Krylov solve per site).
    """

    cores: list[list[np.ndarray]]
    mpo: list[list[list[np.ndarray] | None]]
    coupleJ: list[list[complex]]
    integrator: str = "lanczos"
    thresh: float = 1e-9
    conserve_norm: bool = True
    relax: bool | str = False
    kprev: dict = field(default_factory=dict)
    center: int = 0

    def __post_init__(self):
        self.cores = [[np.array(c, dtype=np.complex128) for c in st] for st in self.cores]
        self.nstate = len(self.cores)
        self.nsite = len(self.cores[0])
        self.mpo = [
            [None if m is None else [np.array(w, dtype=np.complex128) for w in m] for m in row] for row in self.mpo
        ]
        self.pairs = [(i, j) for i in range(self.nstate) for j in range(self.nstate)]
        one = np.ones((1, 1, 1), dtype=np.complex128)
        # per pair: operator blocks ("op") and overlap blocks ("ov", i != j with a scalar term)
        self.left = {(k, ij): {0: one} for k in ("op", "ov") for ij in self.pairs}
        self.right = {(k, ij): {self.nsite - 1: one} for k in ("op", "ov") for ij in self.pairs}
        self._built = False

    def _chains(self, mpo=None, coupleJ=None):
        """(kind, (i, j), cores, factor) for every chain that enters H."""
        mpo = self.mpo if mpo is None else mpo
        cj = self.coupleJ if coupleJ is None else coupleJ
        out = []
        for i, j in self.pairs:
            if mpo[i][j] is not None:
                out.append(("op", (i, j), mpo[i][j], 1.0))
            if i != j and cj[i][j] != 0.0:
                d = [c.shape[1] for c in self.cores[j]]
                out.append(("ov", (i, j), [np.eye(n, dtype=np.complex128)[None, :, :, None] for n in d], cj[i][j]))
        return out

    def build_right_envs(self):
        for kind, (i, j), w, _ in self._chains():
            R = self.right[(kind, (i, j))]
            for p in range(self.nsite - 1, 0, -1):
                R[p - 1] = env_update_right(R[p], self.cores[j][p], w[p], bra=self.cores[i][p])
        self._built = True

    # ---- stacked vectors ---------------------------------------------------
    @staticmethod
    def _stack(xs):
        return np.concatenate([x.reshape(-1) for x in xs])

    @staticmethod
    def _split(v, shapes):
        out, o = [], 0
        for sh in shapes:
            n = int(np.prod(sh))
            out.append(v[o : o + n].reshape(sh))
            o += n
        return out

    def _heff(self, p, shapes):
        chains = self._chains()

        def mv(v):
            xs = self._split(v, shapes)
            ys = [np.zeros(sh, dtype=np.complex128) for sh in shapes]
            for i in range(self.nstate):
                if self.coupleJ[i][i] != 0.0:
                    ys[i] = ys[i] + self.coupleJ[i][i] * xs[i]
            for kind, (i, j), w, f in chains:
                ys[i] = ys[i] + f * heff_apply(self.left[(kind, (i, j))][p], w[p], self.right[(kind, (i, j))][p], xs[j])
            return self._stack(ys)

        return mv

    def _keff(self, pl, pr, shapes):
        chains = self._chains()

        def mv(v):
            xs = self._split(v, shapes)
            ys = [np.zeros(sh, dtype=np.complex128) for sh in shapes]
            for i in range(self.nstate):
                if self.coupleJ[i][i] != 0.0:
                    ys[i] = ys[i] + self.coupleJ[i][i] * xs[i]
            for kind, (i, j), _, f in chains:
                ys[i] = ys[i] + f * keff_apply(self.left[(kind, (i, j))][pl], self.right[(kind, (i, j))][pr], xs[j])
            return self._stack(ys)

        return mv

    def _exp(self, scale, matvec, xs, site):
        shapes = [x.shape for x in xs]
        fn = sil_lanczos if self.integrator == "lanczos" else sil_arnoldi
        # _iter_info sizes the Krylov space by the LARGEST state tensor, not the stack (_integrator.py:178-186)
        size = max(x.size for x in xs)
        out, k = fn(scale, matvec, self._stack(xs), self.thresh, self.kprev.get(site, 0), self.conserve_norm, size)
        self.kprev[site] = k
        return self._split(out, shapes)

    def sweep(self, dt: float, forward: bool):
        n, S = self.nsite, self.nstate
        end = n - 1 if forward else 0
        zs = -1.0 if self.relax else -1.0j
        for p in range(0, n) if forward else range(n - 1, -1, -1):
            xs = [self.cores[s][p] for s in range(S)]
            shapes = [x.shape for x in xs]
            if self.relax == "improved":
                new, k = lanczos_ground_state(self._heff(p, shapes), self._stack(xs), self.thresh)
                self.kprev[p] = k
                new = self._split(new / np.linalg.norm(new), shapes)
            else:
                new = self._exp(zs * dt / 2, self._heff(p, shapes), xs, p)
            for s in range(S):
                self.cores[s][p] = new[s]
            if p == end:
                break
            svals = []
            for s in range(S):
                if forward:
                    A, sv = qr_psi2Asigma(self.cores[s][p])
                    self.cores[s][p] = A
                else:
                    sv, B = qr_psi2sigmaB(self.cores[s][p])
                    self.cores[s][p] = np.ascontiguousarray(B)
                svals.append(sv)
            for kind, (i, j), w, _ in self._chains():
                if forward:
                    self.left[(kind, (i, j))][p + 1] = env_update_left(
                        self.left[(kind, (i, j))][p], self.cores[j][p], w[p], bra=self.cores[i][p]
                    )
                else:
                    self.right[(kind, (i, j))][p - 1] = env_update_right(
                        self.right[(kind, (i, j))][p], self.cores[j][p], w[p], bra=self.cores[i][p]
                    )
            if self.relax != "improved":
                shp = [x.shape for x in svals]
                mk = self._keff(p + 1, p, shp) if forward else self._keff(p, p - 1, shp)
                svals = self._exp(-zs * dt / 2, mk, svals, p)
            for s in range(S):
                if forward:
                    self.cores[s][p + 1] = np.tensordot(svals[s], self.cores[s][p + 1], axes=(1, 0))
                else:
                    self.cores[s][p - 1] = np.tensordot(self.cores[s][p - 1], svals[s], axes=(2, 0))
        self.center = end

    def propagate(self, dt: float):
        if not self._built:
            self.build_right_envs()
        self.sweep(dt, True)
        self.sweep(dt, False)

    # ---- observables (site-0 centred) ----------------------------------------
    def pop_states(self) -> list[float]:
        """pop_states (_mps_cls.py:682-703)."""
        return [float(np.linalg.norm(st[0]) ** 2) for st in self.cores]

    def norm(self) -> float:
        return math.sqrt(sum(self.pop_states()))

    def expectation(self, mpo=None, coupleJ=None) -> complex:
        """sum_ij <Psi_i|O_ij|Psi_j> with fresh right blocks (MPSCoef.expectation, _mps_cls.py:540-612)."""
        cj = self.coupleJ if coupleJ is None else coupleJ
        if mpo is not None and coupleJ is None:
            cj = [[0.0] * self.nstate for _ in range(self.nstate)]
        tot = 0.0 + 0.0j
        for i in range(self.nstate):
            if cj[i][i] != 0.0:
                tot += cj[i][i] * np.vdot(self.cores[i][0], self.cores[i][0])
        one = np.ones((1, 1, 1), dtype=np.complex128)
        for _, (i, j), w, f in self._chains(mpo, cj):
            R = one
            for p in range(self.nsite - 1, 0, -1):
                R = env_update_right(R, self.cores[j][p], w[p], bra=self.cores[i][p])
            sig = heff_apply(one, w[0], R, self.cores[j][0])
            tot += f * np.vdot(self.cores[i][0].reshape(-1), sig.reshape(-1))
        return complex(tot)

    def autocorr(self) -> complex:
        """sum_i <Psi_i^*|Psi_i> (only the diagonal pairs carry an "auto" block, _mps_mpo.py:386-395)."""
        return sum(overlap(st, st, conj_bra=False) for st in self.cores)


# --------------------------------------------------------------------------
# Simulator.operate: variational application of an MPO to the state
# --------------------------------------------------------------------------
def operate(cores0: list[np.ndarray], mpo: list[np.ndarray], maxstep: int = 10, conv_tol: float = 1.0e-8, shift: complex = 0.0):
    """WFunc.apply_dipole (wavefunction.py:303-351) for the MPS standard method:
    fit phi ~ O|psi_0> / ||O|psi_0>|| in the bond dimensions of psi_0 by sweeps in which every
    site tensor is replaced by the mixed-environment apply (bra = phi, ket = psi_0),
    MPSCoef.apply_dipole / apply_dipole_along_sweep / apply_superOp_direct
    (_mps_cls.py:421-450, :718-796, :2733-2778).  Returns (norm, cores of phi, iterations)."""
    n = len(cores0)
    ket = [np.array(c, dtype=np.complex128) for c in cores0]
    bra = [c.copy() for c in ket]
    mpo = [np.asarray(w, dtype=np.complex128) for w in mpo]
    # the scalar term coupleJ * ovlp (_contraction.py:1200-1216) goes through the OVERLAP blocks of
    # the bra / ket pair, which are not identities here (phi != psi_0): carry it as a second
    # operator chain with identity cores
    ops = [mpo] + ([[np.eye(c.shape[1], dtype=np.complex128)[None, :, :, None] for c in ket]] if shift != 0.0 else [])
    coef = [1.0, shift]
    one = np.ones((1, 1, 1), dtype=np.complex128)
    # construct_op_sites with superblock_states_ket: right blocks from the initial bra / ket pair
    right = [{n - 1: one} for _ in ops]
    for k, w in enumerate(ops):
        for p in range(n - 1, 0, -1):
            right[k][p - 1] = env_update_right(right[k][p], ket[p], w[p], bra=bra[p])
    left = [{0: one} for _ in ops]
    norm = 0.0

    def site(p):
        nonlocal norm
        y = sum(coef[k] * heff_apply(left[k][p], w[p], right[k][p], ket[p]) for k, w in enumerate(ops))
        norm = float(np.linalg.norm(y))
        bra[p] = y / norm

    it = 0
    for it in range(1, maxstep + 1):
        prev = [c.copy() for c in bra]
        for p in range(n):  # ->
            site(p)
            if p == n - 1:
                break
            bra[p], sv = qr_psi2Asigma(bra[p])
            bra[p + 1] = np.tensordot(sv, bra[p + 1], axes=(1, 0))
            ket[p], sv = qr_psi2Asigma(ket[p])
            ket[p + 1] = np.tensordot(sv, ket[p + 1], axes=(1, 0))
            for k, w in enumerate(ops):
                left[k][p + 1] = env_update_left(left[k][p], ket[p], w[p], bra=bra[p])
        for p in range(n - 1, -1, -1):  # <-
            site(p)
            if p == 0:
                break
            sv, B = qr_psi2sigmaB(bra[p])
            bra[p] = np.ascontiguousarray(B)
            bra[p - 1] = np.tensordot(bra[p - 1], sv, axes=(2, 0))
            sv, B = qr_psi2sigmaB(ket[p])
            ket[p] = np.ascontiguousarray(B)
            ket[p - 1] = np.tensordot(ket[p - 1], sv, axes=(2, 0))
            for k, w in enumerate(ops):
                right[k][p - 1] = env_update_right(right[k][p], ket[p], w[p], bra=bra[p])
        if abs(1 - abs(overlap(bra, prev))) < conv_tol:  # _is_converged, wavefunction.py:285-301
            break
    return norm, bra, it


def operate_multi(states0, mpo, coupleJ, maxstep: int = 10, conv_tol: float = 1.0e-8):
    """:func:`operate` for several electronic states: ``states0[i]`` the site-0-centred MPS of
    state i, ``mpo[i][j]`` / ``coupleJ[i][j]`` the operator blocks.  All states' site tensors are
    replaced together, sigma_i = sum_j O_ij psi0_j with the blocks of (bra = phi_i, ket = psi0_j),
    and normalised by the norm of the stack (apply_superOp_direct, _mps_cls.py:2733-2778); the
    scalar terms go through the overlap blocks of the pair, which are not identities here even
    for i == j.  Returns (norm, states of phi, iterations)."""
    S, n = len(states0), len(states0[0])
    ket = [[np.array(c, dtype=np.complex128) for c in st] for st in states0]
    bra = [[c.copy() for c in st] for st in ket]
    chains = []  # (i, j, cores, factor)
    for i in range(S):
        for j in range(S):
            if mpo[i][j] is not None:
                chains.append((i, j, [np.asarray(w, dtype=np.complex128) for w in mpo[i][j]], 1.0))
            if coupleJ[i][j] != 0.0:
                chains.append((i, j, [np.eye(c.shape[1], dtype=np.complex128)[None, :, :, None] for c in ket[j]], coupleJ[i][j]))
    one = np.ones((1, 1, 1), dtype=np.complex128)
    right = [{n - 1: one} for _ in chains]
    left = [{0: one} for _ in chains]
    for k, (i, j, w, _) in enumerate(chains):
        for p in range(n - 1, 0, -1):
            right[k][p - 1] = env_update_right(right[k][p], ket[j][p], w[p], bra=bra[i][p])
    norm = 0.0

    def site(p):
        nonlocal norm
        ys = [None] * S
        for k, (i, j, w, f) in enumerate(chains):
            y = f * heff_apply(left[k][p], w[p], right[k][p], ket[j][p])
            ys[i] = y if ys[i] is None else ys[i] + y
        if any(y is None for y in ys):  # the reference fails on a state no block feeds (_contraction.py:555)
            raise ValueError("operate: every state needs at least one operator block or scalar term acting into it")
        norm = math.sqrt(sum(float(np.linalg.norm(y)) ** 2 for y in ys))
        for i in range(S):
            bra[i][p] = ys[i] / norm

    it = 0
    for it in range(1, maxstep + 1):
        prev = [[c.copy() for c in st] for st in bra]
        for p in range(n):  # ->
            site(p)
            if p == n - 1:
                break
            for st in (bra, ket):
                for s in range(S):
                    st[s][p], sv = qr_psi2Asigma(st[s][p])
                    st[s][p + 1] = np.tensordot(sv, st[s][p + 1], axes=(1, 0))
            for k, (i, j, w, _) in enumerate(chains):
                left[k][p + 1] = env_update_left(left[k][p], ket[j][p], w[p], bra=bra[i][p])
        for p in range(n - 1, -1, -1):  # <-
            site(p)
            if p == 0:
                break
            for st in (bra, ket):
                for s in range(S):
                    sv, B = qr_psi2sigmaB(st[s][p])
                    st[s][p] = np.ascontiguousarray(B)
                    st[s][p - 1] = np.tensordot(st[s][p - 1], sv, axes=(2, 0))
            for k, (i, j, w, _) in enumerate(chains):
                right[k][p - 1] = env_update_right(right[k][p], ket[j][p], w[p], bra=bra[i][p])
        ov = sum(overlap(bra[s], prev[s]) for s in range(S))  # _ints_wf_ovlp_mpssm, wavefunction.py:226-257
        if abs(1 - abs(ov)) < conv_tol:
            break
    return norm, bra, it


def site_rdm(cores: list[np.ndarray], site: int) -> np.ndarray:
    """One-site reduced density matrix rho[j,j'] of a site-0-centred MPS
    (what ``get_reduced_densities`` returns for key (site, site),
    _mps_cls.py:1208-1436): move the centre to ``site`` by QR, trace bonds."""
    cs = [np.array(c) for c in cores]
    for p in range(site):
        A, s = qr_psi2Asigma(cs[p])
        cs[p] = A
        cs[p + 1] = np.tensordot(s, cs[p + 1], axes=(1, 0))
    c = cs[site]
    return np.einsum("ajs,aks->jk", c, np.conj(c))


def reduced_density(cores: list[np.ndarray], remain_nleg) -> np.ndarray:
    """``_get_pure_reduced_density`` (_mps_cls.py:1208-1283) for a site-0-centred
    MPS: per site keep 2 legs (ket, bra), 1 leg (diagonal) or none; sites right
    of the last kept one are right-canonical and drop out.  Axes: kept sites in
    ascending order, (ket, bra) per 2-leg site."""
    legs = list(remain_nleg)
    last = max(i for i, n in enumerate(legs) if n)
    dens = None
    for p in range(last, -1, -1):
        c = cores[p]
        n = legs[p]
        if dens is None:
            sub = {2: "ijk,alk->iajl", 1: "ijk,ajk->iaj"}[n]
            dens = np.einsum(sub, c, np.conj(c))
        else:
            sub = {2: "lmi,bna,ia...->lbmn...", 1: "lmi,bma,ia...->lbm...", 0: "lmi,bma,ia...->lb..."}[n]
            dens = np.einsum(sub, c, np.conj(c), dens)
    return dens[0, 0, ...]


def _liouville_4d(cores, subspace=None):
    """``reshape_mat`` (``define_reshape_mat``, _mps_mpo.py:135-194): (i, j, k) -> (i, n, n, k); a site with a subspace
    projection holds only the entries ``P_inds`` of its n*n leg and is embedded back with zeros elsewhere.
    ``subspace`` = {site: (n, P_inds)}."""
    out = []
    for q, c in enumerate(cores):
        i, j, k = c.shape
        if subspace and q in subspace:
            n, inds = subspace[q]
            full = np.zeros((i, n * n, k), dtype=np.complex128)
            full[:, list(inds), :] = c
            out.append(full.reshape(i, n, n, k))
        else:
            n = math.isqrt(j)
            out.append(c.reshape(i, n, n, k))
    return out


def liouville_expectation(cores: list[np.ndarray], op: list[np.ndarray], subspace=None) -> complex:
    """Tr(O rho) for a vectorised density matrix (site dims n^2, index j = row*n + col)
    and a full-chain MPO ``op`` with n-dimensional physical legs
    (``_exp_liouville``, _mps_cls.py:3769-3838: "ab,bcde,adcf->fe").  With ``subspace`` the projected sites are
    embedded first (the reference's ``_exp_liouville`` takes isqrt of the projected leg and cannot run there;
    this is Tr(O rho) of the density its own partial traces describe)."""
    left = np.ones((1, 1), dtype=np.complex128)
    for c4, w in zip(_liouville_4d(cores, subspace), op):
        left = np.einsum("ab,bcde,adcf->fe", left, c4, w)
    return complex(left[0, 0])


def liouville_partial_trace(cores: list[np.ndarray], remain_nleg, subspace=None) -> np.ndarray:
    """``get_partial_trace`` (_mps_cls.py:1438-1510): trace out sites with 0 legs, keep
    the diagonal (1 leg) or both legs (2) of the others; the right-most kept site always
    keeps both legs.  ``subspace``: see ``_liouville_4d``."""
    legs = list(remain_nleg)
    center = max(i for i, n in enumerate(legs) if n)
    resh = _liouville_4d(cores, subspace)
    left = np.array([1.0 + 0.0j])
    for q in range(center):
        d = resh[q]
        t = {0: lambda x: np.einsum("ijjl->il", x), 1: lambda x: np.einsum("ijjl->ijl", x), 2: lambda x: x}[legs[q]](d)
        left = np.tensordot(left, t, axes=(-1, 0))
    right = np.array([1.0 + 0.0j])
    for q in range(len(cores) - 1, center, -1):
        right = np.einsum("ijjl->il", resh[q]) @ right
    return np.tensordot(left, np.tensordot(resh[center], right, axes=(-1, 0)), axes=(-1, 0))


def project_subspace_mpo(mpo: list[np.ndarray], inds: dict) -> list[np.ndarray]:
    """``TensorHamiltonian.project_subspace`` (hamiltonian_cls.py:852-880): bra and ket legs of the named sites
    restricted to the kept indices (4-leg cores ``[:, ix_(P, P), :]``, diagonal 3-leg cores ``[:, P, :]``)."""
    out = []
    for q, w in enumerate(mpo):
        if q in inds:
            P = list(inds[q])
            w = w[:, P, :] if w.ndim == 3 else w[:, np.ix_(P, P)[0], np.ix_(P, P)[1], :]
        out.append(np.ascontiguousarray(w))
    return out


def project_subspace_cores(cores: list[np.ndarray], inds: dict, m_aux_max: int) -> list[np.ndarray]:
    """``MPSCoefMPO.project_subspace`` (_mps_mpo.py:196-220): the physical leg of the (already canonicalised) cores
    sliced to the kept indices, then every bond trimmed to the caps of the projected lattice.  No re-orthogonalisation
    follows in the reference: the sliced "B" tensors are used as they are."""
    out = [c[:, list(inds[q]), :] if q in inds else c for q, c in enumerate(cores)]
    caps = bond_dims([c.shape[1] for c in out], m_aux_max)
    return [np.ascontiguousarray(c[:dl, :, :dr]) for c, (dl, dr) in zip(out, caps)]


def hermitise(cores: list[np.ndarray]) -> list[np.ndarray]:
    """``MPSCoef.hermitise`` (_mps_cls.py:2289-2312) / ``svd_conj_mpdo`` (:2516-2562): rho <- (rho + rho^dagger) / 2.

    Every core becomes the direct sum of itself and its dagger (conjugate, physical legs swapped; first site: both
    side by side with the factor 1/2, last site: stacked), then left to right the two-site matrix is SVD-truncated
    to the OLD bond dimension (rho_L = U[:, :chi], rho_R = (S Vh)[:chi]) and the chain is canonicalised to site 0
    without renormalisation (``canonicalize(..., orthogonal_center=0, incremental=False)``, :3519-3524)."""
    L = len(cores)
    four = [c.reshape(c.shape[0], math.isqrt(c.shape[1]), math.isqrt(c.shape[1]), c.shape[2]) for c in cores]
    dag = [np.conj(c).transpose(0, 2, 1, 3) for c in four]
    if L == 1:
        return [(0.5 * (four[0] + dag[0])).reshape(cores[0].shape)]
    chi = [c.shape[2] for c in cores[:-1]]
    dbl = []
    for q, (c, cd) in enumerate(zip(four, dag)):
        a, b, _, d = c.shape
        if q == 0:
            t = 0.5 * np.concatenate((c, cd), axis=3)
        elif q == L - 1:
            t = np.concatenate((c, cd), axis=0)
        else:
            t = np.zeros((2 * a, b, b, 2 * d), dtype=np.complex128)
            t[:a, ..., :d] = c
            t[a:, ..., d:] = cd
        dbl.append(t.reshape(t.shape[0], b * b, t.shape[3]))
    for q in range(L - 1):
        B = np.tensordot(dbl[q], dbl[q + 1], axes=(2, 0))
        i, j, l, m = B.shape
        U, S, Vh = scipy.linalg.svd(B.reshape(i * j, l * m), full_matrices=False)
        k = min(chi[q], len(S))
        dbl[q] = U[:, :k].reshape(i, j, k)
        dbl[q + 1] = (S[:k, None] * Vh[:k]).reshape(k, l, m)
    return canonicalize_site0(dbl, scale=None)


# --------------------------------------------------------------------------
# synthetic inputs -- used by tests and bench
# --------------------------------------------------------------------------
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


def synthetic_mps(dims: list[int], D: int, seed: int = 1):
    """Full-rank random MPS, canonicalised to site 0 and normalised."""
    rng = np.random.default_rng(seed)
    bd = bond_dims(dims, D)
    cores = []
    for (dl, dr), d in zip(bd, dims):
        c = rng.standard_normal((dl, d, dr)) + 1j * rng.standard_normal((dl, d, dr))
        cores.append(c / math.sqrt(2.0 * dl * d))  # O(1) singular values: no overflow on long chains
    return canonicalize_site0(cores)


def flops_heff(Dl, d, Dr, Ml, Mr) -> float:
    """F_H of SURVEY 8(d): complex MAC = 8 flop."""
    return 8.0 * (Dl * Dl * Ml * d * Dr + Dl * Dr * Ml * Mr * d * d + Dl * Dr * Dr * Mr * d)
