"""MPO two-site compression sweeps on the MI355X (the only other O(D^3) code of the
reference besides the TDVP sweep).

Mirrors ``/root/reference/pytdscf/_mpo_cls.py``:
``guess_bond_dimension`` (:290-311), ``merge_mpos_twodot`` (:601-704, the direct sum of several
full-dimensional grid MPOs built bond by bond with an SVD of every two-site block), ``sweep_qr`` (:790-808),
``sweep_lq`` (:811-830), ``sweep_compress_twodot`` (:745-787) and the driver ``_compress_block_by_block``
(:880-911).  Cores are the reference's diagonal grid-MPO cores ``(M_l, n, M_r)``, real or complex.

Every factorisation runs on the device through the C ABI: the one-sided Jacobi SVD (``mitdvp_svd``,
``csrc/svd.hip``) and the Householder QR (``mitdvp_gauge_trf``, ``csrc/qr.hip``); there is no host
LAPACK path.  U and Vh differ from LAPACK's by the sign / phase of singular-vector pairs, i.e. the cores
differ by a gauge on the bonds; singular values, truncation ranks and the represented operator agree.
"""

from __future__ import annotations

import numpy as np

from . import engine as E


def guess_bond_dimension(svals, rate: float = 0.999999999999, trace=None) -> int:
    """Smallest rank whose singular values hold the fraction ``rate`` of their plain sum (_mpo_cls.py:290-311)."""
    svals = np.asarray(svals, dtype=float)
    if trace is None:
        trace = float(np.sum(svals))
    elif np.sum(svals) / trace < rate:
        return int(svals.size)
    if rate == 1.0:
        return int(svals.size)
    if not 0.0 < rate < 1.0:
        raise ValueError("contribution rate must be 0.0 < `rate` < 1.0")
    cumsum, rank = 0.0, 0
    while cumsum / trace < rate and rank < svals.size:
        cumsum += svals[rank]
        rank += 1
    return rank


def _svd(mat):
    """Economic SVD on the device; singular values descending; real input gives real factors."""
    real = not np.iscomplexobj(mat)
    U, s, Vh, _ = E.svd(np.ascontiguousarray(mat, dtype=np.complex128))
    order = np.argsort(-s, kind="stable")
    U, s, Vh = U[:, order], s[order], Vh[order, :]
    if real:  # rotations built from a real Gram matrix stay real; drop the zero imaginary parts
        U, Vh = np.ascontiguousarray(U.real), np.ascontiguousarray(Vh.real)
    return U, s, Vh


def _qr(mat):
    """Economic QR on the device.  Tall or square: Householder; wide (m < n): the left singular
    vectors serve as the orthonormal factor (same span, bond dimension m)."""
    m, n = mat.shape
    real = not np.iscomplexobj(mat)
    if m >= n:
        Q, R = E.gauge_trf(np.ascontiguousarray(mat, dtype=np.complex128).reshape(m, 1, n), "Psi2Asigma")
        Q = Q.reshape(m, n)
    else:
        U, s, Vh = _svd(mat)
        Q, R = U, s[:, None] * Vh
    if real:
        Q, R = np.ascontiguousarray(np.real(Q)), np.ascontiguousarray(np.real(R))
    return Q, R


def merge_mpos_twodot(mpos, k: int = 50, rate: float = 0.999999999999):
    """Direct sum of full-dimensional MPOs, compressed bond by bond (_mpo_cls.py:601-704).

    ``k`` (the sparse-SVD size of the reference's > 1e5 branch) is accepted for signature
    compatibility; the device SVD is always the full one."""
    nsite = len(mpos[0])
    merged = [np.concatenate([m[0] for m in mpos], axis=2)]
    if merged[0].shape[0] != 1:
        raise ValueError("the first core of every MPO must have left bond 1")
    for right in range(1, nsite):
        k_dim, m_dim = [0], [0]
        l_dim = mpos[0][right].shape[1]
        for m in mpos:
            k_dim.append(k_dim[-1] + m[right].shape[0])
            m_dim.append(m_dim[-1] + m[right].shape[2])
        core_left = merged[-1]
        mat_left = core_left.reshape(-1, core_left.shape[2])
        if right == nsite - 1:
            core_right = np.concatenate([m[-1] for m in mpos], axis=0)
            if core_right.shape[-1] != 1:
                raise ValueError("the last core of every MPO must have right bond 1")
            mat_right = core_right[:, :, 0]
        else:  # block-diagonal placement of the terms' cores: rows k, columns (grid point, m)
            mat_right = np.zeros((k_dim[-1], l_dim * m_dim[-1]), dtype=np.result_type(*[m[right] for m in mpos]))
            view = mat_right.reshape(k_dim[-1], l_dim, m_dim[-1])
            for i, m in enumerate(mpos):
                view[k_dim[i] : k_dim[i + 1], :, m_dim[i] : m_dim[i + 1]] = m[right]
        mat = mat_left @ mat_right
        nrm = np.linalg.norm(mat)
        mat = mat / nrm
        U, s, Vh = _svd(mat)
        bd = guess_bond_dimension(s, rate=rate)
        core_left = U[:, :bd].reshape(core_left.shape[0], core_left.shape[1], bd)
        sv = s[:bd, None] * Vh[:bd, :]
        core_right = sv.reshape(bd, l_dim, 1 if right == nsite - 1 else m_dim[-1])
        merged[-1] = core_left * np.sqrt(nrm)
        merged.append(core_right * np.sqrt(nrm))
    return merged


def sweep_qr(mpo):
    """Left-orthogonalise sites 0 .. nsite-2 (_mpo_cls.py:790-808)."""
    nsite = len(mpo)
    for p in range(nsite - 1):
        core = mpo[p]
        nrm = np.linalg.norm(core)
        core = core / nrm
        Q, R = _qr(core.reshape(core.shape[0] * core.shape[1], core.shape[2]))
        mpo[p] = Q.reshape(core.shape[0], core.shape[1], -1) * np.sqrt(nrm)
        mpo[p + 1] = np.einsum("ij,jkl->ikl", R, mpo[p + 1]) * np.sqrt(nrm)
    return mpo


def sweep_lq(mpo):
    """Right-orthogonalise sites nsite-1 .. 1 (_mpo_cls.py:811-830)."""
    nsite = len(mpo)
    for p in range(nsite - 1, 0, -1):
        core = mpo[p]
        nrm = np.linalg.norm(core)
        core = core / nrm
        Lt, Qt = _qr(np.ascontiguousarray(core.reshape(core.shape[0], core.shape[1] * core.shape[2]).T))
        mpo[p] = Lt.T.reshape(-1, core.shape[1], core.shape[2]) * np.sqrt(nrm)
        mpo[p - 1] = np.einsum("ijk,kl->ijl", mpo[p - 1], Qt.T) * np.sqrt(nrm)
    return mpo


def sweep_compress_twodot(mpo, rate: float = 0.999999999, left_to_right: bool = True):
    """SVD-truncate every bond through its two-site block (_mpo_cls.py:745-787)."""
    nsite = len(mpo)
    sites = range(0, nsite - 1) if left_to_right else range(nsite - 2, -1, -1)
    for left in sites:
        right = left + 1
        cl, cr = mpo[left], mpo[right]
        mat = np.einsum("ijk,klm->ijlm", cl, cr).reshape(cl.shape[0] * cl.shape[1], cr.shape[1] * cr.shape[2])
        nrm = np.linalg.norm(mat)
        mat = mat / nrm
        U, s, Vh = _svd(mat)
        bd = guess_bond_dimension(s, rate)
        if left_to_right:
            new_l = U[:, :bd]
            new_r = s[:bd, None] * Vh[:bd, :]
        else:
            new_l = U[:, :bd] * s[None, :bd]
            new_r = Vh[:bd, :]
        mpo[left] = new_l.reshape(cl.shape[0], cl.shape[1], bd) * np.sqrt(nrm)
        mpo[right] = new_r.reshape(bd, cr.shape[1], cr.shape[2]) * np.sqrt(nrm)
    return mpo


def compress_block_by_block(mpos, rate: float = 0.999999999, nsweep: int = 1, k: int = 1000, sub_mpo: int = 50):
    """``_compress_block_by_block`` (_mpo_cls.py:880-911): merge the terms in groups of ``sub_mpo``,
    canonicalise + compress each group, merge the groups, then ``nsweep`` canonicalise + compress passes."""
    groups = []
    for lo in range(0, len(mpos), sub_mpo):
        g = merge_mpos_twodot(mpos[lo : lo + sub_mpo], k=k, rate=rate)
        g = sweep_qr(g)
        g = sweep_compress_twodot(g, rate=rate, left_to_right=False)
        groups.append(g)
    mpo = merge_mpos_twodot(groups, k=k, rate=rate)
    for _ in range(nsweep):
        mpo = sweep_qr(mpo)
        mpo = sweep_compress_twodot(mpo, rate=rate, left_to_right=False)
    return mpo
