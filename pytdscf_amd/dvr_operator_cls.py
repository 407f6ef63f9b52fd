"""``pytdscf.dvr_operator_cls``: operators on DVR grids as tensor operators / MPOs.

Setup-time helpers beside the sweep: the reference's scripts build their Hamiltonian with
these before they reach the engine (dvr_operator_cls.py:1080-1252).  Own implementations;
``TensorOperator`` lives in ``api.py``."""

from __future__ import annotations

from itertools import product

import numpy as np

from .api import TensorOperator  # noqa: F401


def construct_fulldimensional(dvr_prims, func=None, db=None, ref_ene=0.0, dipole=False, efield=(1.0e-02, 1.0e-02, 1.0e-02)):
    """Full-dimensional diagonal operator ``func(q_1, .., q_f) - ref_ene`` on the product DVR
    grid, ``{(0, 1, .., f-1): TensorOperator}`` (dvr_operator_cls.py:1080-1132)."""
    if db is not None:
        raise NotImplementedError("electronic-structure databases (ase.db) are not read here; pass func=")
    if func is None:
        raise TypeError("construct_fulldimensional needs func=")
    grids = [np.asarray(p.get_grids()) for p in dvr_prims]
    op = TensorOperator(shape=tuple(len(g) for g in grids), only_diag=True)
    for idx in product(*[range(len(g)) for g in grids]):
        op.tensor_orig[idx] = func(*[g[i] for g, i in zip(grids, idx)]) - ref_ene
    return {tuple(range(len(dvr_prims))): op}


def construct_kinetic_mpo(dvr_prims, coefs=None):
    """MPO of ``sum_i -coef_i/2 d^2/dQ_i^2`` with bond dimension 2 (dvr_operator_cls.py:1199-1250):
    ``[T_1, 1] [[1, 0], [T_2, 1]] ... [[1], [T_f]]``."""
    n = len(dvr_prims)
    coefs = [1.0] * n if coefs is None else list(coefs)
    mpo = []
    for i, (p, c) in enumerate(zip(dvr_prims, coefs)):
        t = -0.5 * np.asarray(p.get_2nd_derivative_matrix_dvr()) * c
        g = t.shape[0]
        if n == 1:
            w = np.zeros((1, g, g, 1), dtype=np.complex128)
            w[0, :, :, 0] = t
        elif i == 0:
            w = np.zeros((1, g, g, 2), dtype=np.complex128)
            w[0, :, :, 0], w[0, :, :, 1] = t, np.eye(g)
        elif i == n - 1:
            w = np.zeros((2, g, g, 1), dtype=np.complex128)
            w[0, :, :, 0], w[1, :, :, 0] = np.eye(g), t
        else:
            w = np.zeros((2, g, g, 2), dtype=np.complex128)
            w[0, :, :, 0], w[1, :, :, 0], w[1, :, :, 1] = np.eye(g), t, np.eye(g)
        mpo.append(w)
    return mpo


def construct_kinetic_operator(dvr_prims, coefs=None, forms="mpo"):
    """Kinetic energy operator as one MPO key or as a sum of one-site terms
    (dvr_operator_cls.py:1135-1196)."""
    n = len(dvr_prims)
    coefs = [1.0] * n if coefs is None else list(coefs)
    if forms.lower() == "mpo":
        return {tuple((i, i) for i in range(n)): TensorOperator(mpo=construct_kinetic_mpo(dvr_prims, coefs))}
    if forms.lower() == "sop":
        return {((i, i),): TensorOperator(tensor=-0.5 * np.asarray(p.get_2nd_derivative_matrix_dvr()) * c, only_diag=False, legs=(i, i))
                for i, (p, c) in enumerate(zip(dvr_prims, coefs))}
    raise ValueError("forms must be 'sop' or 'mpo'")


def _tt_round_diag(cores, rate, cap=None, power=2):
    """Tensor-train rounding of a chain of diagonal (r, n, r') cores: right-to-left QR sweep, then
    left-to-right SVDs that keep the leading singular values whose squared sum (``power=2``) or
    plain sum (``power=1``, what the reference's final sweep_compress_twodot measures,
    _mpo_cls.py:764) reaches ``rate``."""
    cores = [np.array(c) for c in cores]
    for i in range(len(cores) - 1, 0, -1):
        r, n, rr = cores[i].shape
        q, t = np.linalg.qr(cores[i].reshape(r, n * rr).T)  # C_i = T^T Q^T
        cores[i] = q.T.reshape(-1, n, rr)
        cores[i - 1] = np.tensordot(cores[i - 1], t.T, axes=(2, 0))
    for i in range(len(cores) - 1):
        r, n, rr = cores[i].shape
        u, sv, vh = np.linalg.svd(cores[i].reshape(r * n, rr), full_matrices=False)
        if rate >= 1.0:  # lossless: only numerically zero directions go
            k = int(np.count_nonzero(sv > 1.0e-13 * sv[0])) if sv[0] > 0 else 1
        else:
            w = sv**power
            tot, cum, k = w.sum(), 0.0, 0
            while k < len(sv) and (tot == 0.0 or cum / tot < rate):
                cum += w[k]
                k += 1
        k = max(min(k, cap) if cap else k, 1)
        cores[i] = u[:, :k].reshape(r, n, k)
        cores[i + 1] = np.tensordot(sv[:k, None] * vh[:k], cores[i + 1], axes=(1, 0))
    return cores


def construct_nMR_recursive(dvr_prims, nMR=3, ndof=None, func=None, db=None, df=None, active_dofs=None, site_order=None,
                            zero_indices=None, return_tensor=False, include_const_in_mpo=False, ref_ene=None, dipole=False,
                            efield=(1.0, 1.0, 1.0), rate=1.0, k=200, nsweep=1):
    """n-mode-representation operator from explicit mode functions,
    ``func = {(i,): f_i, (i, j): f_ij, ...}`` -> full-chain diagonal MPO cores of
    ``sum_key f_key(q_key)`` (dvr_operator_cls.py:691-990, the ``func`` branch; the database /
    dataframe branches and the inclusion-exclusion separation they need are not read here).
    Every term is decomposed exactly, the terms are summed as a direct sum with identity
    fill-ins and the chain is rounded to the contribution ``rate`` (bond cap ``k``)."""
    from itertools import combinations

    from .operators import merge_operator_terms

    if func is None or db is not None or df is not None:
        raise NotImplementedError("construct_nMR_recursive: only the func={modes: callable} form is implemented")
    if site_order is not None or active_dofs is not None or zero_indices is not None or return_tensor:
        raise NotImplementedError("construct_nMR_recursive: site_order / active_dofs / zero_indices / return_tensor")
    n = len(dvr_prims)
    dims = [len(p.get_grids()) for p in dvr_prims]
    scalar = float(func[()]()) if () in func else 0.0
    terms = []
    for order in range(1, nMR + 1):
        for modes in combinations(range(n), order):
            if modes not in func:
                continue
            grids = [np.asarray(dvr_prims[p].get_grids()) for p in modes]
            op = TensorOperator(shape=tuple(len(g) for g in grids), only_diag=True, legs=modes)
            for idx in product(*[range(len(g)) for g in grids]):
                op.tensor_orig[idx] = func[modes](*[g[i] for g, i in zip(grids, idx)])
            # non-adjacent modes: decompose on the listed modes, the gaps get identity fill-ins in the merge
            op.decompose(decompose_type="QRD")
            terms.append((op.tensor_decomposed, list(modes)))
    if include_const_in_mpo and scalar != 0.0:
        terms.append(([np.full((1, dims[0], 1), scalar)], [0]))
    if not terms:
        raise ValueError("construct_nMR_recursive: func holds no mode function up to nMR")
    full = merge_operator_terms(terms, dims)  # 4-leg cores of a diagonal operator
    diag = [np.real_if_close(np.einsum("aiib->aib", w)) for w in full]
    if rate < 1.0 or (k and max(c.shape[2] for c in diag) > k):
        diag = _tt_round_diag(diag, min(rate, 1.0), cap=k, power=1)  # the criterion of the final sweep_compress_twodot
    return diag


def tensor_dict_to_mpo(tensor_dict, rate: float = 1.0, nsweep: int = 1):
    """n-mode-representation grid tensors ``{(i,): v_i[q_i], (i, j): v_ij[q_i, q_j], ...}`` -> one
    full-chain DIAGONAL MPO (3-leg cores (M_l, n, M_r)) whose value at a grid point is the sum of
    the tensors' values there (dvr_operator_cls.py:1012-1048; reference test
    tests/test_compress_mpo.py).  The scalar entry ``tensor_dict[()]`` is not part of the MPO.
    Every term is tensor-train decomposed exactly, the terms are summed as a direct sum with
    identity fill-ins and the chain is rounded to the contribution ``rate``."""
    from .api import TensorOperator
    from .operators import merge_operator_terms

    if not (0.0 < rate <= 1.0):
        raise ValueError("rate must be 0.0 < rate <= 1.0")
    dims = []
    while (len(dims),) in tensor_dict:
        dims.append(int(np.asarray(tensor_dict[(len(dims),)]).shape[0]))
    if not dims:
        raise ValueError("tensor_dict needs the one-mode terms (0,), (1,), ... to size the sites")
    terms = []
    for key, t in tensor_dict.items():
        if key == ():
            continue
        t = np.asarray(t)
        if t.shape != tuple(dims[i] for i in key):
            raise ValueError(f"tensor of key {key} has shape {t.shape}, the grids {tuple(dims[i] for i in key)}")
        if list(key) != sorted(set(key)):
            raise ValueError(f"key {key} must list distinct sites in ascending order")
        op = TensorOperator(tensor=t, legs=tuple(key), only_diag=True)
        op.decompose()
        terms.append((op.tensor_decomposed, op.sites))
    full = merge_operator_terms(terms, dims)
    diag = [np.ascontiguousarray(np.einsum("ajjb->ajb", w)) for w in full]
    for _ in range(max(1, int(nsweep))):
        diag = _tt_round_diag(diag, rate, power=1)
    if all(np.abs(w.imag).max() == 0.0 for w in diag):
        diag = [np.ascontiguousarray(w.real) for w in diag]
    return diag
