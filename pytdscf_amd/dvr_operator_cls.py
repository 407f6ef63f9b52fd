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
