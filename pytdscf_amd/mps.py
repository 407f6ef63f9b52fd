"""Host-side MPS helpers mirroring ``LatticeInfo`` / ``SiteCoef.init_random``
(reference _mps_cls.py:2616-2703, _site_cls.py:410-476)."""

from __future__ import annotations

import math

import numpy as np


def bond_dims(dims, m_aux_max):
    """``LatticeInfo.get_bond_dim`` (_mps_cls.py:2616-2631)."""
    n = len(dims)
    out = []
    for i in range(n):
        left = 1 if i == 0 else min(m_aux_max, math.prod(dims[:i]))
        right = 1 if i == n - 1 else min(m_aux_max, math.prod(dims[i + 1 :]))
        out.append((min(left, dims[i] * right, m_aux_max), min(left * dims[i], right, m_aux_max)))
    return out


def product_state_cores(weights, bond_dim, space="hilbert"):
    """Hartree-product initial cores, zero padded to the capped bond dimensions
    (``SiteCoef.init_random``, _site_cls.py:438-463).  ``weights[i]`` is either
    a 1-D weight vector or an explicit 3-D core."""
    dims = [np.asarray(w).shape[-2] if np.asarray(w).ndim == 3 else len(w) for w in weights]
    out = []
    for (dl, dr), d, w in zip(bond_dims(dims, bond_dim), dims, weights):
        data = np.zeros((dl, d, dr), dtype=np.complex128)
        a = np.asarray(w, dtype=np.complex128)
        if a.ndim == 1:
            if space == "hilbert":
                data[0, :, 0] = a / np.linalg.norm(a)
            else:
                s = math.isqrt(d)
                data[0, :, 0] = a / np.trace(a.reshape(s, s))
        elif a.ndim == 3:
            i, j, k = a.shape
            data[:i, :j, :k] = a
        else:
            raise ValueError("initial core must be 1-D weights or a 3-D core")
        out.append(data)
    return out
