#!/usr/bin/env python3
"""Golden vectors for the MPO two-site compression sweeps, produced by running the
REFERENCE's own functions (``/root/reference/pytdscf/_mpo_cls.py``: ``merge_mpos_twodot`` :601-704,
``sweep_qr`` :790-808, ``sweep_compress_twodot`` :745-787, ``_compress_block_by_block`` :880-911,
``guess_bond_dimension`` :290-311).  Development container only; third-party stand-ins as in
``make_golden.py``.  The fixture holds inputs (random 3-leg grid-MPO terms) and the reference's outputs
(bond dimensions, the dense operator each result represents, singular-value spectra)."""

from __future__ import annotations

import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, REPO, make_third_party_stubs  # noqa: E402


def dense(cores):
    import numpy as np

    t = cores[0]
    for c in cores[1:]:
        t = np.tensordot(t, c, axes=([-1], [0]))
    return t.reshape(t.shape[1:-1])


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not present")
    tmp = tempfile.mkdtemp(prefix="golden_mpo_")
    stubs = os.path.join(tmp, "stubs")
    make_third_party_stubs(stubs)
    sys.path.insert(0, REF)
    sys.path.insert(0, stubs)
    sys.path.insert(0, REPO)
    os.chdir(tmp)
    import numpy as np
    import pytdscf  # noqa: F401
    from pytdscf import _mpo_cls as M

    rng = np.random.default_rng(20261004)
    nsite, ngrid, nterm = 5, 8, 7
    out = {}
    mpos = []
    x = np.linspace(-1.0, 1.0, ngrid)
    pool = [np.ones(ngrid), x, x**2 - 0.3, np.exp(-x * x)]
    for k in range(nterm):
        term = []
        if k < nterm - 1:  # product terms drawn from a small pool of site functions: the sum has a low TT rank
            for p in range(nsite):
                f = pool[int(rng.integers(0, 3 if p % 2 else 4))] * (1.0 + 0.3 * rng.standard_normal())
                term.append(f.reshape(1, ngrid, 1).copy())
        else:  # one small rank-2 random term: kept at the tight rate, cut at the loose one
            bonds = [1, 2, 2, 2, 2, 1]
            for p in range(nsite):
                term.append((1e-7 if p == 0 else 1.0) * rng.standard_normal((bonds[p], ngrid, bonds[p + 1])))
        mpos.append(term)
        for p, c in enumerate(term):
            out[f"in_{k}_{p}"] = c
    out["nterm"], out["nsite"] = np.array(nterm), np.array(nsite)
    # the operators are compared on a fixed random sample of grid points (the dense tensor has 8^5 entries)
    idx = rng.integers(0, ngrid, size=(600, nsite))
    out["sample_idx"] = idx
    pick = lambda t: t[tuple(idx.T)]  # noqa: E731
    total = sum(dense(t) for t in mpos)
    out["dense_sum"] = pick(total)

    for tag, rate in (("tight", 0.999999999999), ("loose", 0.99999)):
        merged = M.merge_mpos_twodot([[c.copy() for c in t] for t in mpos], rate=rate)
        out[f"merge_{tag}_bonds"] = np.array([c.shape[2] for c in merged[:-1]])
        out[f"merge_{tag}_dense"] = pick(dense(merged))
        canon = M.sweep_qr([c.copy() for c in merged])
        out[f"qr_{tag}_dense"] = pick(dense(canon))
        comp = M.sweep_compress_twodot([c.copy() for c in canon], rate=rate, left_to_right=False)
        out[f"comp_{tag}_bonds"] = np.array([c.shape[2] for c in comp[:-1]])
        out[f"comp_{tag}_dense"] = pick(dense(comp))
        comp2 = M.sweep_compress_twodot([c.copy() for c in comp], rate=rate, left_to_right=True)
        out[f"comp2_{tag}_bonds"] = np.array([c.shape[2] for c in comp2[:-1]])
        out[f"comp2_{tag}_dense"] = pick(dense(comp2))
    blk = M._compress_block_by_block([[c.copy() for c in t] for t in mpos], 0.999999999, 1, 1000)
    out["block_bonds"] = np.array([c.shape[2] for c in blk[:-1]])
    out["block_dense"] = pick(dense(blk))
    out["gbd_svals"] = np.array([0.9, 0.3, 0.1, 0.01, 1e-4, 1e-9])
    out["gbd_ranks"] = np.array([M.guess_bond_dimension(out["gbd_svals"], r) for r in (0.5, 0.9, 0.99, 0.999999, 1.0)])
    np.savez_compressed(os.path.join(HERE, "mpo_compress.npz"), **out)
    print("wrote mpo_compress.npz:", {k: out[k].tolist() for k in out if k.endswith("bonds")})


if __name__ == "__main__":
    main()
