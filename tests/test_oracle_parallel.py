"""CPU: the oracle of the site-range sharded sweep (oracle/tdvp_parallel_oracle.py).  The reference's
MPI implementation cannot run here (no mpi4py), so this oracle is pinned by what can be checked:
one rank IS the serial oracle; without terms across the junctions the result is the serial one to
rounding; with them the deviation from the serial sweep falls as dt^2; the norm stays at 1 to the same
order; the split rule matches parallel_split_indices' contiguous ranges."""

import numpy as np
import pytest

from oracle import tdvp_oracle as orc
from oracle import tdvp_parallel_oracle as par


def _chain(L=8, d=3, M=4, D=8):
    return orc.synthetic_mpo(L, d, M, seed=0), orc.synthetic_mps([d] * L, D, seed=1)


def _serial(mps, mpo, dt, nstep):
    s = orc.OracleMPS([c.copy() for c in mps], mpo)
    s.build_right_envs()
    for _ in range(nstep):
        s.propagate(dt)
    return s


def test_split_rule():
    assert par.split_sites(64, 8) == [(8 * i, 8 * i + 8) for i in range(8)]
    assert par.split_sites(10, 3) == [(0, 4), (4, 7), (7, 10)]
    with pytest.raises(ValueError):
        mpo, mps = _chain(L=4)
        par.ParallelOracle(mps, mpo, 3)


def test_initial_state_is_reproduced():
    mpo, mps = _chain()
    for n in (1, 2, 3, 4):
        p = par.ParallelOracle([c.copy() for c in mps], mpo, n)
        g = p.gather()
        assert abs(abs(orc.overlap(g, mps)) - 1) < 1e-12 and abs(p.norm() - 1) < 1e-12


def test_one_rank_is_the_serial_sweep():
    mpo, mps = _chain()
    s = _serial(mps, mpo, 0.3, 3)
    p = par.ParallelOracle([c.copy() for c in mps], mpo, 1)
    for _ in range(3):
        p.step(0.3)
    assert abs(abs(orc.overlap(s.cores, p.gather())) - 1) < 1e-13


@pytest.mark.parametrize("nrank", [2, 3, 4])
def test_deviation_from_serial_falls_as_dt_squared(nrank):
    mpo, mps = _chain()
    errs = []
    for dt in (0.2, 0.1):
        nstep = int(round(0.4 / dt))
        s = _serial(mps, mpo, dt, nstep)
        p = par.ParallelOracle([c.copy() for c in mps], mpo, nrank)
        for _ in range(nstep):
            p.step(dt)
        g = p.gather()
        errs.append(1 - abs(orc.overlap(s.cores, g)) / np.sqrt(abs(orc.overlap(g, g))))
        assert abs(p.norm() - 1) < 1e-3
    assert errs[0] < 1e-5 and errs[1] < 0.4 * errs[0]  # infidelity ~ dt^2 at fixed total time


def test_no_coupling_across_the_junction_is_exact():
    """Blocks that do not talk to each other (MPO bond 1 at the junction, product state across it): the
    frozen boundary blocks are exact, so the sharded sweep must equal the serial one to rounding."""
    L, d, D = 8, 3, 4
    rng = np.random.default_rng(3)
    left = orc.synthetic_mpo(4, d, 3, seed=5)
    right = orc.synthetic_mpo(4, d, 3, seed=6)
    # H = H_left (x) 1 + 1 (x) H_right as one chain: direct sum with bond 2 at the junction ([H_l, 1] x [1; H_r])
    eye = np.eye(d).reshape(1, d, d, 1)
    def ident(n):
        return [eye.copy() for _ in range(n)]
    def dsum(a, b, first, last):
        out = []
        for p, (x, y) in enumerate(zip(a, b)):
            ml, mr = x.shape[0] + y.shape[0], x.shape[3] + y.shape[3]
            w = np.zeros((1 if first and p == 0 else ml, d, d, 1 if last and p == len(a) - 1 else mr), complex)
            if first and p == 0:
                w[0, :, :, : x.shape[3]] = x[0]
                w[0, :, :, x.shape[3] :] = y[0]
            elif last and p == len(a) - 1:
                w[: x.shape[0], :, :, 0] = x[..., 0]
                w[x.shape[0] :, :, :, 0] = y[..., 0]
            else:
                w[: x.shape[0], :, :, : x.shape[3]] = x
                w[x.shape[0] :, :, :, x.shape[3] :] = y
            out.append(w)
        return out
    mpo = dsum(left + ident(4), ident(4) + right, True, True)
    a = orc.synthetic_mps([d] * 4, D, seed=7)
    b = orc.synthetic_mps([d] * 4, D, seed=8)
    a[-1] = a[-1][:, :, :1] if a[-1].shape[2] > 1 else a[-1]
    mps = orc.canonicalize_site0([c.copy() for c in a] + [c.copy() for c in b])
    s = _serial(mps, mpo, 0.3, 2)
    p = par.ParallelOracle([c.copy() for c in mps], mpo, 2)
    for _ in range(2):
        p.step(0.3)
    g = p.gather()
    assert abs(abs(orc.overlap(s.cores, g)) / np.sqrt(abs(orc.overlap(g, g))) - 1) < 1e-10
    assert abs(p.norm() - 1) < 1e-10
