"""GPU: the "edge" form of the H_eff apply (csrc/engine.hip::heff_apply_edge, csrc/zgemm.hip::reduce_epilogue) against
the oracle's plain three-leg contraction (oracle/tdvp_oracle.py::heff_apply; reference
multiplyH_MPS_direct_MPO._op_lcr_dot, _contraction.py:1038-1173, with the identity short-cuts of _mps_mpo.py:510-523).

The form applies when every non-zero (c, t) block of the MPO core sits in a row whose left block is the identity or in a
column whose right block is the identity (finite-state-machine MPOs between canonical environments, and direct sums of
them: the Liouville-space generator of C5 has three start and three end states).  It is two GEMMs whose 64 x 64 tiles are
contracted with a reduced core in the epilogue; the M-fold intermediates of the three-stage chain are never written.
mitdvp_heff_apply_center issues the apply exactly as a local exponential does and reports the form in bit 4 of its flags.

Tolerance: 1e-12 relative in the max norm over the whole output (complex128, sums of up to 512 x 128 terms).
"""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EDGE = 16


def _crandn(rng, *s):
    a = rng.standard_normal(s + (2,))
    return a.view(np.complex128).reshape(s)


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def _engine(L, edge="1", **kw):
    """an engine with MITDVP_EDGE_APPLY set while it is created (the variable is read there): "1" = the form wherever it
    is valid, "0" = never, None = the library's size rule"""
    from pytdscf_amd import TDVPEngine

    old = os.environ.get("MITDVP_EDGE_APPLY")
    if edge is None:
        os.environ.pop("MITDVP_EDGE_APPLY", None)
    else:
        os.environ["MITDVP_EDGE_APPLY"] = edge
    try:
        return TDVPEngine(L, **kw)
    finally:
        if old is None:
            os.environ.pop("MITDVP_EDGE_APPLY", None)
        else:
            os.environ["MITDVP_EDGE_APPLY"] = old


def _to_site(eng, c):
    eng.build_envs(1)
    for _ in range(c):  # centre to site c: left-canonical sites and their blocks behind it
        eng.split_center(True)
        eng.absorb_bond(True)


def _check_center(orc, eng, mpo, c, rng, want_edge=True, tol=1e-12):
    got, flags = eng.heff_apply_center()
    assert bool(flags & EDGE) == want_edge, flags
    if want_edge:
        assert flags & 7 == 0, flags
    Lb, Rb, psi = eng.get_env(0, c), eng.get_env(1, c + 1), eng.get_site(c)
    assert _rel(got, orc.heff_apply(Lb, mpo[c], Rb, psi)) < tol
    x = _crandn(rng, *psi.shape)  # nothing about the apply may depend on x being the state
    got, _ = eng.heff_apply_center(x)
    assert _rel(got, orc.heff_apply(Lb, mpo[c], Rb, x)) < tol


@pytest.mark.parametrize("mode", ["3m", "4m"])
def test_c3_shape_interior_and_tapering_sites(mode):
    """BASELINE configs[2]'s chain (L=6, d=32, D=128, M=16; bonds 1,32,128,128,128,32,1): the two interior sites
    (128 x 32 x 128: two row blocks of the epilogue) and the tapering ones (32 x 32 x 128, 128 x 32 x 32)."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import engine as E
    from pytdscf_amd import synthetic as syn

    L, d, D, M = 6, 32, 128, 16
    mpo = syn.synthetic_mpo(L, d, M, seed=0)
    rng = np.random.default_rng(7)
    E.set_gemm_mode(mode)
    try:
        for c in (1, 2, 3, 4):
            eng = _engine(L)
            eng.set_mpo(mpo)
            eng.init_random([d] * L, D, seed=1)
            _to_site(eng, c)
            _check_center(orc, eng, mpo, c, rng)
            eng.close()
    finally:
        E.set_gemm_mode("3m")


def test_c5_shape_direct_sum_generator():
    """The Liouville-space generator of configs[4] (d = 4, M = 16: a direct sum of three finite-state machines, so three
    identity states on either bond and 64 (u, v) pairs per tile = four column blocks of the epilogue), D = 512 reached at
    reduced length."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import synthetic as syn

    L, D = 14, 512
    mpo = syn.synthetic_liouvillian_mpo(L, 16, seed=0, gamma=0.002)
    eng = _engine(L, edge=None, integrator="arnoldi", conserve_norm=False)  # the size rule selects the form here
    eng.set_mpo(mpo)
    eng.init_random([4] * L, D, seed=3)
    c = L // 2
    assert eng.get_site_shape(c)[:3] == (D, 4, D)
    _to_site(eng, c)
    _check_center(orc, eng, mpo, c, np.random.default_rng(8))
    eng.close()


def test_ragged_groups_and_a_general_core():
    """d = 12, M = 10: neither divides 64 (tiles hold 5 x 6 whole groups, 60 of their 64 rows / columns are used, bonds that
    are no multiple of the tile); then the same chain under a core with a block between two general states: the edge
    form must not be chosen, the result stays right."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import synthetic as syn

    L, d, D, M = 6, 12, 50, 10
    rng = np.random.default_rng(9)
    mpo = syn.synthetic_mpo(L, d, M, seed=2)
    eng = _engine(L)
    eng.set_mpo(mpo)
    eng.init_random([d] * L, D, seed=4)
    assert eng.get_site_shape(2)[:3] == (D, d, D)
    _to_site(eng, 2)
    _check_center(orc, eng, mpo, 2, rng)
    eng.close()
    general = [w.copy() for w in mpo]
    general[2][3, :, :, 4] = 0.01 * _crandn(rng, d, d)  # neither state 3 (left) nor state 4 (right) is an identity state
    eng = _engine(L)
    eng.set_mpo(general)
    eng.init_random([d] * L, D, seed=4)
    _to_site(eng, 2)
    _check_center(orc, eng, general, 2, rng, want_edge=False)
    eng.close()


def test_sweeps_with_and_without_the_edge_form_agree():
    """A whole time step of the C3 chain with the form on (MITDVP_EDGE_APPLY=1) and off (=0; read when the engine is
    created): same Krylov counts, energies and autocorrelation to 1e-10, states to fidelity 1e-10."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import synthetic as syn

    L, d, D, M, dt = 6, 32, 128, 16, 1.0
    mpo = syn.synthetic_mpo(L, d, M, seed=0)
    res = {}
    for on in ("1", "0"):
        eng = _engine(L, edge=on)
        eng.set_mpo(mpo)
        eng.init_random([d] * L, D, seed=1)
        eng.propagate(dt)
        res[on] = (eng.expectation(), eng.autocorr(), eng.krylov_stats(), eng.get_mps(), eng.norm())
        eng.close()
    e1, a1, k1, s1, n1 = res["1"]
    e0, a0, k0, s0, n0 = res["0"]
    assert k1 == k0
    assert abs(e1 - e0) < 1e-10 * abs(e0) and abs(a1 - a0) < 1e-10 * abs(a0)
    assert abs(n1 - 1) < 1e-12 and abs(n0 - 1) < 1e-12
    assert abs(abs(orc.overlap(s0, s1)) - 1) < 1e-10


def test_c4_shape_forced():
    """The C4 interior shape (1024 x 16 x 1024, M = 32) with the form forced on (the size rule keeps the trimmed
    three-stage chain there): whole output against the oracle."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import synthetic as syn

    L, d, D, M = 7, 16, 1024, 32
    mpo = syn.synthetic_mpo(L, d, M, seed=0)
    eng = _engine(L)
    eng.set_mpo(mpo)
    eng.init_random([d] * L, D, seed=1)
    c = 3
    _to_site(eng, c)
    got, flags = eng.heff_apply_center()
    assert flags & EDGE, flags
    Lb, Rb, psi = eng.get_env(0, c), eng.get_env(1, c + 1), eng.get_site(c)
    ch = max(1, min(D, int(6.4e7 // (M * d * D))))
    assert _rel(got, orc.heff_apply_chunked(Lb, mpo[c], Rb, psi, ch)) < 1e-12
    eng.close()
