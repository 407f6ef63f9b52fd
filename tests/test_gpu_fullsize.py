"""GPU: BASELINE.json's full sizes through size-independent properties
(the oracle cannot finish these in seconds)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize(
    "name,shape",
    [("C4", (1024, 16, 1024, 32, 32)), ("C5", (512, 4, 512, 16, 16)), ("C3", (128, 32, 128, 16, 16)), ("C2", (32, 10, 32, 6, 6))],
)
def test_heff_apply_full_shape_properties(name, shape):
    """At the interior-site shape of each BASELINE config: the 3M and 4M complex
    products (different arithmetic, different tiles) agree, and the apply is linear."""
    from pytdscf_amd import engine as E

    r = E.heff_selfcheck(*shape)
    print(name, r)
    assert r["rel_3m_vs_4m"] < 1e-13
    assert r["linearity_defect"] < 1e-13
    assert np.isfinite(r["norm_Hx"]) and r["norm_Hx"] > 0


def test_c3_full_size_sweep_properties():
    """C3 at full size (L=6, d=32, D=128, M=16): norm to 1e-12, energy conserved,
    propagation reversible, real energy for the Hermitian MPO."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    L, d, D, M = 6, 32, 128, 16
    eng = TDVPEngine(L)
    eng.set_mpo(orc.synthetic_mpo(L, d, M, seed=0))
    eng.init_random([d] * L, D, seed=1)
    e0 = eng.expectation()
    a0 = eng.autocorr()
    assert abs(e0.imag) < 1e-12 * max(1.0, abs(e0))
    for _ in range(2):
        eng.propagate(1.0)
    assert abs(eng.norm() - 1) < 1e-12
    assert abs(eng.expectation() - e0) < 1e-7 * abs(e0)
    for _ in range(2):
        eng.propagate(-1.0)
    assert abs(eng.autocorr() - a0) < 1e-6 * abs(a0)
    assert abs(eng.expectation() - e0) < 1e-7 * abs(e0)


def test_c5_liouvillian_trace_like_invariants():
    """C5-type generator at reduced length (L=16, d=4, D=256, M=16): with zero damping
    the vectorised von Neumann generator H (x) 1 - 1 (x) H^T is Hermitian, so the
    Arnoldi / conserve_norm=False path must keep the norm by itself."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    L, D = 16, 256
    eng = TDVPEngine(L, integrator="arnoldi", conserve_norm=False)
    eng.set_mpo(orc.synthetic_liouvillian_mpo(L, 16, seed=0, gamma=0.0))
    eng.init_random([4] * L, D, seed=3)
    for _ in range(2):
        eng.propagate(0.5)
    assert abs(eng.norm() - 1) < 1e-8  # unitary up to thresh_sil accumulation; nothing renormalises


def test_adaptive_growth_properties_large():
    """Adaptive bond dimension at a size the oracle cannot follow in seconds
    (L=12, d=8, M=8, ranks 32 -> 128): norm to 1e-12, ranks never shrink, never exceed
    Dmax nor what the neighbouring bonds allow, grow by at most dD per half-sweep, and the
    energy of the Hermitian chain drifts only at the level of the projection error (the
    zero-padded propagator H P is not Hermitian: the reference's scheme, not a defect here)."""
    import time

    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    L, d, M, D0, Dmax, dD = 12, 8, 8, 32, 128, 24
    eng = TDVPEngine(L)
    eng.set_mpo(orc.synthetic_mpo(L, d, M, seed=2))
    eng.init_random([d] * L, D0, seed=3)
    eng.set_adaptive(True, Dmax=Dmax, dD=dD, p_proj=1e-10)
    e0 = eng.expectation()
    prev = eng.bond_dims()
    hist = [prev]
    t0 = time.time()
    for _ in range(3):
        eng.propagate(0.5)
        bd = eng.bond_dims()
        assert all(b >= a for a, b in zip(prev, bd)) and max(bd) <= Dmax
        assert all(b - a <= 2 * dD for a, b in zip(prev, bd))  # one step = two half-sweeps
        prev = bd
        hist.append(bd)
    print("bond dims", hist, "s/step", (time.time() - t0) / 3)
    assert max(prev) > D0
    for i, b in enumerate(prev):  # an isometry cannot be wider than its row space
        l, n, r, _ = eng.get_site_shape(i)
        assert r <= l * n and l <= n * r
    assert abs(eng.norm() - 1) < 1e-12
    e1 = eng.expectation()
    print("energy", e0, e1)
    assert abs(e1.imag) < 1e-10 * abs(e1) and abs(e1 - e0) < 2e-3  # |H| = O(1) for this chain
    eng.close()


def test_multistate_conservation_at_moderate_size():
    """Two coupled electronic states, L=8 d=8 D=64: energy and norm are conserved, the populations
    move and sum to one (size-independent properties; no oracle at this size)."""
    from oracle import tdvp_oracle as orc  # synthetic input builders only
    from pytdscf_amd import MultiStateEngine

    L, d, D, M, S = 8, 8, 64, 8, 2
    rng = np.random.default_rng(3)
    crandn = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)  # noqa: E731
    raw = [[crandn(a, d, b) for a, b in orc.bond_dims([d] * L, D)] for _ in range(S)]
    mpo = [[orc.synthetic_mpo(L, d, M, seed=0), None], [None, orc.synthetic_mpo(L, d, M, seed=1)]]
    w = [0.05 * crandn(a, d, d, b) for a, b in zip([1] + [3] * (L - 1), [3] * (L - 1) + [1])]
    mpo[0][1] = w
    mpo[1][0] = [np.ascontiguousarray(np.conj(c.transpose(0, 2, 1, 3))) for c in w]
    eng = MultiStateEngine(L, S)
    eng.set_hamiltonian(mpo, [[0.0, 0.01 + 0.02j], [0.01 - 0.02j, 0.3]])
    eng.set_states(raw, weights=[0.9, 0.1])
    e0, p0 = eng.expectation(), eng.pop_states()
    assert abs(e0.imag) < 1e-12
    for _ in range(3):
        eng.propagate(0.3)
    e1, p1 = eng.expectation(), eng.pop_states()
    assert abs(e1 - e0) < 1e-8 * max(1.0, abs(e0))
    assert abs(eng.norm() - 1.0) < 1e-12 and abs(sum(p1) - 1.0) < 1e-12
    assert abs(p1[0] - p0[0]) > 1e-6
    assert max(eng.krylov_stats()) <= 20
    eng.close()


def test_heff_throughput_floor_at_c4_shape():
    """Regression guard for the dominant kernel: three H_eff applies at the C4 interior shape
    (D=1024, d=16, M=32; 1.1e13 flop each) run at 81 TFLOP/s algorithmic on an MI355X
    (profiles/r01_bench_C4.json); fail when a change drops it below 60."""
    from pytdscf_amd import engine as E

    dl, d, dr, m = 1024, 16, 1024, 32
    ms = E.bench_heff(dl, d, dr, m, m, reps=3, warmup=1)
    flops = 8.0 * (dl * dl * m * d * dr + dl * dr * m * m * d * d + dl * dr * dr * m * d)
    tflops = flops / (ms * 1e-3) / 1e12
    assert tflops > 60.0, f"H_eff apply at the C4 shape: {tflops:.1f} TFLOP/s ({ms:.1f} ms)"
