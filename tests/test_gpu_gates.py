"""GPU parity of one-site gates (Model(one_gate_to_apply=...), apply_one_gate) through
the C ABI against the reference's golden runs and the pinned oracle."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_gate_chain_golden(golden):
    from pytdscf_amd import TDVPEngine

    g = golden("gate_chain.npz")
    n = int(g["nsite"])
    mpo = [g[f"mpo{i}"] for i in range(n)]
    init = [g[f"init{i}"] for i in range(n)]
    dt = float(g["dt_au"])
    for ns in (1, 3):
        eng = TDVPEngine(n)
        eng.set_mpo(mpo)
        eng.set_mps(init, canonicalize=True)
        eng.set_gates({1: g["U1"], 4: g["U4"]})  # a full gate and a diagonal one
        e_last = None
        for _ in range(ns):
            e_last = eng.expectation()
            eng.propagate(dt)
        assert eng.krylov_stats() == list(g[f"n{ns}_krylov"])
        el = float(g[f"n{ns}_energy_last"])
        assert abs(e_last.real - el) < 1e-8 * abs(el)
        assert abs(eng.norm() - float(g[f"n{ns}_norm"])) < 1e-12
        ac = complex(g[f"n{ns}_autocorr"])
        assert abs(eng.autocorr() - ac) < 1e-8 * abs(ac)
        ef = float(g[f"n{ns}_energy_final"].real)
        assert abs(eng.expectation().real - ef) < 1e-8 * abs(ef)
        for i, c in enumerate(eng.get_mps()):
            np.testing.assert_allclose(c, g[f"n{ns}_final{i}"], atol=1e-9)
        eng.close()


def test_liouville_supergate_golden(golden):
    """Vectorised density matrix + one-site super-gate exp(D dt) (tests/test_mixedstate.py:373-413)."""
    from pytdscf_amd import TDVPEngine
    from pytdscf_amd.mps import product_state_cores
    from pytdscf_amd.operators import merge_operator_terms

    g = golden("gate_liouville.npz")
    n = int(g["nsite"])
    mpo = [g[f"mpo{i}"] for i in range(n)]
    init = product_state_cores([g[f"rho{i}"] for i in range(n)], int(g["bond_dim"]), space="liouville")
    ops = {"sz2": merge_operator_terms([([g["sz"]], [2])], [2] * n), "sz1sx3": merge_operator_terms([([g["sz"], g["sx"]], [1, 3])], [2] * n)}
    dt = float(g["dt_au"])
    for ns in (1, 3):
        eng = TDVPEngine(n, integrator="arnoldi", conserve_norm=False)
        eng.set_mpo(mpo)
        for k, op in enumerate(ops.values(), start=1):
            eng.set_trace_op(op, k)
        eng.set_mps(init, canonicalize=True, scale=None)
        eng.set_gates({2: g["G2"]})
        for _ in range(ns):
            eng.propagate(dt)
        assert eng.krylov_stats() == list(g[f"n{ns}_krylov"])
        assert abs(eng.norm() - float(g[f"n{ns}_norm"])) < 1e-10 * float(g[f"n{ns}_norm"])
        for k, name in enumerate(ops, start=1):
            ref = float(g[f"n{ns}_{name}"])
            assert abs(eng.expect_trace(k).real - ref) < 1e-8 * abs(ref) + 1e-12
        np.testing.assert_allclose(eng.partial_trace((0, 0, 2)), g[f"n{ns}_pt2"], atol=1e-10)
        np.testing.assert_allclose(eng.partial_trace((0, 2, 0, 1)), g[f"n{ns}_pt13"], atol=1e-10)
        eng.close()


@pytest.mark.parametrize("center_last", [False, True])
def test_apply_gates_on_demand_vs_oracle(center_last):
    """apply_one_gate outside a step: centre 0 (WFunc.apply_one_gate) and centre L-1
    (after a forward half-sweep); gates left and right of the centre and on it."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    L, d, M, D = 6, 3, 4, 7
    rng = np.random.default_rng(8)
    mpo = orc.synthetic_mpo(L, d, M, seed=2)
    init = [rng.standard_normal((a, d, b)) + 1j * rng.standard_normal((a, d, b)) for a, b in orc.bond_dims([d] * L, D)]
    gates = {s: rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d)) for s in (0, 2, 5)}
    gates[3] = rng.standard_normal(d) + 1j * rng.standard_normal(d)  # diagonal form
    st = orc.OracleMPS(orc.canonicalize_site0(init), mpo)
    eng = TDVPEngine(L)
    eng.set_mpo(mpo)
    eng.set_mps(init, canonicalize=True)
    if center_last:
        st.build_right_envs()
        st.sweep(0.3, True)
        eng.sweep(0.3, True)
    c = L - 1 if center_last else 0
    orc.apply_one_gate(st.cores, c, gates)
    eng.set_gates(gates)
    eng.apply_gates()
    eng.set_gates(None)
    for i, (a, b) in enumerate(zip(eng.get_mps(), st.cores)):
        np.testing.assert_allclose(a, b, atol=1e-10)
        assert eng.get_site_shape(i)[3] == (0 if i == c else (1 if i < c else 2))
    if not center_last:  # environments were invalidated: the next step rebuilds them
        st2 = orc.OracleMPS([x.copy() for x in st.cores], mpo)
        st2.cores[0] = st2.cores[0] / np.linalg.norm(st2.cores[0])
        nrm = eng.norm()
        e_eng = eng.expectation() / nrm**2
        assert abs(e_eng - st2.expectation()) < 1e-9 * abs(st2.expectation())
    eng.close()


def test_gate_bad_arguments():
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import TDVPEngine

    eng = TDVPEngine(4)
    eng.set_mpo(orc.synthetic_mpo(4, 2, 3, seed=1))
    eng.set_mps(orc.synthetic_mps([2] * 4, 2), canonicalize=True)
    with pytest.raises(ValueError):
        eng.set_gates({9: np.eye(2)})
    with pytest.raises(ValueError):
        eng.set_gates({1: np.zeros((2, 3))})
    eng.set_gates({1: np.eye(3)})  # wrong dimension for the site: reported when applied
    with pytest.raises(ValueError):
        eng.apply_gates()
    eng.close()
