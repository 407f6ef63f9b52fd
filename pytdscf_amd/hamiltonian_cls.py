"""``pytdscf.hamiltonian_cls`` surface: ``TensorHamiltonian`` (MPO form) and the polynomial /
sum-of-products form (``PolynomialHamiltonian``, ``TermProductForm``, ``TermOneSiteForm``,
``read_potential_nMR``; hamiltonian_cls.py:25-330, :365-460, :882-1030).

The engine contracts MPOs only.  For the MPS *standard method* (no single-particle functions:
the site index IS the primitive basis) a sum of products of one-site matrices is an MPO of bond
dimension = number of terms, so a ``PolynomialHamiltonian`` is converted exactly
(``to_tensor_hamiltonian``: one-site matrices from the primitive basis, direct sum of the product
terms, lossless rounding) and runs on the same sweep.  The reference's MPS-MCTDH variants of these
operators (time-dependent single-particle functions) are a different method and stay out of scope."""

from __future__ import annotations

import itertools
import math

import numpy as np

from .api import TensorHamiltonian, TensorOperator  # noqa: F401
from .operators import compress_mpo, merge_operator_terms


class TermProductForm:
    """coef * prod_k op_keys[k] on site op_dofs[k], e.g. ``TermProductForm(1.0, [0, 1], ["q^1", "q^2"])``."""

    def __init__(self, coef, op_dofs, op_keys):
        if len(op_dofs) != len(op_keys):
            raise ValueError("op_dofs and op_keys must have the same length")
        self.coef, self.op_dofs, self.op_keys = coef, list(op_dofs), list(op_keys)
        self.mode_ops = dict(zip(self.op_dofs, self.op_keys))

    def __repr__(self):
        return f"{self.coef:+.4e} " + " * ".join(f"{k}_{d}" for d, k in zip(self.op_dofs, self.op_keys))


class TermOneSiteForm:
    """coef * op_key on site op_dof."""

    def __init__(self, coef, op_dof: int, op_key: str):
        self.coef, self.op_dof, self.op_key = coef, int(op_dof), op_key

    def __repr__(self):
        return f"{self.coef:+.4e} {self.op_key}_{self.op_dof}"


class PolynomialHamiltonian:
    """``PolynomialHamiltonian(ndof, nstate=1, name="hamiltonian", matJ=None)``: scalar terms
    ``coupleJ[i][j]``, one-site terms ``onesite[i][j]`` and product terms ``general[i][j]``."""

    def __init__(self, ndof: int, nstate: int = 1, name: str = "hamiltonian", matJ=None):
        self.ndof, self.nstate, self.name = int(ndof), int(nstate), name
        if matJ is None:
            self.coupleJ = [[complex(0.0) for _ in range(nstate)] for _ in range(nstate)]
        else:
            if len(matJ) != nstate or len(matJ[0]) != nstate:
                raise ValueError("matJ must be square matrix")
            self.coupleJ = [[complex(matJ[i][j]) for j in range(nstate)] for i in range(nstate)]
        self.onesite = [[[] for _ in range(nstate)] for _ in range(nstate)]
        self.general = [[[] for _ in range(nstate)] for _ in range(nstate)]

    def set_HO_potential(self, basinfo, *, enable_onesite=True) -> None:
        """-d^2/2 + omega^2 (q - q0)^2 / 2 on every mode of every state, expanded in q^2, q^1 and a
        constant around the basis centre q0 (hamiltonian_cls.py:400-434)."""
        for s in range(self.nstate):
            for idof in range(self.ndof):
                pb = basinfo.get_primbas(s, idof)
                q0, w = pb.origin_mwc, pb.freq_au
                terms = [TermOneSiteForm(-0.5, idof, "d^2"), TermOneSiteForm(w**2 / 2, idof, "q^2")]
                if q0 != 0.0:
                    terms.append(TermOneSiteForm(-(w**2) * q0, idof, "q^1"))
                    self.coupleJ[s][s] += w**2 / 2 * q0**2
                (self.onesite if enable_onesite else self.general)[s][s] += terms

    def set_HO_potential_ham1(self, basinfo) -> None:
        for s in range(self.nstate):
            self.onesite[s][s] += [TermOneSiteForm(1.0, idof, "ham1") for idof in range(self.ndof)]

    def set_LVC(self, bas_info, first_order_coupling) -> None:
        """Linear vibronic coupling: harmonic wells + ``{(i, j): {mode: coef}}`` * q_mode between states."""
        self.set_HO_potential(bas_info, enable_onesite=True)
        for (i, j), coupling in first_order_coupling.items():
            for idof, coef in coupling.items():
                self.onesite[i][j].append(TermOneSiteForm(coef, idof, "q^1"))

    # ---- conversion to the MPO form the engine contracts --------------------------------
    def to_tensor_hamiltonian(self, basinfo) -> TensorHamiltonian:
        nst = self.nstate
        prims = [[basinfo.get_primbas(s, i) for i in range(self.ndof)] for s in range(nst)]
        for s in range(1, nst):
            for a, b in zip(prims[0], prims[s]):
                if (a.nprim, a.freq_au, a.origin_mwc) != (b.nprim, b.freq_au, b.origin_mwc):
                    raise NotImplementedError("polynomial Hamiltonians with different primitive bases per electronic state "
                                              "(integrals between displaced bases)")
        dims = [p.nprim for p in prims[0]]
        pot = [[{} for _ in range(nst)] for _ in range(nst)]
        for i, j in itertools.product(range(nst), repeat=2):
            site_sum = {}  # one-site terms are summed per site first
            terms = []
            for t in list(self.onesite[i][j]) + list(self.general[i][j]):
                dofs, keys = ([t.op_dof], [t.op_key]) if isinstance(t, TermOneSiteForm) else (t.op_dofs, t.op_keys)
                if not dofs:
                    raise ValueError("a product term needs at least one operator; scalars go to coupleJ")
                if len(dofs) == 1:
                    m = complex(t.coef) * prims[j][dofs[0]].op_matrix(keys[0])
                    site_sum[dofs[0]] = site_sum.get(dofs[0], 0.0) + m
                    continue
                order = np.argsort(dofs)
                if len(set(dofs)) != len(dofs):
                    raise ValueError(f"repeated site in product term {t!r}")
                cores = []
                for n_, k in enumerate(order):
                    m = prims[j][dofs[k]].op_matrix(keys[k]).astype(np.complex128)
                    cores.append((complex(t.coef) * m if n_ == 0 else m)[None, :, :, None])
                terms.append((cores, [int(dofs[k]) for k in order]))
            for p, m in sorted(site_sum.items()):
                terms.append(([np.asarray(m, dtype=np.complex128)[None, :, :, None]], [p]))
            if terms:
                mpo = merge_operator_terms(terms, dims)
                mpo = compress_mpo(mpo) if len(mpo) > 1 else mpo
                key = tuple((p, p) for p in range(self.ndof))
                pot[i][j][key] = TensorOperator(mpo=mpo, legs=tuple(x for p in range(self.ndof) for x in (p, p)))
            if self.coupleJ[i][j] != 0.0:
                c = complex(self.coupleJ[i][j])
                pot[i][j][()] = c.real if c.imag == 0.0 else c
        return TensorHamiltonian(self.ndof, potential=pot, name=self.name)


def read_potential_nMR(potential_emu, *, active_modes=None, name="hamiltonian", cut_off=None, dipole_emu=None,
                       print_out=False, active_momentum=None, div_factorial=True, efield=(1.0, 1.0, 1.0)):
    """Polynomial (n-mode-representation, Taylor) potential from force constants
    ``{(1, 1): k_11, (1, 2, 2): k_122, ...}`` (1-based mode labels, derivatives NOT divided by the
    factorials) plus the kinetic energy -d^2/2 per mode (hamiltonian_cls.py:882-1030).  With
    ``dipole_emu`` the operator is mu . efield instead and carries no kinetic energy."""
    if active_modes is None:
        src = dipole_emu if dipole_emu is not None else potential_emu
        if src is None:
            raise ValueError("active_modes must be set")
        active_modes = sorted(set(itertools.chain.from_iterable(src.keys())))
    k_orig, scalar = potential_emu, 0.0
    if dipole_emu is not None:
        active_momentum, k_orig = False, {}
        for key, val in dipole_emu.items():
            if key == ():
                scalar += float(np.dot(val, efield))
            else:
                k_orig[key] = float(np.dot(val, efield))
    site_of = {mode: i for i, mode in enumerate(active_modes)}
    nmode = len(active_modes)
    powers = {}
    for key, value in k_orig.items():
        if key == ():
            scalar = value
            continue
        if not set(key) <= set(active_modes):
            continue
        deg = [0] * nmode
        for mode in key:
            deg[site_of[mode]] += 1
        if tuple(deg) in powers:
            raise ValueError("duplicated keys in k_orig")
        powers[tuple(deg)] = value
    ham = PolynomialHamiltonian(nmode, 1, name, [[scalar]])
    terms = []
    if active_momentum is None:
        terms += [TermProductForm(-0.5, [i], ["d^2"]) for i in range(nmode)]
    elif active_momentum:
        terms += [TermProductForm(coef, [site_of[mode]], ["d^2"]) for mode, coef in dict(active_momentum).items()]
    for deg, value in powers.items():
        dofs = [i for i, o in enumerate(deg) if o > 0]
        fac = 1.0
        if div_factorial:
            for i in dofs:
                fac /= math.factorial(deg[i])
        coef = fac * value
        if cut_off is not None and abs(coef) < cut_off:
            continue
        terms.append(TermProductForm(coef, dofs, [f"q^{deg[i]}" for i in dofs]))
    for t in terms:
        (ham.onesite if len(t.op_dofs) == 1 else ham.general)[0][0].append(t)
    return ham
