"""Minimal primitive bases of the user surface: what an input
script needs to size the sites and to build small d x d operator matrices.
Interfaces follow pytdscf/basis/{exciton,boson,ho,abc}.py; the maths is the
textbook form (no reference code)."""

from __future__ import annotations

import math

import numpy as np

from . import units as _units


class Exciton:
    """N-level electronic site; ``pytdscf.basis.Exciton(nstate, names=None)``."""

    def __init__(self, nstate: int, names: list[str] | None = None) -> None:
        self.nstate = int(nstate)
        self.names = [f"S{i}" for i in range(nstate)] if names is None else list(names)
        if len(self.names) != self.nstate or not all(isinstance(n, str) for n in self.names):
            raise AssertionError("names must be a list of nstate strings")

    def get_annihilation_matrix(self) -> np.ndarray:
        return np.diag(np.ones(self.nstate - 1), k=1)  # |i><i+1|

    def get_creation_matrix(self) -> np.ndarray:
        return self.get_annihilation_matrix().T

    @property
    def nprim(self) -> int:
        return self.nstate

    def __len__(self) -> int:
        return self.nstate


class Boson:
    """Truncated bosonic mode; ``pytdscf.basis.Boson(nstate)``: b, b^dagger, number."""

    def __init__(self, nstate: int) -> None:
        self.nstate = int(nstate)

    def get_annihilation_matrix(self) -> np.ndarray:
        return np.diag(np.sqrt(np.arange(1, self.nstate)), k=1)

    def get_creation_matrix(self) -> np.ndarray:
        return self.get_annihilation_matrix().T

    def get_number_matrix(self) -> np.ndarray:
        return np.diag(np.arange(self.nstate, dtype=float))

    @property
    def nprim(self) -> int:
        return self.nstate

    def __len__(self) -> int:
        return self.nstate


class HarmonicOscillator:
    """Harmonic-oscillator DVR in mass-weighted coordinates,
    ``HarmonicOscillator(ngrid, omega, q_eq=0.0, units="cm-1")``
    (pytdscf/basis/ho.py:53-86, DVR conventions of basis/abc.py:117-170):
    grids = eigenvalues of <n|q|n'>, ``get_unitary()[n, alpha]`` with the sign of
    every grid column fixed so that the ground-state row is positive."""

    def __init__(self, ngrid: int, omega: float, q_eq: float = 0.0, units: str = "cm-1", dimnsionless: bool = False):
        self.ngrid = int(ngrid)
        u = units.lower()
        if u in ("cm1", "cm-1", "kaiser"):
            self.omega = omega / _units.au_in_cm1
        elif u in ("au", "hartree", "a.u."):
            self.omega = float(omega)
        elif u == "ev":
            self.omega = omega / _units.au_in_eV
        else:
            raise ValueError(f"{units} must be [cm1, au, eV]")
        self.freq_cm1 = self.omega * _units.au_in_cm1
        self.q_eq = q_eq / math.sqrt(self.omega) if dimnsionless else q_eq
        self._grids = None
        self._unitary = None

    @property
    def nprim(self) -> int:
        return self.ngrid

    def __len__(self) -> int:
        return self.ngrid

    def get_pos_rep_matrix(self) -> np.ndarray:
        n = np.arange(1, self.ngrid)
        off = np.sqrt(n / (2.0 * self.omega))
        return np.diag(off, 1) + np.diag(off, -1) + self.q_eq * np.eye(self.ngrid)

    def get_1st_derivative_matrix_fbr(self) -> np.ndarray:
        n = np.arange(1, self.ngrid)
        off = np.sqrt(n * self.omega / 2.0)
        return np.diag(off, 1) - np.diag(off, -1)  # <n-1|d/dq|n> = sqrt(n w/2)

    def get_2nd_derivative_matrix_fbr(self) -> np.ndarray:
        n = np.arange(self.ngrid)
        d2 = np.diag(-self.omega * (n + 0.5))
        off = 0.5 * self.omega * np.sqrt((n[:-2] + 1.0) * (n[:-2] + 2.0))
        return d2 + np.diag(off, 2) + np.diag(off, -2)

    def _diag(self):
        if self._grids is None:
            val, vec = np.linalg.eigh(self.get_pos_rep_matrix())
            vec = vec * np.where(vec[0, :] < 0, -1.0, 1.0)[None, :]
            self._grids, self._unitary = val, vec

    def get_grids(self):
        self._diag()
        return list(self._grids)

    def get_unitary(self) -> np.ndarray:
        self._diag()
        return self._unitary

    def get_1st_derivative_matrix_dvr(self) -> np.ndarray:
        u = self.get_unitary()
        return u.conj().T @ self.get_1st_derivative_matrix_fbr() @ u

    def get_2nd_derivative_matrix_dvr(self) -> np.ndarray:
        u = self.get_unitary()
        return u.conj().T @ self.get_2nd_derivative_matrix_fbr() @ u


class _GridDVR:
    """Shared surface of the equidistant-grid DVRs (pytdscf/basis/abc.py:17-170): ``nprim``,
    ``len()``, iteration over the grid, FBR <-> DVR transformation of the derivative matrices."""

    ngrid: int

    @property
    def nprim(self) -> int:
        return self.ngrid

    def __len__(self) -> int:
        return self.ngrid

    def __iter__(self):
        return iter(self.get_grids())

    def __call__(self, n: int, q):
        return self.dvr_func(n, q)

    def get_grids(self):
        return list(self._grids)

    def get_sqrt_weights(self, k: int = 0):
        return [math.sqrt(self.deltax)] * self.ngrid

    def dvr_func(self, n: int, q):
        """chi_n(q) = sum_j phi_j(q) U[j, n] (abc.py:172-187)."""
        u = self.get_unitary()
        return sum(self.fbr_func(j, q) * u[j, n] for j in range(self.ngrid))


class Sine(_GridDVR):
    """Sine (particle-in-a-box) DVR, ``Sine(ngrid, length, x0=0.0, units="angstrom",
    doAnalytical=True, include_terminal=True)`` (pytdscf/basis/sin.py:15-66; Beck et al.,
    Phys. Rep. 324, 1 (2000), appendix B.4.2).  phi_j(x) = sqrt(2/L) sin(j pi (x - x0) / L),
    j = 1..N; grid x_a = x0 + a L / (N + 1), a = 1..N (the box ends are not grid points; with
    ``include_terminal`` the box is widened by one spacing on either side so that the first and
    last grid points are the given ends)."""

    def __init__(self, ngrid: int, length: float, x0: float = 0.0, units: str = "angstrom", doAnalytical: bool = True,
                 include_terminal: bool = True):
        if type(ngrid) is not int:
            raise TypeError(f"ngrid argument must be integer but {ngrid} is given.")
        if not doAnalytical:
            raise NotImplementedError("Sine DVR: only the analytical matrices (doAnalytical=True)")
        u = units.lower()
        if u in ("angstrom", "å"):
            self.L, self.x0 = length / _units.au_in_angstrom, x0 / _units.au_in_angstrom
        elif u in ("bohr", "a.u.", "au"):
            self.L, self.x0 = float(length), float(x0)
        else:
            raise NotImplementedError
        self.ngrid = ngrid
        if include_terminal:
            dx = self.L / (ngrid - 1)
            self.x0 -= dx
            self.L = (ngrid + 1) * dx
        self.lb, self.ub = x0, x0 + self.L  # as the reference sets them (sin.py:62-63)
        self.label = "Sine"
        self.doAnalytical = True
        self.deltax = self.L / (ngrid + 1)
        self._grids = self.x0 + self.deltax * np.arange(1, ngrid + 1)

    def fbr_func(self, n: int, x):
        x = np.asarray(x, dtype=float)
        inside = (self.x0 <= x) & (x <= self.x0 + self.L)
        return math.sqrt(2.0 / self.L) * np.sin((n + 1) * math.pi * (x - self.x0) / self.L) * inside

    def get_pos_rep_matrix(self) -> np.ndarray:
        """<phi_j| cos(pi (x - x0) / L) |phi_k> = (delta_{j,k+1} + delta_{j,k-1}) / 2."""
        off = 0.5 * np.ones(self.ngrid - 1)
        return np.diag(off, 1) + np.diag(off, -1)

    def get_unitary(self) -> np.ndarray:
        """U[j, a] = sqrt(2 / (N + 1)) sin(j a pi / (N + 1)) (symmetric)."""
        k = np.arange(1, self.ngrid + 1)
        return math.sqrt(2.0 / (self.ngrid + 1)) * np.sin(np.outer(k, k) * math.pi / (self.ngrid + 1))

    def get_1st_derivative_matrix_fbr(self) -> np.ndarray:
        """Antisymmetric, non-zero for odd j - k.  NOTE: the published element is
        (4 / L) j k / (j^2 - k^2); the reference evaluates (4 / L) j k (j + k) / (j - k)
        (sin.py:127-133) and that is what is reproduced here, so that scripts give the same numbers."""
        j = np.arange(1, self.ngrid + 1, dtype=float)
        J, K = np.meshgrid(j, j, indexing="ij")
        with np.errstate(divide="ignore", invalid="ignore"):
            m = 4.0 / self.L * J * K * (J + K) / (J - K)
        m[(np.abs(J - K) % 2) == 0] = 0.0
        return m

    def get_2nd_derivative_matrix_fbr(self) -> np.ndarray:
        return -np.diag((math.pi / self.L * np.arange(1, self.ngrid + 1)) ** 2)

    def get_1st_derivative_matrix_dvr(self) -> np.ndarray:
        u = self.get_unitary()
        return u.conj().T @ self.get_1st_derivative_matrix_fbr() @ u

    def get_2nd_derivative_matrix_dvr(self) -> np.ndarray:
        """Closed form of U^T D2 U (appendix B.4.2 with the sign of the first diagonal term
        corrected, as in sin.py:160-186)."""
        n1 = self.ngrid + 1
        a = np.arange(1, self.ngrid + 1) * math.pi / n1
        s, c = np.sin(a), np.cos(a)
        sign = (-1.0) ** np.abs(np.subtract.outer(np.arange(self.ngrid), np.arange(self.ngrid)))
        with np.errstate(divide="ignore", invalid="ignore"):
            m = 2.0 * sign / n1 ** 2 * np.outer(s, s) / np.subtract.outer(c, c) ** 2
        np.fill_diagonal(m, 1.0 / 3.0 + 1.0 / (6.0 * n1 ** 2) - 1.0 / (2.0 * (n1 * s) ** 2))
        return -((math.pi / self.deltax) ** 2) * m


class Exponential(_GridDVR):
    """Exponential (Fourier, periodic) DVR, ``Exponential(ngrid, length, x0=0.0)``
    (pytdscf/basis/exponential.py:10-66; Colbert & Miller, J. Chem. Phys. 96, 1982 (1992)):
    phi_j(x) = exp(2 pi i j (x - x0) / L) / sqrt(L), j = -(N-1)/2..(N-1)/2, N odd, grid
    x_a = x0 + a L / N, a = 0..N-1."""

    def __init__(self, ngrid: int, length: float, x0: float = 0.0, doAnalytical: bool = True):
        if type(ngrid) is not int:
            raise TypeError(f"ngrid argument must be integer but {ngrid} is given.")
        if ngrid % 2 == 0:
            raise ValueError("ngrid must be odd number.")
        if not doAnalytical:
            raise NotImplementedError("Numerical Integral of complex exponential is somehow difficult.")
        self.ngrid = ngrid
        self.x0, self.L = float(x0), float(length)
        self.lb, self.ub = self.x0, self.x0 + self.L
        self.label = "Exponential"
        self.doAnalytical = True
        self.deltax = self.L / ngrid
        self._grids = self.x0 + self.deltax * np.arange(ngrid)

    def fbr_func(self, n: int, x):
        j = n - self.ngrid // 2
        return np.exp(2.0j * math.pi * j * (np.asarray(x, dtype=float) - self.x0) / self.L) / math.sqrt(self.L)

    def get_pos_rep_matrix(self) -> np.ndarray:
        raise NotImplementedError

    def get_unitary(self) -> np.ndarray:
        """U[j, a] = conj(phi_j(x_a)) (exponential.py:196-211; not normalised to a unitary there)."""
        j = np.arange(self.ngrid) - self.ngrid // 2
        return np.exp(-2.0j * math.pi * np.outer(j, self._grids - self.x0) / self.L) / math.sqrt(self.L)

    def _diff(self):
        return np.subtract.outer(np.arange(self.ngrid), np.arange(self.ngrid))

    def get_1st_derivative_matrix_dvr(self) -> np.ndarray:
        """(pi / L) (-1)^(a-b) / sin(pi (a - b) / N) for a != b, stored SYMMETRICALLY like the
        reference does (exponential.py:93-106; the operator itself is antisymmetric)."""
        k = np.abs(self._diff())
        with np.errstate(divide="ignore", invalid="ignore"):
            m = math.pi / self.L * (-1.0) ** k / np.sin(-math.pi * k / self.ngrid)
        np.fill_diagonal(m, 0.0)
        return m

    def get_2nd_derivative_matrix_dvr(self) -> np.ndarray:
        k = self._diff()
        with np.errstate(divide="ignore", invalid="ignore"):
            m = -2.0 * math.pi ** 2 / self.L ** 2 * (-1.0) ** np.abs(k) * np.cos(math.pi * k / self.ngrid) / np.sin(math.pi * k / self.ngrid) ** 2
        np.fill_diagonal(m, -(math.pi ** 2) / 3.0 / self.L ** 2 * (self.ngrid ** 2 - 1))
        return m

    def get_1st_derivative_matrix_fbr(self) -> np.ndarray:
        u = self.get_unitary()
        return u @ self.get_1st_derivative_matrix_dvr() @ u.T  # as exponential.py:108-115

    def get_2nd_derivative_matrix_fbr(self) -> np.ndarray:
        u = self.get_unitary()
        return u @ self.get_2nd_derivative_matrix_dvr() @ u.T


class PrimBas_HO:
    """Harmonic-oscillator EIGENFUNCTION primitive basis (FBR) in mass-weighted coordinates,
    ``PrimBas_HO(origin, freq_cm1, nprim, origin_is_dimless=True)`` (pytdscf/basis/ho.py:255-315):
    chi_n(q) = HO eigenfunction n of frequency omega centred at ``origin_mwc``.  The operator
    matrices are the exact integrals <chi_m| q^k |chi_n>, <chi_m| d^k/dq^k |chi_n> (not powers of
    a truncated matrix): ladder operators in a basis enlarged by k, then cut back."""

    def __init__(self, origin: float, freq_cm1: float, nprim: int, origin_is_dimless: bool = True):
        self.freq_cm1 = freq_cm1
        self.nprim = int(nprim)
        self.freq_au = freq_cm1 / _units.au_in_cm1
        if origin_is_dimless:
            self.origin_mwc = origin / math.sqrt(self.freq_au)
            self.origin = origin
        else:
            self.origin_mwc = origin
            self.origin = origin * math.sqrt(self.freq_au)

    def __len__(self) -> int:
        return self.nprim

    def _ladder(self, extra: int):
        n = self.nprim + extra
        a = np.diag(np.sqrt(np.arange(1, n)), 1)  # annihilation operator
        return a, a.T

    def q_matrix(self, order: int = 1) -> np.ndarray:
        """<m| q^order |n> with q the absolute mass-weighted coordinate (basis centred at origin_mwc)."""
        a, ad = self._ladder(order)
        q = self.origin_mwc * np.eye(a.shape[0]) + (a + ad) / math.sqrt(2.0 * self.freq_au)
        return np.linalg.matrix_power(q, order)[: self.nprim, : self.nprim]

    def d_matrix(self, order: int = 1) -> np.ndarray:
        """<m| d^order/dq^order |n>; d/dq = sqrt(omega / 2) (a - a^+)."""
        a, ad = self._ladder(order)
        d = math.sqrt(self.freq_au / 2.0) * (a - ad)
        return np.linalg.matrix_power(d, order)[: self.nprim, : self.nprim]

    def op_matrix(self, key: str) -> np.ndarray:
        """Site matrix of an operator key of the polynomial Hamiltonians: "ovlp", "q^k" ("q" = "q^1"),
        "d^k", "ham1" (= -d^2/2 + omega^2 (q - q0)^2 / 2, diagonal omega (n + 1/2))."""
        if key == "ovlp":
            return np.eye(self.nprim)
        if key == "ham1":
            return np.diag(self.freq_au * (np.arange(self.nprim) + 0.5))
        name, _, power = key.partition("^")
        k = int(power) if power else 1
        if name == "q":
            return self.q_matrix(k)
        if name == "d":
            return self.d_matrix(k)
        raise ValueError(f"unknown operator key {key!r}")
