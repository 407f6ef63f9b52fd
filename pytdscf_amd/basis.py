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
