"""Unit conversion factors with the reference's names (pytdscf/units.py:28-56).
Like the reference they are taken from ``scipy.constants`` (CODATA), so the two
packages agree bit for bit on the same SciPy."""

from scipy.constants import physical_constants as _pc

au_in_cm1 = _pc["atomic unit of energy"][0] / (_pc["speed of light in vacuum"][0] * 1.0e02) / _pc["Planck constant"][0]
Hartree_in_cm1 = au_in_cm1
au_in_fs = _pc["atomic unit of time"][0] / 1.0e-15
au_in_eV = _pc["Hartree energy in eV"][0]
Has_in_eV = au_in_eV
au_in_dalton = _pc["electron mass"][0] / _pc["atomic mass constant"][0]
au_in_AMU = au_in_dalton
au_in_angstrom = _pc["Bohr radius"][0] / 1.0e-10
Bohr_in_angstrom = au_in_angstrom
au_in_debye = _pc["atomic unit of electric dipole mom."][0] * _pc["speed of light in vacuum"][0] * 1.0e21
