"""Module alias for ``pytdscf.hamiltonian_cls``."""
from .api import TensorHamiltonian  # noqa: F401
