"""pytdscf_amd -- MI355X-native one-site TDVP sweep engine behind PyTDSCF's surface."""

from . import units  # noqa: F401
from .api import BasInfo, Model, Simulator, TensorHamiltonian, TensorOperator, WFunc  # noqa: F401
from .basis import Boson, Exciton, Exponential, HarmonicOscillator, PrimBas_HO, Sine  # noqa: F401
from .engine import MultiStateEngine, TDVPEngine, TDVPEnsemble  # noqa: F401
from . import dvr_operator_cls, hamiltonian_cls, kraus, spectra  # noqa: F401,E402
from .hamiltonian_cls import PolynomialHamiltonian, read_potential_nMR  # noqa: F401,E402
from .dvr_operator_cls import (  # noqa: F401,E402
    construct_fulldimensional,
    construct_kinetic_mpo,
    construct_kinetic_operator,
    construct_nMR_recursive,
)

__version__ = "0.1.0"

__all__ = [
    "TDVPEngine", "TDVPEnsemble", "MultiStateEngine", "Simulator", "Model", "BasInfo", "TensorHamiltonian", "TensorOperator", "WFunc",
    "Exciton", "Boson", "HarmonicOscillator", "Sine", "Exponential", "PrimBas_HO", "PolynomialHamiltonian", "read_potential_nMR", "units", "dvr_operator_cls", "kraus", "spectra",
    "construct_fulldimensional", "construct_kinetic_mpo", "construct_kinetic_operator", "construct_nMR_recursive",
    "__version__",
]
