"""pytdscf_amd -- MI355X-native one-site TDVP sweep engine behind PyTDSCF's surface."""

from . import units  # noqa: F401
from .api import BasInfo, Model, Simulator, TensorHamiltonian, TensorOperator, WFunc  # noqa: F401
from .basis import Boson, Exciton, Exponential, HarmonicOscillator, Sine  # noqa: F401
from .engine import MultiStateEngine, TDVPEngine  # noqa: F401
from . import dvr_operator_cls  # noqa: F401,E402

__all__ = [
    "TDVPEngine", "MultiStateEngine", "Simulator", "Model", "BasInfo", "TensorHamiltonian", "TensorOperator", "WFunc",
    "Exciton", "Boson", "HarmonicOscillator", "Sine", "Exponential", "units",
]
