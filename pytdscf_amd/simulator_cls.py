"""Module alias for ``pytdscf.simulator_cls``."""
from .api import Simulator  # noqa: F401
