"""Module alias for ``pytdscf.dvr_operator_cls``."""
from .api import TensorOperator  # noqa: F401
