"""Module alias so scripts written for ``pytdscf.model_cls`` import unchanged."""
from .api import BasInfo, Model  # noqa: F401
