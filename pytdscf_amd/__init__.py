"""pytdscf_amd -- MI355X-native one-site TDVP sweep engine behind PyTDSCF's surface."""

from .engine import TDVPEngine  # noqa: F401

__all__ = ["TDVPEngine"]
