"""Post-processing helpers with the reference's names (``pytdscf/util``)."""

from .read_nc import read_nc  # noqa: F401
