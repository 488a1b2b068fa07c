"""Mirror of the ``spconv`` package namespace used by the reference (``import spconv.pytorch``)."""
from . import pytorch  # noqa: F401
