"""Mirror of the reference's ``network`` package surface for the FCRN path."""
from . import FCRN  # noqa: F401
