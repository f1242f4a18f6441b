"""MI355X-native monocular-depth training hot path: drop-in for xeTaiz/mono-depth-estimation's network/*.py (FCRN, BTS, MiDaS,
VNL, Eigen, DORN, MyNet), criteria.py and metrics.py surface, plus the input pipeline (augment) and the data-parallel gradient
exchange (dp).  All device arithmetic lives in libmde_hip.so (hand-written gfx950 HIP kernels behind the C ABI in
include/mde_hip.h); this package is the Python host that mirrors the reference's nn.Module / loss-callable interface.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
