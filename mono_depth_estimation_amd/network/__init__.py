"""Mirror of the reference's ``network`` package: FCRN (network/FCRN.py), Bts, MiDaS, VNL, Eigen, Dorn, MyNet -- same module
names, class names, constructor arguments and state_dict keys; every forward / backward runs on libmde_hip.so."""
from . import Bts, Dorn, Eigen, FCRN, MiDaS, MyNet, VNL  # noqa: F401
