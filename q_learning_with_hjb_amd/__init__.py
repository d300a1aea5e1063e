"""q_learning_with_hjb_amd -- MI355X-native batched rollouts + HJB value learning.

Drop-in for the hot path of HaoxiangYou/Q_Learning_with_HJB: the `Dynamics` / `Controller` /
`VHJBController` surface of the reference, with every per-environment computation running as a
hand-written gfx950 HIP kernel behind a C ABI (include/hjbx.h, csrc/).  See DESIGN.md.
"""
from . import _abi
from ._abi import EULER, RK4, ZOH, RESIDUAL_NORMALISED, RESIDUAL_RAW, build_library

__all__ = ["_abi", "EULER", "RK4", "ZOH", "RESIDUAL_NORMALISED", "RESIDUAL_RAW", "build_library"]
__version__ = "0.1.0"
