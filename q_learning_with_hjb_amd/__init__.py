"""q_learning_with_hjb_amd -- MI355X-native batched rollouts + HJB value learning.

Drop-in for the hot path of HaoxiangYou/Q_Learning_with_HJB: the `Dynamics` / `Controller` /
`VHJBController` surface of the reference, with every per-environment computation running as a
hand-written gfx950 HIP kernel behind a C ABI (include/hjbx.h, csrc/).  See DESIGN.md.
"""
from . import _abi
from ._abi import EULER, RK4, ZOH, RESIDUAL_NORMALISED, RESIDUAL_RAW, build_library

_ARITHMETICS = {"f32": 0, "bf16x3": 1, "f16x2": 2}


def set_value_network_arithmetic(name: str) -> str:
    """How the fused kernels form the float32 products of the ReLU value network (process-wide; include/hjbx.h HJBX_OPT_MLP_ARITHMETIC,
    DESIGN.md 4.5): "f32" (default: the f32 MFMA, bitwise an fmaf chain -- the reference's float32 network arithmetic), or the faster opt-in
    emulations "bf16x3" (three exact bfloat16 pieces per operand, 1.6x) and "f16x2" (two float16 pieces per operand, scaled per environment,
    2.8x; narrower than float32: 22 significant bits).  Returns the previous setting."""
    if name not in _ARITHMETICS:
        raise ValueError(f"arithmetic must be one of {sorted(_ARITHMETICS)}, got {name!r}")
    prev = _abi.set_option(_abi.OPT_MLP_ARITHMETIC, _ARITHMETICS[name])
    return {v: k for k, v in _ARITHMETICS.items()}[prev]


def value_network_arithmetic() -> str:
    return {v: k for k, v in _ARITHMETICS.items()}[_abi.set_option(_abi.OPT_MLP_ARITHMETIC, -1)]


__all__ = ["_abi", "EULER", "RK4", "ZOH", "RESIDUAL_NORMALISED", "RESIDUAL_RAW", "build_library", "set_value_network_arithmetic",
           "value_network_arithmetic"]
__version__ = "0.3.0"
