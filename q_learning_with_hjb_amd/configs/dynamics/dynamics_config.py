"""Dynamics config dataclasses -- field names and float32 conversion follow the reference
(configs/dynamics/dynamics_config.py:6-58) so existing call sites construct them unchanged."""
from dataclasses import dataclass
from typing import Sequence

import numpy as np

from ..gin_lite import configurable


@dataclass
class DynamicsConfig:
    seed: int
    dt: float
    umin: Sequence[float]
    umax: Sequence[float]
    x0_mean: Sequence[float]
    x0_std: Sequence[float]  # half-width of a UNIFORM draw, despite the name (dynamics_basic.py:29)

    def __post_init__(self):
        for k in ("x0_mean", "x0_std", "umin", "umax"):
            setattr(self, k, np.array(getattr(self, k), dtype=np.float32))
        self.state_dim = self.x0_mean.shape[0]
        self.control_dim = self.umin.shape[0]


@configurable
@dataclass
class LinearDynamicsConfig(DynamicsConfig):
    A: Sequence[Sequence[float]]
    B: Sequence[Sequence[float]]

    def __post_init__(self):
        super().__post_init__()
        self.A = np.array(self.A, dtype=np.float32)
        self.B = np.array(self.B, dtype=np.float32)


@configurable
@dataclass
class CartpoleDynamicsConfig(DynamicsConfig):
    mc: float
    mp: float
    g: float
    l: float


@configurable
@dataclass
class AcrobotDynamicsConfig(DynamicsConfig):
    """New: the reference hard-codes these in dynamics/acrobot.py:8-16 (its constructor is stale)."""
    l1: float
    l2: float
    m1: float
    m2: float
    I1: float
    I2: float
    g: float


@configurable
@dataclass
class Quadrotors2DConfig(DynamicsConfig):
    g: float
    m: float
    r: float
    I: float


@configurable
@dataclass
class NearHoverQuadcopterConfig(DynamicsConfig):
    g: float
    m: float
    kT: float
    n0: float
