"""Dynamics configuration records.

Field names (and the float32 conversion of the array fields) follow the reference's dataclasses
(configs/dynamics/dynamics_config.py:6-58) so existing call sites and .gin files construct them
unchanged; `gin_lite.configurable` fills unspecified constructor arguments from parsed bindings.
Added here: validation, `AcrobotDynamicsConfig` (the reference hard-codes those constants in
dynamics/acrobot.py:8-16) and `system_params()` = the parameter packing of `hjbx_system_create`."""
from dataclasses import dataclass
from typing import Sequence

import numpy as np

from ..gin_lite import configurable

_ARRAY_FIELDS = ("x0_mean", "x0_std", "umin", "umax")


def _f32(values) -> np.ndarray:
    return np.array(values, dtype=np.float32)


@dataclass
class DynamicsConfig:
    # RNG seed for NumPy's global generator (the reference seeds it in Dynamics.__init__)
    seed: int
    # integration step [s]
    dt: float
    # control box
    umin: Sequence[float]
    umax: Sequence[float]
    # initial states are drawn UNIFORMLY from x0_mean +- x0_std (despite the name; dynamics_basic.py:29)
    x0_mean: Sequence[float]
    x0_std: Sequence[float]

    def __post_init__(self):
        for name in _ARRAY_FIELDS:
            setattr(self, name, _f32(getattr(self, name)))
        self.state_dim = int(self.x0_mean.shape[0])
        self.control_dim = int(self.umin.shape[0])
        self.validate()

    def validate(self):
        if self.x0_std.shape != self.x0_mean.shape:
            raise ValueError(f"x0_std {self.x0_std.shape} and x0_mean {self.x0_mean.shape} differ in shape")
        if self.umax.shape != self.umin.shape:
            raise ValueError(f"umin {self.umin.shape} and umax {self.umax.shape} differ in shape")
        if not np.all(self.umin <= self.umax):
            raise ValueError("umin must not exceed umax")
        if not self.dt > 0:
            raise ValueError("dt must be positive")

    def system_params(self) -> np.ndarray:
        raise NotImplementedError


@configurable
@dataclass
class LinearDynamicsConfig(DynamicsConfig):
    A: Sequence[Sequence[float]]
    B: Sequence[Sequence[float]]

    def __post_init__(self):
        self.A, self.B = _f32(self.A), _f32(self.B)
        super().__post_init__()

    def discretize(self):
        """(Ad, Bd): exact zero-order-hold discretisation over dt, expm([[A, B], [0, 0]] dt) -- what
        scipy.signal.cont2discrete returns in the reference's notebooks."""
        import scipy.linalg
        n, m = self.A.shape[0], self.B.shape[1]
        aug = np.zeros((n + m, n + m))
        aug[:n, :n] = self.A.astype(np.float64)
        aug[:n, n:] = self.B.astype(np.float64)
        E = scipy.linalg.expm(aug * float(self.dt))
        return E[:n, :n].copy(), E[:n, n:].copy()

    def system_params(self):
        Ad, Bd = self.discretize()   # A, B, then Ad, Bd for the HJBX_ZOH integrator
        return np.concatenate([self.A.astype(np.float64).ravel(), self.B.astype(np.float64).ravel(), Ad.ravel(), Bd.ravel()])


@configurable
@dataclass
class CartpoleDynamicsConfig(DynamicsConfig):
    mc: float  # cart mass
    mp: float  # pole mass
    g: float
    l: float   # pole length

    def system_params(self):
        return np.array([self.mc, self.mp, self.l, self.g], np.float64)


@configurable
@dataclass
class AcrobotDynamicsConfig(DynamicsConfig):
    l1: float
    l2: float
    m1: float
    m2: float
    I1: float
    I2: float
    g: float

    def system_params(self):
        return np.array([self.m1, self.m2, self.l1, self.l2, self.I1, self.I2, self.g], np.float64)


@configurable
@dataclass
class Quadrotors2DConfig(DynamicsConfig):
    g: float
    m: float
    r: float  # rotor arm
    I: float  # moment of inertia

    def system_params(self):
        return np.array([self.m, self.r, self.I, self.g], np.float64)


@configurable
@dataclass
class NearHoverQuadcopterConfig(DynamicsConfig):
    g: float
    m: float
    kT: float  # thrust coefficient
    n0: float  # angular-rate gain

    def system_params(self):
        return np.array([self.g, self.m, self.kT, self.n0], np.float64)
