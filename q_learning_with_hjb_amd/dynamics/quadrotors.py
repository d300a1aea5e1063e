"""Quadrotors2D (planar, n=6, m=2) and NearHoverQuadcopter (n=10, m=3),
reference dynamics/quadrotors.py:9-70 and :102-170."""
import numpy as np

from .. import _abi
from .dynamics_basic import Dynamics


class Quadrotors2D(Dynamics):
    _KIND = _abi.SYS_QUAD2D

    def __init__(self, config) -> None:
        self.g, self.m, self.r, self.I = config.g, config.m, config.r, config.I
        super().__init__(config)

    def _system_params(self, c):
        return np.array([c.m, c.r, c.I, c.g], np.float64)


class NearHoverQuadcopter(Dynamics):
    """state [p_x, p_y, p_z, theta_x, theta_y, v_x, v_y, v_z, omega_x, omega_y], control [Tz, Sx, Sy]"""
    _KIND = _abi.SYS_NEARHOVER

    def __init__(self, config) -> None:
        self.g, self.kT, self.m, self.n0 = config.g, config.kT, config.m, config.n0
        super().__init__(config)

    def _system_params(self, c):
        return np.array([c.g, c.m, c.kT, c.n0], np.float64)
