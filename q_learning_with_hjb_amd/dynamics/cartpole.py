"""Cartpole (theta = pi upright), reference dynamics/cartpole.py:10-64."""
import numpy as np

from .. import _abi
from .dynamics_basic import Dynamics


class Cartpole(Dynamics):
    _KIND = _abi.SYS_CARTPOLE

    def __init__(self, config) -> None:
        self.mc, self.mp, self.l, self.g = config.mc, config.mp, config.l, config.g
        super().__init__(config)

    def _system_params(self, config):
        return config.system_params()

    # Manipulator terms for ONE state -- set-up helpers (linearisation); the batched path never
    # forms them (the kernels use the reduced closed form, csrc/hjbx_systems.hpp).
    def get_M(self, x):
        c = np.cos(x[1])
        return np.array([[self.mc + self.mp, self.mp * self.l * c], [self.mp * self.l * c, self.mp * self.l ** 2]])

    def get_C(self, x):
        return np.array([[0, -self.mp * self.l * x[3] * np.sin(x[1])], [0, 0]])

    def get_G(self, x):
        return np.array([0, self.mp * self.g * self.l * np.sin(x[1])])

    def get_B(self):
        return np.array([1, 0])
