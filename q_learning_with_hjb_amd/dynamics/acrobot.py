"""Acrobot (Tedrake form), reference dynamics/acrobot.py:19-81.

Upstream's constructor is stale (it calls Dynamics.__init__ without a config); this class
takes a config like every other system (configs.defaults.acrobot_dynamics_config carries the
constants of acrobot.py:8-16) and also accepts no argument at all, like `Acrobot()` upstream."""
import numpy as np

from .. import _abi
from .dynamics_basic import Dynamics


class Acrobot(Dynamics):
    _KIND = _abi.SYS_ACROBOT

    def __init__(self, config=None) -> None:
        if config is None:
            from ..configs.defaults import acrobot_dynamics_config
            config = acrobot_dynamics_config()
        self.dim = 2
        self.m1, self.m2, self.l1, self.l2 = config.m1, config.m2, config.l1, config.l2
        self.I1, self.I2, self.g = config.I1, config.I2, config.g
        super().__init__(config)

    def _system_params(self, c):
        return np.array([c.m1, c.m2, c.l1, c.l2, c.I1, c.I2, c.g], np.float64)

    def get_M(self, x):
        k = self.m2 * self.l1 * self.l2 / 2 * np.cos(x[1])
        return np.array([[self.I1 + self.I2 + self.m2 * self.l1 ** 2 + 2 * k, self.I2 + k], [self.I2 + k, self.I2]])

    def get_C(self, x):
        k = self.m2 * self.l1 * self.l2 / 2 * np.sin(x[1])
        return np.array([[-2 * k * x[3], -k * x[3]], [k * x[2], 0]])

    def get_G(self, x):
        s12 = np.sin(x[0] + x[1])
        return np.array([(self.m1 * self.l1 / 2 + self.m2 * self.l1) * self.g * np.sin(x[0]) + self.m2 * self.g * self.l2 / 2 * s12,
                         self.m2 * self.g * self.l2 / 2 * s12])

    def get_B(self):
        return np.array([0, 1])

    def energy(self, x):
        """Total mechanical energy of one state (acrobot.py:61-70); host-side set-up helper."""
        q, dq = x[:2], x[2:]
        c1, c2 = np.cos(q[0]), np.cos(q[1])
        k = self.m2 * self.l1 * self.l2 / 2
        T1 = 0.5 * self.I1 * dq[0] ** 2
        T2 = 0.5 * (self.m2 * self.l1 ** 2 + self.I2 + 2 * k * c2) * dq[0] ** 2 + 0.5 * self.I2 * dq[1] ** 2 \
            + (self.I2 + k * c2) * dq[0] * dq[1]
        U = -self.m1 * self.g * self.l1 / 2 * c1 - self.m2 * self.g * (self.l1 * c1 + self.l2 / 2 * np.cos(q[0] + q[1]))
        return T1 + T2 + U
