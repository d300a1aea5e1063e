"""LinearDynamics: x_dot = A x + B u  (reference dynamics/linear.py:7-22)."""
import numpy as np

from .. import _abi
from .dynamics_basic import Dynamics


class LinearDynamics(Dynamics):
    _KIND = _abi.SYS_LINEAR

    def __init__(self, config) -> None:
        assert config.A.ndim == 2
        assert config.B.ndim == 2
        assert config.A.shape[0] == config.B.shape[0]
        assert config.A.shape[0] == config.A.shape[1]
        self.A = config.A
        self.B = config.B
        self.A_d, self.B_d = config.discretize()   # used when `self.integrator = ZOH`
        super().__init__(config)

    def _system_params(self, config):
        return config.system_params()
