"""LQR for LinearDynamics: u = clip(-K x)  (reference controller/lqr.py:6-26)."""
import numpy as np
import scipy.linalg

from .. import _abi
from .feedback import DeviceFeedbackController


class LQR(DeviceFeedbackController):
    def __init__(self, dynamics, Q: np.ndarray, R: np.ndarray) -> None:
        super().__init__()
        assert Q.ndim == 2
        assert R.ndim == 2
        assert Q.shape[0] == Q.shape[1]
        assert R.shape[0] == R.shape[1]
        assert dynamics.A.shape[1] == Q.shape[1]
        assert dynamics.B.shape[1] == R.shape[1]
        self.dynamics = dynamics
        self.Q = Q
        self.R = R
        self.P = scipy.linalg.solve_continuous_are(self.dynamics.A, self.dynamics.B, self.Q, self.R)
        self.K = np.dot(scipy.linalg.inv(self.R), np.dot(self.dynamics.B.T, self.P))
        self.umin, self.umax = self.dynamics.get_control_limit()

    def _descriptor(self):
        n, m = self.dynamics.get_dimension()
        return _abi.make_controller(_abi.CTRL_LINEAR_FEEDBACK, n, m, self.K, wrap_error=False)
