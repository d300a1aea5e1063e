"""Hover LQR for Quadrotors2D and NearHoverQuadcopter: u = clip(-K wrap(x - xf) + uf)
(reference controller/quadrotors_model_based_controller.py:7-38 and :40-75).
The min-snap waypoint planner of that file (:77-233) is never called by a rollout and is out of scope."""
import numpy as np
import scipy.linalg

from .. import _abi
from .feedback import DeviceFeedbackController


class _HoverLQR(DeviceFeedbackController):
    def _finish(self):
        self.P = scipy.linalg.solve_continuous_are(self.A, self.B, self.Q, self.R)
        self.K = np.dot(scipy.linalg.inv(self.R), np.dot(self.B.T, self.P))

    def _descriptor(self):
        n, m = self.dynamics.get_dimension()
        return _abi.make_controller(_abi.CTRL_LINEAR_FEEDBACK, n, m, self.K, xf=self.xf, uf=self.uf, wrap_error=True)


class Quadrotors2DHoveringController(_HoverLQR):
    def __init__(self, dynamics, xf: np.ndarray, Q: np.ndarray, R: np.ndarray) -> None:
        super().__init__()
        self.dynamics, self.xf, self.Q, self.R = dynamics, np.asarray(xf), np.asarray(Q), np.asarray(R)
        self.umin, self.umax = self.dynamics.get_control_limit()
        if np.linalg.norm(self.xf[2:]) > 0:
            raise ValueError("Final Velocity or Angle is not zero")
        d = self.dynamics
        self.uf = d.m * d.g / 2 * np.ones(2)
        self.A = np.vstack([np.hstack([np.zeros((3, 3)), np.eye(3)]), np.array([0, 0, -d.g, 0, 0, 0]), np.zeros((2, 6))])
        self.B = np.vstack([np.zeros((4, 2)), np.ones((1, 2)) / d.m, np.array([d.r / d.I, -d.r / d.I])])
        self._finish()


class NearHoverQuadcopterHoveringController(_HoverLQR):
    def __init__(self, dynamics, xf: np.ndarray, Q: np.ndarray, R: np.ndarray) -> None:
        super().__init__()
        self.dynamics, self.xf, self.Q, self.R = dynamics, np.asarray(xf), np.asarray(Q), np.asarray(R)
        self.umin, self.umax = self.dynamics.get_control_limit()
        if np.linalg.norm(self.xf[3:]) > 0:
            raise ValueError("Final Velocity or Angle is not zero")
        d = self.dynamics
        self.uf = np.array([d.g * d.m / d.kT, 0, 0])
        self.A = np.vstack([np.hstack([np.zeros((5, 5)), np.eye(5)]),
                            np.array([0, 0, 0, d.g, 0, 0, 0, 0, 0, 0]),
                            np.array([0, 0, 0, 0, d.g, 0, 0, 0, 0, 0]),
                            np.zeros((3, 10))])
        self.B = np.vstack([np.zeros((7, 3)), np.array([d.kT / d.m, 0, 0]), np.array([0, d.n0, 0]), np.array([0, 0, d.n0])])
        self._finish()
