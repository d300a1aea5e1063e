"""Infinite-horizon LQR for `LinearDynamics`: u = clip(-K x, umin, umax), K = R^-1 B' P with P the
stabilising CARE solution (reference controller/lqr.py:6-26).  The gain is computed once on the host;
the feedback itself runs in the `hjbx_controller` / `hjbx_rollout_feedback` kernels."""
import numpy as np
import scipy.linalg

from .. import _abi
from .feedback import DeviceFeedbackController


def _square(name, a):
    a = np.asarray(a)
    if a.ndim != 2 or a.shape[0] != a.shape[1]:
        raise AssertionError(f"{name} must be a square matrix, got shape {a.shape}")
    return a


class LQR(DeviceFeedbackController):
    def __init__(self, dynamics, Q: np.ndarray, R: np.ndarray) -> None:
        super().__init__()
        Q, R = _square("Q", Q), _square("R", R)
        n, m = dynamics.A.shape[1], dynamics.B.shape[1]
        if Q.shape[0] != n or R.shape[0] != m:
            raise AssertionError(f"Q is {Q.shape} and R is {R.shape} but the system has n={n}, m={m}")
        self.dynamics, self.Q, self.R = dynamics, Q, R
        A, B = np.asarray(dynamics.A, np.float64), np.asarray(dynamics.B, np.float64)
        self.P = scipy.linalg.solve_continuous_are(A, B, Q, R)
        self.K = scipy.linalg.solve(R, B.T @ self.P)          # R^-1 B' P, (m, n)
        self.umin, self.umax = dynamics.get_control_limit()

    def _descriptor(self):
        n, m = self.dynamics.get_dimension()
        # the reference feeds the raw state (no target, no wrap): lqr.py:26
        return _abi.make_controller(_abi.CTRL_LINEAR_FEEDBACK, n, m, self.K, wrap_error=False)
