"""Analytic minimum-time controller for the double integrator (|u| <= umax): bang-bang about the switching curve
p = -v|v| / (2 umax), zero inside the target ball -- `get_analytical_control` of the reference's
examples/double_integrator_optimal_time.ipynb (cell 18), used there as ground truth next to the level-set solution.
Runs in the `hjbx_controller` / `hjbx_rollout_feedback` kernels like the other closed-form laws."""
import numpy as np

from .. import _abi
from .feedback import DeviceFeedbackController


class DoubleIntegratorTimeOptimalController(DeviceFeedbackController):
    def __init__(self, dynamics, metric: float = 1e-4, xf=(0.0, 0.0)) -> None:
        super().__init__()
        n, m = dynamics.get_dimension()
        if (n, m) != (2, 1) or not np.allclose(np.asarray(dynamics.A), [[0, 1], [0, 0]]) or not np.allclose(np.asarray(dynamics.B), [[0], [1]]):
            raise ValueError("the analytic minimum-time law is for the double integrator A=[[0,1],[0,0]], B=[[0],[1]]")
        self.dynamics = dynamics
        self.metric = float(metric)          # squared radius of the target ball (the notebook's `metric`)
        self.xf = np.asarray(xf, np.float64)
        self.umin, self.umax = dynamics.get_control_limit()

    def _descriptor(self):
        return _abi.make_controller(_abi.CTRL_DI_TIME_OPTIMAL, 2, 1, np.zeros((1, 2)), xf=self.xf, wrap_error=False,
                                    eps_region=self.metric)

    def time_to_target(self, x0, max_time: float = 15.0):
        """Closed-loop time until |x - xf|^2 <= metric for each start state (the notebook's time-to-origin), in seconds;
        `max_time` where the target is not reached.  One fused kernel launch for the whole batch."""
        steps = int(round(max_time / self.dynamics.dt))
        out = self.rollout(x0, steps, log_traj=False, log_u=False, stop_at_target=True)
        ds = out["done_step"]
        return ds * self.dynamics.dt if isinstance(ds, np.ndarray) else ds.to(out["x_final"].dtype) * self.dynamics.dt
