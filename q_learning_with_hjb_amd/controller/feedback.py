"""Shared machinery of the closed-form controllers: they all evaluate on the device through
hjbx_controller_* and can run a whole closed loop in ONE kernel launch (hjbx_rollout_feedback_*)."""
from __future__ import annotations

import numpy as np
import torch

from .. import _abi, _ops
from ..dynamics.dynamics_basic import Dynamics, _from_device, _to_device
from .controller_basic import Controller


class DeviceFeedbackController(Controller):
    """Base of LQR / hover / energy-shaping controllers: owns an `hjbx_controller` descriptor."""

    dynamics: Dynamics

    def _descriptor(self) -> _abi.HjbxController:
        raise NotImplementedError

    def get_control_efforts(self, x):
        """u (m,) or (B, m) for x (n,) or (B, n); numpy or torch like the Dynamics methods."""
        t, one, kind = _to_device(x)
        u = _ops.controller(self.dynamics.system, self._descriptor(), t)
        return _from_device(u, one, kind)

    def rollout(self, x0, steps, task=None, terminate=False, log_traj=True, log_u=True, log_cost=False, stop_at_target=False):
        """Closed loop `for t: u = self(x); x = simulate(x, u)` for `steps` steps, fused in one kernel.

        x0: (B, n) (or (n,)).  Returns a dict with time-major `traj` (steps+1, B, n), `u` (steps, B, m),
        optional `cost`, `total_cost`, `done_step`, `x_final`; numpy in -> numpy out.  `stop_at_target` ends an
        environment once it is inside the controller's target ball (`done_step` = that step index)."""
        t, one, kind = _to_device(x0)
        out = _ops.rollout_feedback(self.dynamics.system, self._descriptor(), t, int(steps), task=task,
                                    integrator=self.dynamics.integrator, terminate=terminate, log_traj=log_traj, log_u=log_u,
                                    log_cost=log_cost, stop_at_target=stop_at_target)
        if kind == "cuda":
            return out
        return {k: (None if v is None else (v.cpu().numpy() if kind == "numpy" else v.cpu())) for k, v in out.items()}
