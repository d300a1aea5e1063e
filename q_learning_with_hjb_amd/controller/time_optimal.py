"""Analytic minimum-time controller for the double integrator (|u| <= umax): bang-bang about the switching curve
p = -v|v| / (2 umax), zero inside the target ball -- `get_analytical_control` of the reference's
examples/double_integrator_optimal_time.ipynb (cell 18), used there as ground truth next to the level-set solution.
Runs in the `hjbx_controller` / `hjbx_rollout_feedback` kernels like the other closed-form laws."""
import numpy as np
import torch

from .. import _abi
from .controller_basic import Controller
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


class TimeOptimalVHJBController(Controller):
    """Value learning for minimum-time control: the "ours" experiment of the reference's
    examples/double_integrator_optimal_time.ipynb (cells 5, 7, 9, 11), batched on the device.

    * value net: `PDValueApproximator` of cell 5 = the VHJB network V = |MLP(x)|^2 + 1e-3 |x|^2 with `sin` activations
      (`activation="relu"` gives the stock vhjb.py network; all three activations run on the fused MFMA kernels);
    * control law u = -sign(gradV @ B) (cells 9, 11) generalised to a box: `hjbx_task.law = HJBX_LAW_BANGBANG`;
    * data: `num_states` points uniform in xf +- state_halfwidth, running cost 1 outside the target ball |e|^2 <= metric
      and 0 inside (cell 7) -- computed in the residual kernel, not stored;
    * loss: mean |gradV . (f1 + f2 u) + running_cost| (pd_hjb_loss, cell 11) = `hjbx_hjb_residual` in RAW mode with
      done = 0; Adam(lr); minibatches of `batch_size`, shuffled, last partial batch kept (drop_last=False, cell 7);
    * evaluation: time until |e|^2 <= metric from random starts, capped at `max_T` (get_mean_and_std_of_policy, cell 9).
    Works for any control-affine system of this package; the notebook's instance is the double integrator with dt = 0.01,
    |u| <= 1 and zero-order-hold stepping (cell 4).  PARITY UNPINNED: the notebook needs JAX and records no numbers."""

    def __init__(self, dynamics, metric: float = 1e-4, xf=None, features=(128, 128, 64), activation: str = "sin",
                 epsilon_scalar: float = 1e-3, lr: float = 1e-3, batch_size: int = 256, num_states: int = 2 ** 16,
                 state_halfwidth=1.0, seed: int = 0, device=None, dtype=None, obs_min=None, obs_max=None) -> None:
        super().__init__()
        import torch
        from .. import _ops
        from .vhjb import ValueFunctionApproximator
        self.dynamics = dynamics
        n, m = dynamics.get_dimension()
        self.state_dim, self.control_dim = n, m
        self.metric = float(metric)
        self.xf = np.zeros(n) if xf is None else np.asarray(xf, np.float64)
        self.device = torch.device(device) if device is not None else _ops.require_device()
        self.dtype = torch.float32 if dtype is None else dtype
        self.umin, self.umax = dynamics.get_control_limit()
        self._task = _abi.make_task(n, m, np.eye(n), np.eye(m), None, self.xf, np.zeros(m), obs_min, obs_max, 0.0,
                                    law=_abi.LAW_BANGBANG, target_r2=self.metric)
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(seed)
        self.value_function_approximator = ValueFunctionApproximator(
            dynamics, features, np.zeros(n), np.ones(n), self.xf, epsilon_scalar, dtype=self.dtype, device=self.device,
            generator=self._gen, activation=activation)
        builtin = dynamics.system.kind != _abi.SYS_USER
        self.fused = (activation in ValueFunctionApproximator.FUSED_ACTIVATIONS and self.dtype == torch.float32 and tuple(features) == (128, 128, 64)
                      and builtin)
        # the parameter gradient of pd_hjb_loss through hjbx_value_loss_grad_f32 (RAW residual, done = 0) and the Adam step through
        # hjbx_mix_adam_f32, like VHJBController (the sin network: state dimension <= 4); HJBX_FUSED_PARAM_GRAD=0 keeps PyTorch autograd
        import os
        self.fused_param_grad = (self.fused and (activation != "sin" or n <= 4) and self.device.type == "cuda"
                                 and os.environ.get("HJBX_FUSED_PARAM_GRAD", "1") != "0")
        self._adam_ticket = torch.zeros((1,), dtype=torch.int32, device=self.device) if self.fused_param_grad else None
        self._ones = None
        self.graph_updates = self.device.type == "cuda"   # optimiser steps of `train` replay from a hipGraph
        self._graphed = None
        self.optimizer = torch.optim.Adam(self.value_function_approximator.parameters(), lr=lr, betas=(0.9, 0.999), eps=1e-8,
                                          capturable=self.graph_updates)
        self.batch_size = int(batch_size)
        hw = torch.as_tensor(np.broadcast_to(np.asarray(state_halfwidth, np.float64), (n,)).copy(), dtype=self.dtype, device=self.device)
        self._halfwidth = hw
        xf_t = torch.as_tensor(self.xf, dtype=self.dtype, device=self.device)
        u01 = torch.rand((int(num_states), n), generator=self._gen, device=self.device, dtype=self.dtype)
        self.states = _ops.wrap(dynamics.system, ((2.0 * u01 - 1.0) * hw + xf_t).contiguous())
        self._zeros = torch.zeros(max(self.batch_size, 1), dtype=self.dtype, device=self.device)

    # -- policy ------------------------------------------------------------------------------------------
    def get_v_gradient(self, x):
        import torch
        with torch.no_grad():
            if self.fused:
                return self.value_function_approximator.fused_value_grad(x, want_v=False)[1]
            return self.value_function_approximator.value_and_grad(x)[1]

    def get_control_efforts(self, x):
        """u = umax / umin / 0 by the sign of f2' gradV (cell 9's get_control); numpy in -> numpy out, tensors stay on the device."""
        import torch
        from .. import _ops
        is_np = not isinstance(x, torch.Tensor)
        xt = torch.as_tensor(np.asarray(x), dtype=self.dtype, device=self.device) if is_np else x.to(self.dtype)
        single = xt.ndim == 1
        xt = xt.reshape(-1, self.state_dim).contiguous()
        u = _ops.control_from_grad(self.dynamics.system, self._task, xt, self.get_v_gradient(xt).contiguous())
        u = u[0] if single else u
        return u.cpu().numpy() if is_np else u

    def time_to_target(self, x0, max_time: float = 15.0):
        """Seconds until |x - xf|^2 <= metric under the learned law for each start state; `max_time` where it is not reached."""
        import torch
        from .. import _ops
        is_np = not isinstance(x0, torch.Tensor)
        x = torch.as_tensor(np.asarray(x0), dtype=self.dtype, device=self.device) if is_np else x0.to(dtype=self.dtype, device=self.device)
        x = x.reshape(-1, self.state_dim).contiguous()
        T = int(round(max_time / self.dynamics.dt))
        B = x.shape[0]
        sysh, integ = self.dynamics.system, self.dynamics.integrator
        done_step = torch.full((B,), -1, dtype=torch.int32, device=self.device)
        if self.fused:
            left, t0 = T + 1, 0
            while left > 0:  # chunks keep the (steps, B) cost / done logs small
                k = min(left, 256)
                out = _ops.vhjb_rollout(sysh, self._task, self.value_function_approximator.descriptor(), x, k, T, done_step, t_first=t0,
                                        integrator=integ, log_traj=False, log_u=False, log_residual=False, want_x_out=True)
                x, t0, left = out["x_out"], t0 + k, left - k
        else:
            x, xn = x.clone(), torch.empty_like(x)   # ping-pong buffers: never step into the caller's tensor
            c = torch.empty((B,), dtype=self.dtype, device=self.device)
            d = torch.empty_like(c)
            for t in range(T + 1):
                g = self.get_v_gradient(x).contiguous()
                _ops.vhjb_step(sysh, self._task, t, T, x, g, xn, c, d, done_step, integrator=integ)
                x, xn = xn, x
                if t % 64 == 63 and bool((done_step >= 0).all()):
                    break
        tt = done_step.clamp(min=0).to(self.dtype) * self.dynamics.dt
        return tt.cpu().numpy() if is_np else tt

    def get_mean_and_std_of_policy(self, num_of_trajectory: int = 20, max_T: float = 15.0):
        import torch
        u01 = torch.rand((num_of_trajectory, self.state_dim), generator=self._gen, device=self.device, dtype=self.dtype)
        x0 = (2.0 * u01 - 1.0) * self._halfwidth + torch.as_tensor(self.xf, dtype=self.dtype, device=self.device)
        tt = self.time_to_target(x0.contiguous(), max_T)
        return float(tt.mean()), float(tt.std(unbiased=False))

    # -- learning ----------------------------------------------------------------------------------------
    def hjb_loss(self, xs, weights=None):
        from .vhjb import _HJBResidualSum
        _, g = self.value_function_approximator.value_and_grad(xs, weights=weights)
        if self._zeros.shape[0] < xs.shape[0]:
            self._zeros = self._zeros.new_zeros(xs.shape[0])
        s, _ = _HJBResidualSum.apply(g, xs, self._zeros[:xs.shape[0]], self.dynamics.system, self._task, _abi.RESIDUAL_RAW)
        return s / xs.shape[0]

    def params_update(self, xs):
        import torch
        if self.fused_param_grad:
            # mean |gradV . xdot + running_cost| over the minibatch (cell 11) = sum / B: the flat buffer's hjb part divided by its count of
            # interior samples (all of them: done = 0) -- hjbx_mix_adam_f32 with regularisation 0; its eps only guards 0 / 0 of the unused
            # termination half
            from .. import _ops
            from .vhjb import adam_state
            vf = self.value_function_approximator
            params = list(vf.parameters())
            B = xs.shape[0]
            if self._zeros.shape[0] < B:
                self._zeros = self._zeros.new_zeros(B)
            z = self._zeros[:B]
            if self._ones is None or self._ones.shape[0] < B:
                self._ones = torch.ones(B, dtype=self.dtype, device=self.device)
            # (costs = 1: the unused termination residual |V / (cost + eps) - 1| done must stay finite with this task's eps = 0)
            flat = _ops.value_loss_grad(self.dynamics.system, self._task, vf.descriptor(), xs, self._ones[:B], z, _abi.RESIDUAL_RAW)
            m, v, steps = adam_state(self.optimizer, params)
            g = self.optimizer.param_groups[0]
            losses = _ops.mix_adam(flat, 0.0, 1e-30, [p.data for p in params], m, v, steps, self._adam_ticket, g["lr"], g["betas"][0], g["betas"][1], g["eps"])
            return losses[1]
        # differentiate w.r.t. fresh leaves aliasing the parameters (see VHJBController._update_core): a hipGraph capture of
        # this step must not meet grad accumulators that an older, still-alive autograd graph created on another stream
        params = list(self.value_function_approximator.parameters())
        leaves = [p.detach().requires_grad_(True) for p in params]
        loss = self.hjb_loss(xs, weights=leaves)
        for p, g in zip(params, torch.autograd.grad(loss, leaves)):
            p.grad = g
        self.optimizer.step()
        return loss.detach()

    def params_update_graphed(self, xs):
        """`params_update` replayed from a hipGraph (captured for the full minibatch shape; a ragged last batch runs eagerly).
        The returned loss is the graph's static output: consume it before the next call."""
        from .vhjb import GraphedStep
        if xs.shape[0] != self.batch_size:
            return self.params_update(xs)
        if self._graphed is None:
            self._graphed = GraphedStep(self.optimizer, list(self.value_function_approximator.parameters()), self.params_update, (xs,))
        return self._graphed(xs)

    def train(self, epochs: int = 100, evaluate: bool = True, verbose: bool = False):
        """-> (losses per epoch, mean time-to-target per epoch, std per epoch), as cell 11 collects them."""
        import torch
        losses, means, stds = [], [], []
        N = self.states.shape[0]
        for epoch in range(epochs):
            if evaluate:
                mu, sd = self.get_mean_and_std_of_policy()
                means.append(mu)
                stds.append(sd)
            perm = torch.randperm(N, generator=self._gen, device=self.device)
            total, nb = torch.zeros((), dtype=self.dtype, device=self.device), 0
            for i in range(0, N, self.batch_size):
                update = self.params_update_graphed if self.graph_updates else self.params_update
                total += update(self.states[perm[i:i + self.batch_size]].contiguous())
                nb += 1
            losses.append(float(total / max(nb, 1)))
            if verbose and (epoch + 1) % 10 == 0:
                print(f"epoch:{epoch + 1}, loss:{losses[-1]}, time to origin:{means[-1] if means else float('nan')}")
        return losses, means, stds
