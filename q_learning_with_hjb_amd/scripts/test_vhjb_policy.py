"""Train-then-evaluate entry point, the counterpart of the reference's scripts/test_vhjb_policy.py
(load_systems :20-130, test_policy :132-154, main :227-240) without the matplotlib / pdb parts.

    python -m q_learning_with_hjb_amd.scripts.test_vhjb_policy --env_name cartpole [--dynamics_config f.gin]
           [--vhjb_controller_config g.gin] [--epochs N] [--eval_batch B] [--T 5]

`test_policy` runs the learned policy and the model-based controller in lock-step from the SAME initial
states (a batch of them instead of one) and returns trajectories, controls and per-step costs l(x,u)*dt."""
from __future__ import annotations

import argparse
import json

import numpy as np
import torch

from .. import _abi, _ops
from ..configs import defaults, gin_lite
from ..configs.controller.vhjb_controller_config import VHJBControllerConfig
from ..configs.dynamics.dynamics_config import (AcrobotDynamicsConfig, CartpoleDynamicsConfig, LinearDynamicsConfig,
                                                NearHoverQuadcopterConfig, Quadrotors2DConfig)
from ..controller.acrobot_energy_shaping import AcrobotEnergyShapingController
from ..dynamics.acrobot import Acrobot
from ..controller.cartpole_energy_shaping import CartpoleEnergyShapingController
from ..controller.lqr import LQR
from ..controller.quadrotors_model_based_controller import (NearHoverQuadcopterHoveringController,
                                                            Quadrotors2DHoveringController)
from ..controller.vhjb import VHJBController
from ..dynamics.cartpole import Cartpole
from ..dynamics.linear import LinearDynamics
from ..dynamics.quadrotors import NearHoverQuadcopter, Quadrotors2D

_ENVS = {
    # name: (dynamics config class, stock dynamics config, dynamics class, stock controller config)
    "lqr": (LinearDynamicsConfig, defaults.linear_dynamics_config, LinearDynamics, defaults.linear_vhjb_config),
    "cartpole": (CartpoleDynamicsConfig, defaults.cartpole_dynamics_config, Cartpole, defaults.cartpole_vhjb_config),
    "acrobot": (AcrobotDynamicsConfig, defaults.acrobot_dynamics_config, Acrobot, defaults.acrobot_vhjb_config),  # no upstream config
    "quadrotors2DHovering": (Quadrotors2DConfig, defaults.quadrotors2d_dynamics_config, Quadrotors2D, defaults.quadrotors2d_vhjb_config),
    "nearHoverQuadcopter": (NearHoverQuadcopterConfig, defaults.near_hover_dynamics_config, NearHoverQuadcopter,
                            defaults.near_hover_vhjb_config),   # notebook-only upstream (examples/10D_quadcopte.ipynb)
}


def load_systems(env_name, dynamics_config=None, vhjb_controller_config=None, activation="relu", **controller_overrides):
    """-> (dynamics, nn_policy, model_based_policy); config files are gin files of the reference's format."""
    cfg_cls, stock_dyn, dyn_cls, stock_ctl = _ENVS[env_name]
    if dynamics_config is None:
        dcfg = stock_dyn()
    else:
        gin_lite.parse_config_file(dynamics_config)
        dcfg = cfg_cls()
    if vhjb_controller_config is None:
        ccfg = stock_ctl(**controller_overrides)
    else:
        gin_lite.parse_config_file(vhjb_controller_config)
        ccfg = VHJBControllerConfig(**controller_overrides)
    dynamics = dyn_cls(dcfg)
    nn_policy = VHJBController(dynamics, ccfg, activation=activation)
    Q, R = np.asarray(ccfg.Q, np.float64), np.asarray(ccfg.R, np.float64)
    if env_name == "lqr":
        model_based = LQR(dynamics, Q, R)
    elif env_name == "cartpole":
        model_based = CartpoleEnergyShapingController(dynamics, Q, R)
    elif env_name == "acrobot":
        model_based = AcrobotEnergyShapingController(dynamics, Q, R)
    elif env_name == "quadrotors2DHovering":
        model_based = Quadrotors2DHoveringController(dynamics, np.asarray(ccfg.xf, np.float64), Q, R)
    else:
        model_based = NearHoverQuadcopterHoveringController(dynamics, np.asarray(ccfg.xf, np.float64), Q, R)
    return dynamics, nn_policy, model_based


@torch.no_grad()
def test_policy(nn_policy: VHJBController, dynamics, model_based_controller, T: float = 5, batch: int = 1, x0=None):
    """Lock-step closed loops of the learned and the model-based policy (reference :132-154), batched.

    Returns a dict of numpy arrays: t_span (S,), xs_learned / xs_model_based (S, B, n), us_* (S-1, B, m),
    cost_learned / cost_model_based (S-1, B) [= running_cost * dt per step]."""
    nn_policy.train_mode = False
    t_span = np.arange(0, T, dynamics.dt)
    steps = t_span.shape[0] - 1
    if x0 is None:
        x0 = dynamics.get_initial_state(batch_size=batch)
    x0 = nn_policy._dev(np.atleast_2d(x0))
    B, n = x0.shape
    sysh, integ = dynamics.system, dynamics.integrator
    # the evaluation loop never terminates an environment: same task, unbounded observation box
    task = _abi.make_task(n, nn_policy.control_dim, nn_policy.Q, nn_policy.R, nn_policy.P, nn_policy.xf, nn_policy.uf, None, None,
                          nn_policy.epsilon, Rinv=nn_policy.R_inv)
    xs = torch.empty((steps + 2, B, n), dtype=x0.dtype, device=x0.device)
    us = torch.empty((max(steps, 1), B, nn_policy.control_dim), dtype=x0.dtype, device=x0.device)
    cost = torch.empty((steps + 1, B), dtype=x0.dtype, device=x0.device)
    done = torch.empty_like(cost)
    done_step = torch.full((B,), -1, dtype=torch.int32, device=x0.device)
    xs[0].copy_(x0)
    for t in range(steps):
        g = nn_policy.get_v_gradient(xs[t])
        _ops.vhjb_step(sysh, task, t, 1 << 30, xs[t], g, xs[t + 1], cost[t], done[t], done_step, u_out=us[t], integrator=integ)
    mb = model_based_controller.rollout(x0, steps, task=task, terminate=False, log_traj=True, log_u=True, log_cost=True)
    return dict(t_span=t_span, xs_learned=xs[:steps + 1].cpu().numpy(), us_learned=us[:steps].cpu().numpy(),
                cost_learned=cost[:steps].cpu().numpy(), xs_model_based=mb["traj"].cpu().numpy(), us_model_based=mb["u"].cpu().numpy(),
                cost_model_based=mb["cost"][:steps].cpu().numpy())


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--env_name", default="lqr", choices=sorted(_ENVS), help="Environment name")
    parser.add_argument("--dynamics_config", help="The path to the dynamics config")
    parser.add_argument("--vhjb_controller_config", help="The path to the config of vhjb controller")
    parser.add_argument("--epochs", type=int, default=None, help="override VHJBControllerConfig.epochs")
    parser.add_argument("--eval_batch", type=int, default=10, help="number of evaluation start states")
    parser.add_argument("--T", type=float, default=5.0)
    parser.add_argument("--warm_start", type=int, default=0, help="seed the replay buffer with this many closed loops of the "
                        "model-based controller before training (BASELINE configs[2]: energy-shaping warm-start + vhjb)")
    args = parser.parse_args(argv)
    over = {} if args.epochs is None else {"epochs": args.epochs}
    dynamics, nn_policy, model_based_policy = load_systems(args.env_name, args.dynamics_config, args.vhjb_controller_config, **over)
    if args.warm_start > 0:
        nn_policy.warm_start(model_based_policy, args.warm_start)
    lists = nn_policy.train()
    res = test_policy(nn_policy, dynamics, model_based_policy, T=args.T, batch=args.eval_batch)
    summary = dict(env=args.env_name, epochs=nn_policy.epochs,
                   final_average_trajectory_cost=lists[0][-1] if lists[0] else None,
                   final_average_trajectory_length=lists[2][-1] if lists[2] else None,
                   final_hjb_loss=lists[4][-1] if lists[4] else None,
                   mean_cost_learned=float(res["cost_learned"].sum(0).mean()),
                   mean_cost_model_based=float(res["cost_model_based"].sum(0).mean()))
    print(json.dumps(summary))
    return lists, res


if __name__ == "__main__":
    main()
