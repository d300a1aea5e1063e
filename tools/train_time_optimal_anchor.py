#!/usr/bin/env python3
"""End-to-end anchor for the time-optimal variant: the "ours" run of the reference's
examples/double_integrator_optimal_time.ipynb (cells 5-11: sin network, 2^16 states, batch 256, Adam 1e-3, 100 epochs) on the
device, then time-to-origin of the learned bang-bang law vs the analytic law and the saturated LQR from the same random starts
(cell 21).  The notebook's recorded outputs (other RNG, Flax init): loss 0.072 (epoch 10) -> 0.026 (epoch 100), time to origin
during learning 4.25 s -> 2.7-3.3 s; final means over 10 starts: learned 2.615 s, analytic 1.572 s, LQR 4.104 s.
    python tools/train_time_optimal_anchor.py --epochs 100"""
import argparse
import json
import os
import sys
import time

import numpy as np
import scipy.linalg
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from q_learning_with_hjb_amd import _abi  # noqa: E402
from q_learning_with_hjb_amd.configs.defaults import linear_dynamics_config  # noqa: E402
from q_learning_with_hjb_amd.controller.lqr import LQR  # noqa: E402
from q_learning_with_hjb_amd.controller.time_optimal import DoubleIntegratorTimeOptimalController, TimeOptimalVHJBController  # noqa: E402
from q_learning_with_hjb_amd.dynamics.linear import LinearDynamics  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--activation", default="sin")
    ap.add_argument("--starts", type=int, default=1000)
    args = ap.parse_args()
    d = LinearDynamics(linear_dynamics_config(dt=0.01, umin=[-1.0], umax=[1.0]))
    d.integrator = _abi.ZOH
    ctl = TimeOptimalVHJBController(d, activation=args.activation, seed=0)
    t0 = time.time()
    losses, means, stds = ctl.train(epochs=args.epochs, verbose=True)
    train_s = time.time() - t0
    x0 = (torch.rand((args.starts, 2), device="cuda", generator=torch.Generator(device="cuda").manual_seed(123)) * 2 - 1).contiguous()
    t_learned = ctl.time_to_target(x0, 15.0)
    t_analytic = DoubleIntegratorTimeOptimalController(d).time_to_target(x0, 15.0)
    A, Bm = np.array([[0.0, 1], [0, 0]]), np.array([[0.0], [1]])
    P = scipy.linalg.solve_continuous_are(A, Bm, np.eye(2), np.array([[0.01]]))      # notebook cell 2: R = 0.01
    K = (Bm.T @ P / 0.01)
    lqr = _abi.make_controller(_abi.CTRL_LINEAR_FEEDBACK, 2, 1, K, wrap_error=False, eps_region=1e-4)
    from q_learning_with_hjb_amd import _ops
    out = _ops.rollout_feedback(d.system, lqr, x0, 1500, integrator=_abi.ZOH, stop_at_target=True, log_traj=False, log_u=False)
    t_lqr = out["done_step"].to(torch.float32) * d.dt
    pick = [9, 19, 29, 39, 49, 59, 69, 79, 89, 99]
    print(json.dumps(dict(epochs=args.epochs, activation=args.activation, updates=args.epochs * ((ctl.states.shape[0] + 255) // 256),
                          train_seconds=round(train_s, 1),
                          loss_at_epochs_10_to_100=[round(losses[i], 5) for i in pick if i < len(losses)],
                          time_to_origin_during_learning=[round(means[i], 3) for i in pick if i < len(means)],
                          starts=args.starts, mean_time_learned=round(float(t_learned.mean()), 3), std_time_learned=round(float(t_learned.std()), 3),
                          reached_learned=round(float((t_learned < 15.0).float().mean()), 4),
                          mean_time_analytic=round(float(t_analytic.mean()), 3), mean_time_lqr=round(float(t_lqr.mean()), 3))))


if __name__ == "__main__":
    main()
