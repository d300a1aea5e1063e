#!/usr/bin/env python3
"""End-to-end anchor: train the VHJB controller with the stock configuration and compare closed-loop costs with the
model-based baseline from the same start states, the way the reference's notebooks report them (SURVEY section 6:
cartpole VHJB 9.1409 vs LQR 9.1410 over 10 s from 10 starts; quadrotor 9.389 vs 9.984).  Not bit-comparable (different
RNG, Flax init) -- a does-learning-work check.   python tools/train_anchor.py --env cartpole --epochs 100"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from q_learning_with_hjb_amd.scripts.test_vhjb_policy import load_systems, test_policy  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="cartpole")
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--trajectories", type=int, default=20, help="rollouts per epoch (reference: 20)")
    ap.add_argument("--T", type=float, default=10.0)
    ap.add_argument("--starts", type=int, default=10)
    ap.add_argument("--seed", type=int, default=0, help="VHJBControllerConfig.seed (reference: 0)")
    ap.add_argument("--activation", default="relu", choices=["relu", "tanh", "sin"], help="relu = controller/vhjb.py; tanh = the notebooks")
    ap.add_argument("--notebook", action="store_true", help="the notebooks' recipe (examples/cartpole_balancing.ipynb cell 10, "
                    "drone_hovering.ipynb cell 10): data set seeded with 256 copies of xf, no boundary set, no termination loss")
    ap.add_argument("--warm_start", type=int, default=0, help="seed the replay buffer with this many closed loops of the model-based "
                    "controller first (BASELINE configs[2]: acrobot energy-shaping warm-start + vhjb)")
    ap.add_argument("--arithmetic", default=None, choices=["f32", "bf16x3", "f16x2"], help="value-network arithmetic of the fused kernels "
                    "(HJBX_OPT_MLP_ARITHMETIC; default: the library's = f32)")
    args = ap.parse_args()
    if args.arithmetic:
        from q_learning_with_hjb_amd import _abi
        _abi.set_option(_abi.OPT_MLP_ARITHMETIC, {"f32": 0, "bf16x3": 1, "f16x2": 2}[args.arithmetic])
    over = {}
    if args.notebook:
        over = dict(num_of_interior_data=256, num_of_boundary_data=0, regularization_peak_value=0.0, regularization_init_value=0.0,
                    regularization_end_value=0.0)
        if args.env == "quadrotors2DHovering":      # examples/drone_hovering.ipynb cell 4 uses a wider position box than the gin file
            over.update(obs_min=[-3, -3, -1.5, -5, -5, -2], obs_max=[3, 3, 1.5, 5, 5, 2])
    dyn, pol, mb = load_systems(args.env, epochs=args.epochs, num_of_trajectories_per_epoch=args.trajectories, seed=args.seed,
                                activation=args.activation, **over)
    if args.notebook:      # 256 copies of xf, not a box around it
        rb = pol.replay_buffer
        rb.x[:rb.size] = torch.as_tensor(np.asarray(pol.xf, np.float64), dtype=rb.x.dtype, device=rb.x.device)
    ws = pol.warm_start(mb, args.warm_start) if args.warm_start > 0 else None
    t0 = time.time()
    lists = pol.train()
    train_s = time.time() - t0
    np.random.seed(123)
    res = test_policy(pol, dyn, mb, T=args.T, batch=args.starts)
    cl, cm = res["cost_learned"].sum(0), res["cost_model_based"].sum(0)
    # the same trained network evaluated in every value-network arithmetic (same start states)
    from q_learning_with_hjb_amd import _abi as _A
    prev = _A.set_option(_A.OPT_MLP_ARITHMETIC, -1)
    cost_by_arithmetic = {}
    if args.activation == "relu":
        for nm, v in (("f32", 0), ("bf16x3", 1), ("f16x2", 2)):
            _A.set_option(_A.OPT_MLP_ARITHMETIC, v)
            np.random.seed(123)
            cost_by_arithmetic[nm] = float(test_policy(pol, dyn, mb, T=args.T, batch=args.starts)["cost_learned"].sum(0).mean())
        _A.set_option(_A.OPT_MLP_ARITHMETIC, prev)
    print(json.dumps(dict(env=args.env, seed=args.seed, arithmetic=args.arithmetic or "f32", activation=args.activation, fused_param_grad=bool(pol.fused_param_grad), device_driven_fit=bool(pol._fit_graph is not None), notebook=args.notebook, epochs=args.epochs,
                          warm_start=None if ws is None else dict(records=ws["records"], average_trajectory_cost=round(ws["average_trajectory_cost"], 2)), updates=pol.update_counter, train_seconds=round(train_s, 1),
                          replay_records=len(pol.replay_buffer), avg_traj_len_first=lists[2][0], avg_traj_len_last=lists[2][-1],
                          hjb_loss_first=lists[4][0] if lists[4] else None, hjb_loss_last=lists[4][-1] if lists[4] else None,
                          mean_cost_learned=float(cl.mean()), mean_cost_model_based=float(cm.mean()), mean_cost_learned_by_eval_arithmetic=cost_by_arithmetic,
                          per_start_learned=[round(float(v), 3) for v in cl], per_start_model_based=[round(float(v), 3) for v in cm])))


if __name__ == "__main__":
    main()
