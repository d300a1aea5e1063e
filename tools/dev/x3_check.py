#!/usr/bin/env python3
"""Development check of the bf16x3-split arithmetic (HJBX_OPT_MLP_ARITHMETIC = 1): value / gradient against the f64 oracle and the
exact-f32 kernel, then the rollout kernel's time in both arithmetics."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_dynamics, make_vhjb_config  # noqa: E402
from oracle import oracle as O  # noqa: E402
from q_learning_with_hjb_amd import _abi, _ops  # noqa: E402
from q_learning_with_hjb_amd.controller.vhjb import VHJBController  # noqa: E402


def main():
    names = sys.argv[1:] or ["cartpole", "nearhover", "acrobot", "quad2d"]
    for name in names:
        d = make_dynamics(name)
        ctl = VHJBController(d, make_vhjb_config(name), dtype=torch.float32)
        vf = ctl.value_function_approximator
        for init in ("random", "lqr"):
            if init == "lqr":
                vf.load_quadratic(ctl.P)
            B = 70000
            rng = np.random.default_rng(3)
            box = np.asarray(ctl.obs_max, np.float64).clip(max=3.0)
            x = torch.as_tensor(np.asarray(ctl.xf, np.float64) + rng.uniform(-1, 1, (B, d.state_dim)) * box, dtype=torch.float32, device="cuda").contiguous()
            W = [w.detach().double().cpu().numpy() for w in vf.weights]
            mlp = O.make_mlp(vf.features, vf._np["mean"], vf._np["std"], vf._np["xf"], vf.epsilon_scalar)
            oV, og = O.value_grad(O.System.from_dynamics(d), mlp, *W, x.cpu().numpy().astype(np.float64))
            res = {}
            for arith in (0, 1, 2):
                _abi.set_option(_abi.OPT_MLP_ARITHMETIC, arith)
                V, gr = _ops.value_grad(d.system, vf.descriptor(), x)
                torch.cuda.synchronize()
                res[arith] = (V.cpu().numpy().astype(np.float64), gr.cpu().numpy().astype(np.float64))
            _abi.set_option(_abi.OPT_MLP_ARITHMETIC, 0)
            sv, sg = np.abs(oV).max(), np.abs(og).max()
            for arith in (0, 1, 2):
                V, gr = res[arith]
                eV = np.abs(V - oV) / (np.abs(oV) + 1e-3 * sv)
                eg = np.abs(gr - og).max(1) / (np.abs(og).max(1) + 1e-3 * sg)
                print(f"{name:9s} {init:6s} arith={arith}: V rel err max {eV.max():.2e} median {np.median(eV):.2e} p99.9 {np.quantile(eV, 0.999):.2e} | "
                      f"grad rel err max {eg.max():.2e} median {np.median(eg):.2e} p99.9 {np.quantile(eg, 0.999):.2e}", flush=True)
        # rollout timing
        B, T = (1 << 18 if name == "quad2d" else 1 << 20), 40
        g = torch.Generator(device="cuda").manual_seed(1)
        x0 = (torch.as_tensor(np.asarray(ctl.xf), device="cuda", dtype=torch.float32) + (torch.rand(B, d.state_dim, generator=g, device="cuda") * 2 - 1) * 0.3).contiguous()
        for arith in (0, 1, 2):
            _abi.set_option(_abi.OPT_MLP_ARITHMETIC, arith)
            ts = []
            for rep in range(4):
                ds = torch.full((B,), -1, dtype=torch.int32, device="cuda")
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = _ops.vhjb_rollout(d.system, ctl._task, vf.descriptor(), x0, T, T, ds, log_traj=True, log_u=True, log_residual=True)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            live = float((ds < 0).float().mean())
            print(f"{name:9s} rollout arith={arith}: {min(ts) / T * 1e3:.4f} ms/step  ({B * T / min(ts):.3e} env-steps/s, still running at the end {live:.3f}), "
                  f"cost sum {float(out['cost'].double().sum()):.6e}", flush=True)
        _abi.set_option(_abi.OPT_MLP_ARITHMETIC, 0)


if __name__ == "__main__":
    main()
