"""Development aid: errors of the fused tanh / sin value-gradient kernel and of the oracle's float build against the f64 oracle, relative to
per-element term scales (activation-agnostic: |W| sums with |h| <= 1 replaced by 1), to decide what the parity test of the smooth
activations can assert.   python tools/dev/act_yardstick.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
from test_gpu_vhjb import controller, oracle_mlp, states_near_target  # noqa: E402


def scales(W, std, eps, e, y, s1, s2):
    W1, W2, W3 = (np.abs(w) for w in W)
    ones1 = np.ones((e.shape[0], W1.shape[1]))
    t_y = np.ones((e.shape[0], W2.shape[1])) @ W3                     # terms of y with |h2| -> 1
    sV = (2 * np.abs(y) * t_y).sum(1) + (y * y).sum(1) + eps * (e * e).sum(1)
    G = (((2 * t_y) @ W3.T) @ W2.T) @ W1.T / np.abs(std) + 2 * eps * np.abs(e)   # |act'| <= 1
    return sV, G


for act in ("tanh", "sin"):
    for name in ("linear", "cartpole", "quad2d", "nearhover"):
        d, ctl = controller(name, torch.float32, activation=act)
        vf = ctl.value_function_approximator
        with torch.no_grad():
            for w in vf.weights:
                w.mul_(1.7)
        x = states_near_target(d, ctl, 20000, 2, 1.5)
        V, g = vf.fused_value_grad(x)
        mlp, W = oracle_mlp(ctl)
        s = O.System.from_dynamics(d)
        xn = x.cpu().numpy().astype(np.float64)
        oV, og = O.value_grad(s, mlp, *W, xn)
        cV, cg = O.value_grad(s, mlp, *W, xn, dtype=np.float32)
        e = O.wrap(s, xn - np.asarray(vf._np["xf"], np.float64)[None, :])
        z = (e - vf._np["mean"]) / vf._np["std"]
        f = {"tanh": np.tanh, "sin": np.sin}[act]
        h1 = f(z @ W[0]); h2 = f(h1 @ W[1]); y = h2 @ W[2]
        sV, G = scales(W, np.asarray(vf._np["std"], np.float64)[None, :], vf.epsilon_scalar, e, y, None, None)
        kv, cv = np.abs(V.cpu().numpy() - oV) / sV, np.abs(np.asarray(cV, np.float64) - oV) / sV
        kg, cg_ = np.abs(g.cpu().numpy() - og) / G, np.abs(np.asarray(cg, np.float64) - og) / G
        q = lambda a: (float(np.max(a)), float(np.quantile(a, 0.999)), float(np.median(a)))
        print(act, name, "V kernel max/p999/med %.2e %.2e %.2e | cpu32 %.2e %.2e %.2e" % (q(kv) + q(cv)),
              " g kernel %.2e %.2e %.2e | cpu32 %.2e %.2e %.2e" % (q(kg) + q(cg_)), flush=True)
