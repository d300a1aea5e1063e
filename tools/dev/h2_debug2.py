import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_dynamics, make_vhjb_config
from q_learning_with_hjb_amd import _abi, _ops
from q_learning_with_hjb_amd.controller.vhjb import VHJBController
for name in ("cartpole", "quad2d"):
    d = make_dynamics(name)
    ctl = VHJBController(d, make_vhjb_config(name), dtype=torch.float32)
    vf = ctl.value_function_approximator
    rng = np.random.default_rng(3)
    B = 4096
    box = np.asarray(ctl.obs_max, np.float64).clip(max=3.0)
    x = torch.as_tensor(np.asarray(ctl.xf, np.float64) + rng.uniform(-1, 1, (B, d.state_dim)) * box, dtype=torch.float32, device="cuda").contiguous()
    W0 = [w.detach().clone() for w in vf.weights]
    for f2, f3 in ((1, 1), (0.5, 1), (2, 1), (1, 0.5), (1, 2), (4, 4), (0.25, 0.25)):
        with torch.no_grad():
            vf.weights[1].copy_(W0[1] * f2); vf.weights[2].copy_(W0[2] * f3)
        out = {}
        for a in (0, 2):
            _abi.set_option(_abi.OPT_MLP_ARITHMETIC, a)
            out[a] = [t.cpu().numpy() for t in _ops.value_grad(d.system, vf.descriptor(), x)]
        _abi.set_option(_abi.OPT_MLP_ARITHMETIC, 0)
        eg = np.abs(out[2][1] - out[0][1]).max(1) / (np.abs(out[0][1]).max(1) + 1e-30)
        eV = np.abs(out[2][0] - out[0][0]) / (np.abs(out[0][0]) + 1e-30)
        print(f"{name}: W2 x{f2} (max {float(vf.weights[1].abs().max()):.4f}), W3 x{f3} (max {float(vf.weights[2].abs().max()):.4f}): V rel err median {np.median(eV):.1e}, grad rel err median {np.median(eg):.1e} max {eg.max():.1e}", flush=True)
