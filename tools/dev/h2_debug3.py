import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_dynamics, make_vhjb_config
from q_learning_with_hjb_amd import _abi, _ops
from q_learning_with_hjb_amd.controller.vhjb import VHJBController
name = "quad2d"
d = make_dynamics(name)
ctl = VHJBController(d, make_vhjb_config(name), dtype=torch.float32)
vf = ctl.value_function_approximator
rng = np.random.default_rng(3)
B = 64
box = np.asarray(ctl.obs_max, np.float64).clip(max=3.0)
x = torch.as_tensor(np.asarray(ctl.xf, np.float64) + rng.uniform(-1, 1, (B, d.state_dim)) * box * 0.3, dtype=torch.float32, device="cuda").contiguous()
_abi.set_option(_abi.OPT_MLP_ARITHMETIC, 2)
g2 = _ops.value_grad(d.system, vf.descriptor(), x)[1].cpu().numpy().astype(np.float64)
_abi.set_option(_abi.OPT_MLP_ARITHMETIC, 0)
g0 = _ops.value_grad(d.system, vf.descriptor(), x)[1].cpu().numpy().astype(np.float64)
W = [w.detach().double().cpu().numpy() for w in vf.weights]
e = x.double().cpu().numpy() - np.asarray(ctl.xf, np.float64)
z = (e - vf._np["mean"]) / vf._np["std"]
a1 = z @ W[0]; h1 = np.maximum(a1, 0); a2 = h1 @ W[1]; h2 = np.maximum(a2, 0); y = h2 @ W[2]
def grad(m1, m2):
    d2 = (2 * y) @ W[2].T * m2
    d1 = d2 @ W[1].T * m1
    return d1 @ W[0].T / vf._np["std"] + 2 * vf.epsilon_scalar * e
M1, M2 = (a1 > 0) * 1.0, (a2 > 0) * 1.0
variants = {"true": grad(M1, M2), "m1=1": grad(1, M2), "m2=1": grad(M1, 1), "both=1": grad(1, 1), "m1<->m2": grad(M2, M1)}
for k, v in variants.items():
    print(f"{k:8s}: |g_f32 - v| / |v| = {np.abs(g0 - v).max() / np.abs(v).max():.2e}   |g_f16x2 - v| / |v| = {np.abs(g2 - v).max() / np.abs(v).max():.2e}")
# which single hidden units explain the difference? least squares of (g2 - true) on per-unit contributions of layer 1 and layer 2
b = 0
d2 = (2 * y[b]) @ W[2].T            # (128,) unmasked
contrib2 = (d2[:, None] * W[1].T * M1[b][None, :]) @ W[0].T / vf._np["std"]     # contribution of each layer-2 unit if it were active: (128, n)
d1 = (d2 * M2[b]) @ W[1].T
contrib1 = d1[:, None] * W[0].T / vf._np["std"]                                   # (128, n) per layer-1 unit
diff = g2[b] - variants["true"][b]
for nm, C, M in (("layer-2 units", contrib2, M2[b]), ("layer-1 units", contrib1, M1[b])):
    coef, res, *_ = np.linalg.lstsq(C.T, diff, rcond=None)
    print(nm, "residual of explaining env 0's error by flipping units:", np.linalg.norm(C.T @ coef - diff) / np.linalg.norm(diff))
print("env 0: err", diff, " g", variants["true"][b])
print("number of active units layer1/layer2:", int(M1[b].sum()), int(M2[b].sum()), " min |a1|", np.abs(a1[b]).min(), " min |a2|", np.abs(a2[b]).min())
