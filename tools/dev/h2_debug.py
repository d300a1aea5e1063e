import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_dynamics, make_vhjb_config
from q_learning_with_hjb_amd import _abi, _ops
from q_learning_with_hjb_amd.controller.vhjb import VHJBController
name = sys.argv[1] if len(sys.argv) > 1 else "quad2d"
d = make_dynamics(name)
ctl = VHJBController(d, make_vhjb_config(name), dtype=torch.float32)
vf = ctl.value_function_approximator
rng = np.random.default_rng(3)
B = 4096
box = np.asarray(ctl.obs_max, np.float64).clip(max=3.0)
x = torch.as_tensor(np.asarray(ctl.xf, np.float64) + rng.uniform(-1, 1, (B, d.state_dim)) * box, dtype=torch.float32, device="cuda").contiguous()
res = {}
for a in (0, 1, 2):
    _abi.set_option(_abi.OPT_MLP_ARITHMETIC, a)
    V, g = _ops.value_grad(d.system, vf.descriptor(), x)
    res[a] = (V.cpu().numpy(), g.cpu().numpy())
_abi.set_option(_abi.OPT_MLP_ARITHMETIC, 0)
np.set_printoptions(linewidth=200, precision=5)
print("W abs max", [float(w.abs().max()) for w in vf.weights], "W abs min nonzero", [float(w.abs()[w.abs() > 0].min()) for w in vf.weights])
print("g f32   ", res[0][1][:4])
print("g bf16x3", res[1][1][:4])
print("g f16x2 ", res[2][1][:4])
err = np.abs(res[2][1] - res[0][1]); sc = np.abs(res[0][1]).max(0)
print("per-component max err / max |g|:", err.max(0) / sc)
print("ratio g2/g0 first rows", res[2][1][:4] / res[0][1][:4])
rel = (err / (np.abs(res[0][1]) + 1e-3 * sc)).max(1)
print("per-env rel err: median %.2e  p10 %.2e  p90 %.2e  max %.2e; fraction < 1e-5: %.3f" % (np.median(rel), np.quantile(rel, .1), np.quantile(rel, .9), rel.max(), (rel < 1e-5).mean()))
print("first 64 envs rel err:", np.array2string(rel[:64], precision=1))
print("V rel err f16x2 vs f32 (first 8):", np.abs(res[2][0][:8] - res[0][0][:8]) / np.abs(res[0][0][:8]))
# same network, states scaled towards the target
for scale in (0.3, 0.03):
    xs = torch.as_tensor(np.asarray(ctl.xf, np.float64) + rng.uniform(-1, 1, (B, d.state_dim)) * box * scale, dtype=torch.float32, device="cuda").contiguous()
    out = {}
    for a in (0, 2):
        _abi.set_option(_abi.OPT_MLP_ARITHMETIC, a)
        out[a] = _ops.value_grad(d.system, vf.descriptor(), xs)[1].cpu().numpy()
    _abi.set_option(_abi.OPT_MLP_ARITHMETIC, 0)
    e2 = np.abs(out[2] - out[0]).max(1) / (np.abs(out[0]).max(1) + 1e-30)
    print(f"states x{scale}: rel err median {np.median(e2):.2e} max {e2.max():.2e}")
