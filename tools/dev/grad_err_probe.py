"""Development aid: where the float32 parameter-gradient kernels differ from PyTorch's float32 autograd (both against float64 autograd)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from q_learning_with_hjb_amd import _abi, _ops  # noqa: E402
from test_gpu_train import _batch, _reference_sums_f64, _unpack  # noqa: E402
from test_gpu_vhjb import controller  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cartpole"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 33
d, ctl = controller(name)
vf = ctl.value_function_approximator
with torch.no_grad():
    for w in vf.weights:
        w.mul_(1.3)
xs, dones, costs = _batch(d, ctl, B, 31)
rh, rt, rsc = _reference_sums_f64(name, ctl, xs, dones, costs)
th, tt, tsc = _reference_sums_f64(name, ctl, xs, dones, costs, dtype=torch.float32)
for label, kern, arith in (("coop/f32", 0, 0), ("pair/f32", 1, 0)):
    _abi.set_option(_abi.OPT_TRAIN_KERNEL, kern)
    flat = _ops.value_loss_grad(d.system, ctl._task, vf.descriptor(), xs, costs, dones)
    gh, gt, sc = _unpack(flat, d.state_dim)
    print(label, "loss sums kernel", sc[:2], "f64", rsc[:2], "torch32", tsc[:2])
    for k in range(3):
        b = rh[k].cpu().numpy().astype(np.float64)
        c = th[k].cpu().numpy().astype(np.float64)
        print(f"   hjb dW{k+1}: |b| {np.linalg.norm(b):.3e}  kernel err {np.linalg.norm(gh[k]-b):.3e}  torch32 err {np.linalg.norm(c-b):.3e}")
# the input gradient itself: fused inference kernel vs torch float32 vs float64
with torch.no_grad():
    V32, g32 = vf.value_and_grad(xs)
Vk, gk = vf.fused_value_grad(xs)
d64, ctl64 = controller(name, torch.float64)
with torch.no_grad():
    for w64, w32 in zip(ctl64.value_function_approximator.weights, vf.weights):
        w64.copy_(w32.double())
    V64, g64 = ctl64.value_function_approximator.value_and_grad(xs.double())
print("gradV: |g| %.3e  kernel err %.3e  torch32 err %.3e" % (float(g64.norm()), float((gk.double() - g64).norm()), float((g32.double() - g64).norm())))
# residual gradient dl/dg for the three g's
for tag, g in (("f64 g", g64.float()), ("kernel g", gk), ("torch32 g", g32)):
    _, dg, sums = _ops.hjb_residual(d.system, ctl._task, xs, g.contiguous(), dones, _abi.RESIDUAL_NORMALISED, want_loss=False)
    _, dg64, sums64 = _ops.hjb_residual(d64.system, ctl64._task, xs.double(), g64.contiguous(), dones.double(), _abi.RESIDUAL_NORMALISED, want_loss=False)
    print(f"   residual kernel (float32) fed {tag}: dl/dg err vs f64 {float((dg.double() - dg64).norm()):.3e} of {float(dg64.norm()):.3e}; loss sum {float(sums[0]):.7f} vs {float(sums64[0]):.7f}")
# the oracle's float build (the reference's statements in float, sequential sums on the CPU) as a fourth source of g
from oracle import oracle as O  # noqa: E402
from test_gpu_vhjb import oracle_mlp  # noqa: E402
mlp, W = oracle_mlp(ctl)
s = O.System.from_dynamics(d)
cV, cg = O.value_grad(s, mlp, *W, xs.cpu().numpy().astype(np.float64), dtype=np.float32)
gc = torch.as_tensor(np.asarray(cg, np.float32), device="cuda")
print("gradV: CPU float build err %.3e" % float((gc.double() - g64).norm()))
_, dg, sums = _ops.hjb_residual(d.system, ctl._task, xs, gc.contiguous(), dones, _abi.RESIDUAL_NORMALISED, want_loss=False)
print(f"   residual kernel (float32) fed CPU-float g: dl/dg err vs f64 {float((dg.double() - dg64).norm()):.3e}; loss sum {float(sums[0]):.7f} vs {float(sums64[0]):.7f}")
for tag, g in (("kernel", gk), ("torch32", g32), ("cpu32", gc)):
    dgx = (g.double() - g64)
    rel = (dgx * g64).sum() / (g64 * g64).sum()
    print(f"   {tag}: projection of the error of g on g itself (relative bias) {float(rel):.3e}; mean signed relative error of the largest components {float((dgx / g64)[g64.abs() > 0.1 * g64.abs().max()].mean()):.3e}")
