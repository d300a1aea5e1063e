import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from q_learning_with_hjb_amd import _abi, _ops
from test_gpu_vhjb import controller, states_near_target
from test_gpu_train import _batch

TILE = 37888
def arr(tile, a, nfeat):
    g0 = a * 32 if a < 8 else 256 + (a - 8) * 16
    t = tile[g0 * 128:(g0 + nfeat // 4) * 128].reshape(nfeat // 32, 4, 2, 32, 4)      # fb, q, hh, env, c
    return t.permute(3, 0, 1, 2, 4).reshape(32, nfeat)                               # env, feature = 32fb + 8q + 4hh + c

for name, B in (("cartpole", 20), ("cartpole", 32)):
    d, ctl = controller(name)
    vf = ctl.value_function_approximator
    xs, dones, costs = _batch(d, ctl, B, 31)
    flat = _ops.value_loss_grad(d.system, ctl._task, vf.descriptor(), xs, costs, dones)
    torch.cuda.synchronize()
    ws = list(_ops._tws.values())[0]
    tile = ws[:TILE * 4].view(torch.float32).clone()
    n = d.state_dim
    small = tile[36864:].reshape(32, 32)
    W1, W2, W3 = [w.detach() for w in vf.weights]
    e = vf.error_coords(xs); z = (e - vf.mean) / vf.std
    a1 = z @ W1; h1 = torch.relu(a1); a2 = h1 @ W2; h2 = torch.relu(a2); y = h2 @ W3
    dy = 2 * y; d2 = (dy @ W3.t()) * (a2 > 0); d1 = (d2 @ W2.t()) * (a1 > 0); g = (d1 @ W1.t()) / vf.std + 2 * vf.epsilon_scalar * e
    V, gf = vf.fused_value_grad(xs)
    def cmp(label, got, want):
        print(f"  {label:6s} max|want| {float(want.abs().max()):.4e} max err {float((got[:B] - want).abs().max()):.3e}")
    print(name, B)
    cmp("z", small[:, :n], z); cmp("g", small[:, 2 * n + 1:3 * n + 1], g); cmp("g(inf)", gf, g)
    cmp("h1", arr(tile, 0, 128), h1); cmp("h2", arr(tile, 4, 128), h2); cmp("dy", arr(tile, 8, 64), dy)
    cmp("d2", arr(tile, 2, 128), d2); cmp("d1", arr(tile, 6, 128), d1)
    li, dg, sums = _ops.hjb_residual(d.system, ctl._task, xs, g.contiguous(), dones)
    gzb = dg / vf.std
    cmp("gzb", small[:, n:2 * n], gzb)
    t1 = (gzb @ W1) * (a1 > 0); cmp("dh1b", arr(tile, 1, 128), t1)
    t2 = (t1 @ W2) * (a2 > 0); cmp("dh2b", arr(tile, 5, 128), t2)
    yb = 2 * (t2 @ W3); cmp("yb", arr(tile, 9, 64), yb)
    t4 = (yb @ W3.t()) * (a2 > 0); cmp("a2b", arr(tile, 3, 128), t4)
    t5 = (t4 @ W2.t()) * (a1 > 0); cmp("a1b", arr(tile, 7, 128), t5)
    gW2 = t1.t() @ d2 + h1.t() @ t4
    P1 = n * 128
    got = flat[P1:P1 + 128 * 128].reshape(128, 128)
    print("  dW2_h max|want|", float(gW2.abs().max()), "err", float((got - gW2).abs().max()), "err vs transpose", float((got - gW2.t()).abs().max()))
