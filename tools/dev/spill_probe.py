"""Development probe: fused rollout vs step-by-step launches for the near-hover system (which field / step / lane differs)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_vhjb import controller, states_near_target
from q_learning_with_hjb_amd import _abi, _ops
d, ctl = controller("nearhover", torch.float32)
ctl.value_function_approximator.load_quadratic(ctl.P, noise=0.05, generator=torch.Generator(device="cuda").manual_seed(3))
B, T = 1000, 12
x0 = states_near_target(d, ctl, B, 8, 1.03)
n, m = d.get_dimension()
vf = ctl.value_function_approximator
traj = torch.empty((T + 2, B, n), device="cuda"); cost = torch.empty((T + 1, B), device="cuda"); done = torch.empty_like(cost)
res = torch.empty_like(cost); ul = torch.empty((T + 1, B, m), device="cuda")
ds = torch.full((B,), -1, dtype=torch.int32, device="cuda")
traj[0].copy_(x0)
for t in range(T + 1):
    g = vf.fused_value_grad(traj[t], want_v=False)[1]
    _ops.vhjb_step(d.system, ctl._task, t, T, traj[t], g, traj[t + 1], cost[t], done[t], ds, u_out=ul[t], resid_t=res[t])
ds1 = torch.full((B,), -1, dtype=torch.int32, device="cuda")
one = _ops.vhjb_rollout(d.system, ctl._task, vf.descriptor(), x0, T + 1, T, ds1, log_u=True, log_residual=True, want_x_out=True)
torch.cuda.synchronize()
for name, a, b in (("traj", one["traj"], traj), ("cost", one["cost"], cost), ("done", one["done"], done), ("u", one["u"], ul), ("residual", one["residual"], res)):
    bad = (a != b)
    while bad.dim() > 2: bad = bad.any(-1)
    print(name, "mismatching (step, env) pairs:", int(bad.sum()), "first steps:", sorted(set(bad.nonzero()[:, 0].tolist()))[:6], "envs:", sorted(set(bad.nonzero()[:, 1].tolist()))[:12], "...", "n envs", len(set(bad.nonzero()[:, 1].tolist())))
print("done_step equal:", bool(torch.equal(ds1, ds)), "  mismatching:", int((ds1 != ds).sum()), ds1[:8].tolist(), ds[:8].tolist())
bad = (one["traj"] != traj).any(-1)
if bad.any():
    t, e = bad.nonzero()[0].tolist()
    print("first traj mismatch at step", t, "env", e, "fused", one["traj"][t, e].tolist(), "stepwise", traj[t, e].tolist())
