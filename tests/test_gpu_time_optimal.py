"""GPU tests of the time-optimal variant (examples/double_integrator_optimal_time.ipynb cells 5-11 of the reference):
HJBX_LAW_BANGBANG in the pointwise / residual / step / fused rollout kernels against the CPU oracle, the analytic
known answers, and the learner built on them.  The notebook needs JAX and records no numbers: PARITY UNPINNED against
the notebook itself; the pins are the oracle restatement, the closed-form minimum-time function and the reference's
.mat ground truth (see tests/test_oracle_golden.py)."""
import numpy as np
import pytest
import torch

from conftest import ANGLE_IDX, SYSTEMS, make_dynamics, orc_system, wrapped_diff
from oracle import oracle as O
from q_learning_with_hjb_amd import _abi, _ops
from q_learning_with_hjb_amd.controller.time_optimal import DoubleIntegratorTimeOptimalController, TimeOptimalVHJBController
from test_gpu_parity import DT, check, dev, sample_states
from test_oracle_golden import _double_integrator, _min_time_value_and_grad

pytestmark = pytest.mark.gpu


def bb_task(d, r2=0.05, xf=None, box=None):
    n, m = d.get_dimension()
    xf = np.asarray(d.x0_mean, np.float64) if xf is None else xf
    return _abi.make_task(n, m, np.eye(n), np.eye(m), 0.5 * np.eye(n), xf, np.zeros(m), None if box is None else -box, box, 1e-10,
                          law=_abi.LAW_BANGBANG, target_r2=r2)


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("name", SYSTEMS)
def test_bang_bang_pointwise_and_residual_vs_oracle(name, prec):
    tdt, ndt, tol = DT[prec]
    d, x, u = sample_states(name, 1000, seed=11, spread=0.4)
    x[:40] = np.asarray(d.x0_mean, np.float64) + 1e-3                         # some states inside the target ball
    s = orc_system(name)
    task = bb_task(d)
    rng = np.random.default_rng(5)
    g = rng.standard_normal(x.shape) * 3
    g[40:60] = 0.0                                                            # sign(0) = 0 -> u = 0
    xd, gd, ud = dev(x, tdt), dev(g, tdt), dev(u, tdt)
    xr, gr, ur = (t.cpu().numpy().astype(np.float64) for t in (xd, gd, ud))
    uo = O.control_from_grad(s, task, xr, gr)
    ug = _ops.control_from_grad(d.system, task, xd, gd)
    check(ug, uo, tol, max_bad_frac=0.0 if prec == "f64" else 0.005)          # fp32 may flip a sign where f2'g ~ 0
    assert set(np.unique(uo)).issubset(set(np.concatenate([d.umin, d.umax, [0.0]]).astype(np.float64).tolist()))
    assert np.all(uo[40:60] == 0)
    check(_ops.running_cost(d.system, task, xd, ud), O.running_cost(s, task, xr, ur), tol)
    done = (rng.uniform(size=len(x)) < 0.2).astype(np.float64)
    dd = dev(done, tdt)
    for mode in (_abi.RESIDUAL_RAW, _abi.RESIDUAL_NORMALISED):
        li, dg, sums = _ops.hjb_residual(d.system, task, xd, gd, dd, mode)
        lo, dgo, so = O.hjb_residual(s, task, xr, gr, done, mode)
        same = (ug.cpu().numpy().astype(np.float64) == uo).all(1)
        sc = np.abs(lo).max() + 1
        check(li[torch.as_tensor(same, device="cuda")], lo[same], tol, sc)
        check(dg[torch.as_tensor(same, device="cuda")], dgo[same], tol, np.abs(dgo).max() + 1)
        if same.all():
            check(sums, so, tol * 10, np.abs(so).max())


def test_bang_bang_known_answer_on_device():
    """gradT of the analytic minimum-time function through the kernels: analytic control back, zero raw residual."""
    d = _double_integrator()
    rng = np.random.default_rng(2)
    x = rng.uniform(-1, 1, (4096, 2))
    keep = (np.abs(x[:, 0] + 0.5 * x[:, 1] * np.abs(x[:, 1])) > 1e-3) & ((x * x).sum(1) > 1e-4)
    _, grad, _ = _min_time_value_and_grad(x[:, 0], x[:, 1])
    task = _abi.make_task(2, 1, np.eye(2), np.eye(1), None, [0, 0], [0], None, None, 0.0, law=_abi.LAW_BANGBANG, target_r2=1e-4)
    xd, gd = dev(x, torch.float64), dev(grad, torch.float64)
    u = _ops.control_from_grad(d.system, task, xd, gd).cpu().numpy()
    c = DoubleIntegratorTimeOptimalController(d)
    ua = _ops.controller(d.system, c._descriptor(), xd).cpu().numpy()
    assert np.array_equal(u[keep], ua[keep])
    li, _, sums = _ops.hjb_residual(d.system, task, xd, gd, torch.zeros(4096, dtype=torch.float64, device="cuda"), _abi.RESIDUAL_RAW)
    assert float(li[torch.as_tensor(keep, device="cuda")].max()) < 1e-9 and float(sums[1]) == 4096


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("name", ["linear", "cartpole", "quad2d"])
def test_bang_bang_vhjb_step_vs_oracle(name, prec):
    """hjbx_vhjb_step with the bang-bang law: target-ball termination, unit running cost, terminal e'Pe."""
    tdt, ndt, tol = DT[prec]
    d, x, _ = sample_states(name, 777, seed=12, spread=0.5)
    x[:30] = np.asarray(d.x0_mean, np.float64) + 2e-3                         # inside the ball -> terminal at this step
    s = orc_system(name)
    n = d.state_dim
    task = bb_task(d, r2=0.02, box=np.full(n, 0.4))
    rng = np.random.default_rng(6)
    g = rng.standard_normal(x.shape) * 3
    xd, gd = dev(x, tdt), dev(g, tdt)
    xr, gr = xd.cpu().numpy().astype(np.float64), gd.cpu().numpy().astype(np.float64)
    ds0 = np.full(len(x), -1, np.int32); ds0[100:120] = 0                      # some already finished
    ds = torch.as_tensor(ds0, device="cuda")
    xn = torch.empty_like(xd); c = torch.empty(len(x), dtype=tdt, device="cuda"); dn = torch.empty_like(c); rs = torch.empty_like(c)
    uo = torch.empty((len(x), d.control_dim), dtype=tdt, device="cuda")
    _ops.vhjb_step(d.system, task, 3, 50, xd, gd, xn, c, dn, ds, u_out=uo, resid_t=rs)
    oxn, ou, oc, od, ods, ors = O.vhjb_step(s, task, 3, 50, xr, gr, ds0)
    assert np.array_equal(ds.cpu().numpy(), ods)
    assert (ods[:30] == 3).all() and (ods == 3).sum() > 30                     # ball + box terminations
    same = (uo.cpu().numpy().astype(np.float64) == ou).all(1)
    assert same.mean() > (0.999 if prec == "f64" else 0.99)
    m = torch.as_tensor(same, device="cuda")
    check(xn[m], oxn[same], tol, np.abs(xr).max(), angle_idx=ANGLE_IDX[name])
    check(c[m], oc[same], tol); check(dn, od, 0.0); check(rs[m], ors[same], tol, np.abs(ors).max() + 1)
    live = (ds0 < 0) & (ods < 0)
    assert np.allclose(oc[live], d.dt)                                         # unit running cost x dt


@pytest.mark.parametrize("activation", ["relu", "sin"])
def test_bang_bang_fused_rollout_equals_stepwise_and_oracle(activation):
    """The persistent MFMA rollout kernel with the bang-bang law (ReLU value net, and the notebook's own sin network): bit-identical to
    value_grad + vhjb_step step by step, done_step equal to the oracle's env-by-env loop on almost every environment."""
    d = _double_integrator()
    d.integrator = _abi.ZOH
    ctl = TimeOptimalVHJBController(d, activation=activation, num_states=1024, seed=3)
    assert ctl.fused
    vf = ctl.value_function_approximator
    B, T = 1000, 40
    x0 = torch.rand((B, 2), device="cuda") * 0.4 - 0.2
    x0[:10] = 0.001
    traj = torch.empty((T + 2, B, 2), device="cuda"); cost = torch.empty((T + 1, B), device="cuda"); done = torch.empty_like(cost)
    ds = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    traj[0].copy_(x0)
    for t in range(T + 1):
        g = vf.fused_value_grad(traj[t], want_v=False)[1]
        _ops.vhjb_step(d.system, ctl._task, t, T, traj[t], g, traj[t + 1], cost[t], done[t], ds, integrator=_abi.ZOH)
    ds1 = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    one = _ops.vhjb_rollout(d.system, ctl._task, vf.descriptor(), x0, T + 1, T, ds1, integrator=_abi.ZOH)
    assert torch.equal(ds1, ds) and torch.equal(one["traj"], traj) and torch.equal(one["cost"], cost) and torch.equal(one["done"], done)
    assert (ds[:10] == 0).all()
    W = [w.detach().cpu().numpy().astype(np.float64) for w in vf.weights]
    mlp = O.make_mlp(vf.features, vf._np["mean"], vf._np["std"], vf._np["xf"], vf.epsilon_scalar, activation=activation)
    ref = O.vhjb_rollout(O.System.from_dynamics(d), ctl._task, mlp, *W, x0.cpu().numpy().astype(np.float64), T, integrator=_abi.ZOH)
    assert (ds.cpu().numpy() == ref["done_step"]).mean() > 0.9              # bang-bang: one fp32 sign flip shifts an arrival
    # time_to_target wraps exactly this
    tt = ctl.time_to_target(x0, max_time=T * d.dt)
    assert torch.equal((tt / d.dt).round().to(torch.int32), ds)


def test_time_optimal_learner_trains_and_evaluates():
    """A short run of the notebook's training loop (sin network, 2^12 states): the HJB loss falls, the per-epoch
    time-to-origin statistics exist, and the learned law returns +-1 / 0 controls."""
    d = _double_integrator()
    d.integrator = _abi.ZOH
    ctl = TimeOptimalVHJBController(d, activation="sin", num_states=4096, seed=0)
    assert ctl.fused and ctl.fused_param_grad                # round 3: the sin network runs on the fused MFMA kernels (n <= 4 for the gradient)
    xs = ctl.states[:256].contiguous()
    # loss == mean |gradV . xdot + running cost| evaluated with plain torch
    with torch.no_grad():
        _, g = ctl.value_function_approximator.value_and_grad(xs)
        u = -torch.sign(g[:, 1:2])
        xdot = torch.cat([xs[:, 1:2], u], 1)
        want = ((g * xdot).sum(1) + ((xs * xs).sum(1) > 1e-4).to(xs.dtype)).abs().mean()
    got = ctl.hjb_loss(xs)
    assert abs(float(got) - float(want)) < 1e-5 * max(1.0, float(want))
    # gradient w.r.t. the weights == torch.autograd through the same expression (u is piecewise constant)
    params = list(ctl.value_function_approximator.parameters())
    ga = torch.autograd.grad(got, params)
    _, g2 = ctl.value_function_approximator.value_and_grad(xs)
    l2 = ((g2 * xdot).sum(1) + ((xs * xs).sum(1) > 1e-4).to(xs.dtype)).abs().mean()
    gb = torch.autograd.grad(l2, params)
    for a, b in zip(ga, gb):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)
    # ... and == the fused kernel's gradient (hjbx_value_loss_grad_f32, RAW residual, done = 0) divided by the batch size; its loss sum too
    z = torch.zeros(256, device="cuda")
    flat = _ops.value_loss_grad(d.system, ctl._task, ctl.value_function_approximator.descriptor(), xs, torch.ones_like(z), z, _abi.RESIDUAL_RAW)
    off = 0
    for b in gb:
        k = b.numel()
        fa = flat[off:off + k].view_as(b) / 256
        assert float((fa - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-7
        off += k
    P = off
    assert abs(float(flat[2 * P]) / 256 - float(want)) < 1e-5 * max(1.0, float(want)) and float(flat[2 * P + 2]) == 256 and float(flat[2 * P + 3]) == 0
    assert float(flat[P:2 * P].abs().max()) == 0.0           # no termination samples
    x0 = ctl.states[:64].clone()
    tt = ctl.time_to_target(x0, 2.0)
    assert torch.equal(x0, ctl.states[:64]) and tt.shape == (64,) and float(tt.max()) <= 2.0   # the caller's states are not stepped in place
    losses, means, stds = ctl.train(epochs=3)
    assert len(losses) == len(means) == len(stds) == 3
    assert losses[-1] < losses[0]
    assert all(0.0 <= m <= 15.0 for m in means)
    uu = ctl.get_control_efforts(np.array([[0.5, 0.5], [-0.5, -0.5]], np.float32))
    assert uu.shape == (2, 1) and set(np.unique(uu)).issubset({-1.0, 0.0, 1.0})


def test_bad_law_is_refused():
    d = _double_integrator()
    task = _abi.make_task(2, 1, np.eye(2), np.eye(1), None, [0, 0], [0], None, None, 0.0, law=7)
    x = torch.zeros((4, 2), device="cuda")
    with pytest.raises(ValueError, match="control law"):
        _ops.control_from_grad(d.system, task, x, x)
    task = _abi.make_task(2, 1, np.eye(2), np.eye(1), None, [0, 0], [0], None, None, 0.0, law=_abi.LAW_BANGBANG, target_r2=-1.0)
    with pytest.raises(ValueError, match="target_r2"):
        _ops.control_from_grad(d.system, task, x, x)


@pytest.mark.parametrize("fused", [True, False])
def test_time_optimal_graphed_update_equals_eager(fused, monkeypatch):
    """The learner's optimiser step replayed from a hipGraph == eager launches (same minibatches, same initial weights), through the fused
    parameter gradient + Adam kernels and through PyTorch autograd (HJBX_FUSED_PARAM_GRAD=0)."""
    monkeypatch.setenv("HJBX_FUSED_PARAM_GRAD", "1" if fused else "0")
    d = _double_integrator()
    d.integrator = _abi.ZOH
    a = TimeOptimalVHJBController(d, activation="sin", num_states=2048, seed=5)
    b = TimeOptimalVHJBController(d, activation="sin", num_states=2048, seed=5)
    assert a.fused_param_grad == fused
    pa, pb = list(a.value_function_approximator.parameters()), list(b.value_function_approximator.parameters())
    assert all(torch.equal(x, y) for x, y in zip(pa, pb)) and torch.equal(a.states, b.states)
    start = [p.detach().clone() for p in pa]
    for k in range(5):
        xs = a.states[256 * k:256 * (k + 1)].contiguous()
        la = float(a.params_update_graphed(xs))
        lb = float(b.params_update(xs))
        assert abs(la - lb) <= 1e-4 * max(1.0, abs(lb))
    for x, y in zip(pa, pb):
        np.testing.assert_allclose(x.detach().cpu().numpy(), y.detach().cpu().numpy(), rtol=2e-3, atol=2e-5)
    assert all(float((p - s).abs().max()) > 1e-3 for p, s in zip(pa, start))
    # a ragged last minibatch falls back to eager launches
    assert np.isfinite(float(a.params_update_graphed(a.states[:100].contiguous())))


def test_time_optimal_fused_and_autograd_updates_agree(monkeypatch):
    """Five updates of the sin learner through the fused kernels (hjbx_value_loss_grad_f32 + hjbx_mix_adam_f32) against five through autograd +
    torch.optim.Adam: same minibatches, losses to 1e-4, weights to a few percent of what they moved (Adam's first steps are +-lr sign(g):
    entries with |g| ~ 1e-8 may differ in sign)."""
    d = _double_integrator()
    d.integrator = _abi.ZOH
    ctls = []
    for fused in ("1", "0"):
        monkeypatch.setenv("HJBX_FUSED_PARAM_GRAD", fused)
        ctls.append(TimeOptimalVHJBController(d, activation="sin", num_states=2048, seed=5))
    a, b = ctls
    assert a.fused_param_grad and not b.fused_param_grad
    pa, pb = list(a.value_function_approximator.parameters()), list(b.value_function_approximator.parameters())
    start = [p.detach().clone() for p in pa]
    for k in range(5):
        xs = a.states[256 * k:256 * (k + 1)].contiguous()
        la, lb = float(a.params_update(xs)), float(b.params_update(xs))
        assert abs(la - lb) <= 1e-4 * max(1.0, abs(lb)), (k, la, lb)
    for x, y, s0 in zip(pa, pb, start):
        moved = float((y - s0).abs().max())
        diff = (x - y).abs()
        assert float(diff.median()) <= 0.02 * moved and float((diff > 0.5 * moved).float().mean()) < 0.01
