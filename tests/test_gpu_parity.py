"""GPU parity: every libhjbx.so entry point (through the C ABI) against the CPU oracle on the same
seeded inputs, and against the reference's golden vectors.  Run on the MI355X box with `-m gpu`.

Tolerances (stated per BASELINE.json north_star):
  * float64 kernels vs the f64 oracle: rtol 1e-12 (atol scaled to the data), integer outputs bit-equal;
  * float32 kernels vs the f64 oracle: rtol 1e-5 per step (atol 1e-5 x data scale);
  * angle columns are compared modulo 2 pi (a 1-ulp difference at the seam flips the representative).
"""
import numpy as np
import pytest
import torch

from conftest import ANGLE_IDX, SYSTEMS, load_golden, make_dynamics, make_vhjb_config, orc_system, wrapped_diff
from oracle import oracle as O
from parity_util import F32_ULP, FACTOR, abs_err, assert_within_cpu_yardstick, check, ratio_stats, residual_term_scales, step_term_scales
from q_learning_with_hjb_amd import _abi, _ops

pytestmark = pytest.mark.gpu

DT = {"f64": (torch.float64, np.float64, 1e-12), "f32": (torch.float32, np.float32, 1e-5)}


def dev(a, tdt):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=tdt, device="cuda").contiguous()


def sample_states(name, B, seed=0, spread=1.5):
    d = make_dynamics(name)
    rng = np.random.default_rng(seed)
    x = d.x0_mean.astype(np.float64) + rng.uniform(-1, 1, (B, d.state_dim)) * np.maximum(d.x0_std.astype(np.float64), 0.5) * spread
    u = rng.uniform(-1.3, 1.3, (B, d.control_dim)) * np.maximum(np.abs(d.umax), np.abs(d.umin)).astype(np.float64)
    return d, x, u


def task_for(name, d):
    cfg = make_vhjb_config(name)
    rng = np.random.default_rng(7)
    n, m = d.get_dimension()
    A = rng.standard_normal((n, n))
    P = A @ A.T / n + np.eye(n)       # any SPD terminal cost
    Q = np.asarray(cfg.Q, np.float64) + 0.1 * (A + A.T) / n * 0.2 + 0.0
    Q = Q @ Q.T                        # dense SPD, exercises the full quadratic form
    Rm = np.asarray(cfg.R, np.float64)
    if m > 1:
        Rm = Rm + 0.1 * np.ones((m, m)) / m
    return _abi.make_task(n, m, Q, Rm, P, cfg.xf, cfg.uf, cfg.obs_min, cfg.obs_max, cfg.epsilon)


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("name", SYSTEMS)
def test_pointwise_kernels(name, prec):
    tdt, ndt, tol = DT[prec]
    d, x, u = sample_states(name, 20001, seed=1)   # ragged: not a multiple of the 256-thread block; large enough for stable max / p99.9 statistics
    s = orc_system(name)
    xd, ud = dev(x, tdt), dev(u, tdt)
    xr, ur = xd.cpu().numpy().astype(np.float64), ud.cpu().numpy().astype(np.float64)  # oracle sees the rounded inputs
    ai = ANGLE_IDX[name]
    # every scale below is PER ELEMENT / PER ENVIRONMENT (the magnitudes of that element's own terms), never a batch-wide maximum
    f1, f2 = _ops.affine(d.system, xd)
    o1, o2 = O.affine(s, xr)
    B, n, m = x.shape[0], d.state_dim, d.control_dim
    row1 = np.abs(o1).max(1, keepdims=True)                                     # an env's f1 entries share their terms (gravity, Coriolis)
    check(f1, o1, tol, row1); check(f2, o2, tol, np.abs(o2).reshape(B, -1).max(1)[:, None, None])
    S_xd = np.abs(o1) + np.einsum("bkj,bj->bk", np.abs(o2), np.abs(ur)) + row1   # xdot_k = f1_k + sum_j f2_kj u_j
    check(_ops.dynamics_step(d.system, xd, ud), O.dynamics_step(s, xr, ur), tol, S_xd)
    check(_ops.wrap(d.system, xd), O.wrap(s, xr), tol, np.pi, angle_idx=ai)
    uc = np.clip(ur, d.umin.astype(np.float64), d.umax.astype(np.float64))
    S_sim = np.abs(xr) + float(d.dt) * (np.abs(o1) + np.einsum("bkj,bj->bk", np.abs(o2), np.abs(uc)) + row1)
    S_sim[:, ai] += np.pi
    for integ in (_abi.EULER, _abi.RK4):
        check(_ops.simulate(d.system, xd, ud, integ), O.simulate(s, xr, ur, integ), tol, S_sim, angle_idx=ai)
        if prec == "f32":
            assert_within_cpu_yardstick(f"simulate integrator {integ}", _ops.simulate(d.system, xd, ud, integ), O.simulate(s, xr, ur, integ, dtype=np.float32),
                                        O.simulate(s, xr, ur, integ), S_sim, angle_idx=ai)
    task = task_for(name, d)
    e = np.abs(O.wrap(s, xr - np.array(task.xf[:n], np.float64)[None, :]))
    Qa, Ra, Pa = (np.abs(np.array(v[: k * k], np.float64).reshape(k, k)) for v, k in ((task.Q, n), (task.R, m), (task.P, n)))
    dua = np.abs(ur - np.array(task.uf[:m], np.float64)[None, :])
    check(_ops.running_cost(d.system, task, xd, ud), O.running_cost(s, task, xr, ur), tol, np.einsum("bi,ij,bj->b", e, Qa, e) + np.einsum("bi,ij,bj->b", dua, Ra, dua))
    check(_ops.termination_cost(d.system, task, xd), O.termination_cost(s, task, xr), tol, np.einsum("bi,ij,bj->b", e, Pa, e))
    rng = np.random.default_rng(3)
    g = rng.standard_normal(x.shape) * 20
    gd = dev(g, tdt); gr = gd.cpu().numpy().astype(np.float64)
    Rinva = np.abs(np.array(task.Rinv[: m * m], np.float64).reshape(m, m))
    S_u = np.abs(d.umax).astype(np.float64)[None, :] + 0.5 * np.einsum("jq,bkq,bk->bj", Rinva, np.abs(o2), np.abs(gr))
    check(_ops.control_from_grad(d.system, task, xd, gd), O.control_from_grad(s, task, xr, gr), tol, S_u)
    u01 = rng.uniform(size=x.shape)
    ud01 = dev(u01, tdt)
    S_x0 = (np.abs(d.x0_mean) + np.abs(d.x0_std)).astype(np.float64)[None, :] + 0.0 * u01
    S_x0[:, ai] += np.pi
    check(_ops.initial_state(d.system, d.x0_mean, d.x0_std, ud01), O.initial_state(s, d.x0_mean, d.x0_std, ud01.cpu().numpy().astype(np.float64)),
          tol, S_x0, angle_idx=ai)


@pytest.mark.parametrize("name", SYSTEMS)
def test_golden_vectors_f64(name):
    """The f64 kernels directly against the reference's own outputs (no oracle in between)."""
    g = load_golden(name)
    d = make_dynamics(name)
    x, u = dev(g["X"], torch.float64), dev(g["U"], torch.float64)
    f1, f2 = _ops.affine(d.system, x)
    B = g["X"].shape[0]
    row1 = np.abs(g["F1"]).max(1, keepdims=True)
    check(f1, g["F1"], 1e-12, row1); check(f2, g["F2"], 1e-12, np.abs(g["F2"]).reshape(B, -1).max(1)[:, None, None])
    S_xd = np.abs(g["F1"]) + np.einsum("bkj,bj->bk", np.abs(g["F2"]), np.abs(g["U"]).reshape(B, -1)) + row1
    check(_ops.dynamics_step(d.system, x, u), g["XDOT"], 1e-12, S_xd)
    S_sim = np.abs(g["X"]) + float(d.dt) * S_xd
    S_sim[:, ANGLE_IDX[name]] += np.pi
    check(_ops.simulate(d.system, x, u), g["XNEXT"], 1e-12, S_sim, angle_idx=ANGLE_IDX[name])
    # wrap seams: bit exact
    got = _ops.wrap(d.system, dev(g["XSEAM"], torch.float64)).cpu().numpy()
    assert np.array_equal(got, g["XSEAMWRAP"])
    if name != "acrobot":
        x0 = _ops.initial_state(d.system, d.x0_mean, d.x0_std, dev(g["U01SEQ"], torch.float64))
        check(x0, g["X0SEQ"], 1e-14)


@pytest.mark.parametrize("name", ["linear", "cartpole", "quad2d", "nearhover"])
def test_dynamics_surface_numpy_roundtrip(name):
    """The reference-shaped Python surface: numpy (n,) in -> numpy out, incl. get_initial_state's RNG stream."""
    g = load_golden(name)
    d = make_dynamics(name)
    x0 = d.get_initial_state()
    assert isinstance(x0, np.ndarray) and x0.dtype == np.float64 and x0.shape == (d.state_dim,)
    np.testing.assert_allclose(x0, g["X0SEQ"][0], rtol=1e-14, atol=1e-15)
    f1, f2 = d.get_control_affine_matrix(g["X"][5])
    np.testing.assert_allclose(f1, g["F1"][5], rtol=1e-12, atol=1e-11)
    assert f2.shape == (d.state_dim, d.control_dim)
    np.testing.assert_allclose(d.dynamics_step(g["X"][5], g["U"][5]), g["XDOT"][5], rtol=1e-12, atol=1e-10)
    xn = d.simulate(g["X"][5], g["U"][5])
    assert np.abs(wrapped_diff(xn, g["XNEXT"][5], ANGLE_IDX[name])).max() < 1e-11
    assert d.get_dimension() == (d.state_dim, d.control_dim)
    xb = d.simulate(g["X"], g["U"])           # batch
    assert xb.shape == g["X"].shape
    xt = d.simulate(torch.as_tensor(g["X"], device="cuda"), torch.as_tensor(g["U"], device="cuda"))  # zero-copy flavour
    assert xt.is_cuda and torch.equal(xt.cpu(), torch.as_tensor(xb))


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("mode", [_abi.RESIDUAL_NORMALISED, _abi.RESIDUAL_RAW])
@pytest.mark.parametrize("name", SYSTEMS)
def test_hjb_residual(name, mode, prec):
    tdt, ndt, tol = DT[prec]
    d, x, _ = sample_states(name, 20077, seed=5)      # ragged, and large enough for the max / p99.9 statistics of the yardstick to be stable
    s = orc_system(name)
    task = task_for(name, d)
    rng = np.random.default_rng(11)
    g = rng.standard_normal(x.shape) * np.where(rng.uniform(size=(x.shape[0], 1)) < 0.5, 2.0, 40.0)  # clipped and unclipped controls
    done = (rng.uniform(size=x.shape[0]) < 0.3).astype(np.float64)
    xd, gd, dd = dev(x, tdt), dev(g, tdt), dev(done, tdt)
    xr, gr = xd.cpu().numpy().astype(np.float64), gd.cpu().numpy().astype(np.float64)
    li, dg, sums = _ops.hjb_residual(d.system, task, xd, gd, dd, mode)
    oli, odg, osums = O.hjb_residual(s, task, xr, gr, done, mode)
    # per element: 1e-12 (f64) / 1e-5 (f32) of the element's own term scale (parity_util.residual_term_scales), no factor; float32 also against
    # the CPU oracle compiled for float (the neutral yardstick)
    S_li, S_dg = residual_term_scales(d, task, s, xr, gr, mode)
    w = 1.0 - done
    check(li, oli, tol, S_li * w)
    check(dg, odg, tol, S_dg * w[:, None])
    if prec == "f32":
        cli, cdg, _ = O.hjb_residual(s, task, xr, gr, done, mode, dtype=np.float32)
        live = done == 0
        assert_within_cpu_yardstick("loss_i", li, cli, oli, S_li, keep=live)
        assert_within_cpu_yardstick("dloss/dgradV", dg, cdg, odg, S_dg, keep=live)
    got = sums.cpu().numpy().astype(np.float64)
    # the kernel accumulates in double: the sum's error is the sum of the elements' errors (+ the final rounding to T)
    assert abs(got[0] - osums[0]) <= tol * (S_li * w).sum() + tol * abs(osums[0])
    assert got[1] == osums[1] and got[2] == osums[2]         # counts are exact
    # determinism: the two-stage reduction has no atomics
    _, _, sums2 = _ops.hjb_residual(d.system, task, xd, gd, dd, mode)
    assert torch.equal(sums, sums2)


def test_hjb_residual_gradient_matches_finite_differences():
    """d loss_i / d gradV from the kernel vs central differences of the kernel's own loss (f64)."""
    d, x, _ = sample_states("quad2d", 64, seed=9)
    task = task_for("quad2d", d)
    rng = np.random.default_rng(2)
    g = rng.standard_normal(x.shape) * 3
    done = np.zeros(x.shape[0])
    xd, dd = dev(x, torch.float64), dev(done, torch.float64)
    _, dg, _ = _ops.hjb_residual(d.system, task, xd, dev(g, torch.float64), dd)
    h = 1e-6
    num = np.zeros_like(g)
    for k in range(g.shape[1]):
        gp, gm = g.copy(), g.copy()
        gp[:, k] += h; gm[:, k] -= h
        lp = _ops.hjb_residual(d.system, task, xd, dev(gp, torch.float64), dd)[0].cpu().numpy()
        lm = _ops.hjb_residual(d.system, task, xd, dev(gm, torch.float64), dd)[0].cpu().numpy()
        num[:, k] = (lp - lm) / (2 * h)
    np.testing.assert_allclose(dg.cpu().numpy(), num, rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_known_answer_lqr_residual_is_zero(prec):
    """for LinearDynamics, gradV = 2 P e gives u = -K e and gradV.xdot + l == 0, so the
    normalised residual |.. + 1| vanishes (identity documented at reference utils/utils.py:33,58)."""
    tdt, ndt, tol = DT[prec]
    import scipy.linalg
    d = make_dynamics("linear")
    A, Bm = np.asarray(d.A, np.float64), np.asarray(d.B, np.float64)
    Q, R = np.eye(2), np.eye(1)
    P = scipy.linalg.solve_continuous_are(A, Bm, Q, R)
    K = np.linalg.inv(R) @ Bm.T @ P
    task = _abi.make_task(2, 1, Q, R, P, [0, 0], [0], [-2, -3], [2, 3], 1e-10)
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (4096, 2))             # |K x| < 5: never clipped
    g = 2 * x @ P
    xd, gd = dev(x, tdt), dev(g, tdt)
    u = _ops.control_from_grad(d.system, task, xd, gd)
    check(u, -(x @ K.T), tol, 0.5 * np.abs(g) @ np.abs(Bm))       # u = -1/2 Rinv B' g: |terms| = 1/2 sum_k |B_k g_k| (Rinv = 1)
    li, _, sums = _ops.hjb_residual(d.system, task, xd, gd, torch.zeros(4096, dtype=tdt, device="cuda"))
    # exactly: vdot = -l, so r = -l/(l+eps) + 1 = eps/(l+eps)  (tiny except next to the origin)
    l = (x * x).sum(1) + ((x @ K.T) ** 2).sum(1)
    lim = 1e-9 if prec == "f64" else 2e-3         # fp32: cancellation of two O(1) terms, amplified by 1/l
    mask = torch.as_tensor(l > (1e-6 if prec == "f64" else 1e-2))
    assert float((li.cpu() - torch.as_tensor(1e-10 / (l + 1e-10), dtype=li.dtype))[mask].abs().max()) < lim
    lr, _, _ = _ops.hjb_residual(d.system, task, xd, gd, torch.zeros(4096, dtype=tdt, device="cuda"), _abi.RESIDUAL_RAW)
    assert float(lr.abs().max()) < (1e-12 if prec == "f64" else 1e-5)


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_termination_residual(prec):
    tdt, ndt, tol = DT[prec]
    rng = np.random.default_rng(4)
    B = 5000
    V = rng.uniform(0, 50, B); cost = rng.uniform(0, 50, B); cost[::7] = 0.0
    done = (rng.uniform(size=B) < 0.5).astype(np.float64)
    Vd, cd, dd = dev(V, tdt), dev(cost, tdt), dev(done, tdt)
    li, dv, sums = _ops.termination_residual(1e-10 if prec == "f64" else 1e-6, Vd, cd, dd)
    oli, odv, osums = O.termination_residual(1e-10 if prec == "f64" else 1e-6, Vd.cpu().numpy().astype(np.float64),
                                             cd.cpu().numpy().astype(np.float64), done)
    m = cost > 0
    Vr, cr = Vd.cpu().numpy().astype(np.float64)[m], cd.cpu().numpy().astype(np.float64)[m]
    eps_t = 1e-10 if prec == "f64" else 1e-6
    check(li[torch.as_tensor(m)], oli[m], tol, (Vr / (cr + eps_t) + 1.0) * done[m])   # |V / (c + eps) - 1|: terms V / (c + eps) and 1
    check(dv[torch.as_tensor(m)], odv[m], tol, 0.0)                                     # sign(.) done / (c + eps): one term, relative
    assert float(sums[2]) == osums[2]


def _ctrl_objects(name):
    """product controller + matching oracle descriptor from the SAME gains"""
    d = make_dynamics(name)
    if name == "linear":
        from q_learning_with_hjb_amd.controller.lqr import LQR
        c = LQR(d, np.eye(2), np.eye(1))
    elif name == "cartpole":
        from q_learning_with_hjb_amd.controller.cartpole_energy_shaping import CartpoleEnergyShapingController
        c = CartpoleEnergyShapingController(d)
    elif name == "acrobot":
        from q_learning_with_hjb_amd.controller.acrobot_energy_shaping import AcrobotEnergyShapingController
        c = AcrobotEnergyShapingController(d)
    elif name == "quad2d":
        from q_learning_with_hjb_amd.controller.quadrotors_model_based_controller import Quadrotors2DHoveringController
        c = Quadrotors2DHoveringController(d, np.zeros(6), np.eye(6), np.eye(2))
    else:
        from q_learning_with_hjb_amd.controller.quadrotors_model_based_controller import NearHoverQuadcopterHoveringController
        c = NearHoverQuadcopterHoveringController(d, np.zeros(10), np.eye(10), np.eye(3))
    return d, c


TRAJ = {"linear": "traj_linear_lqr", "cartpole": "traj_cartpole_es", "acrobot": "traj_acrobot_es", "quad2d": "traj_quad2d_hover",
        "nearhover": "traj_nearhover_hover"}


@pytest.mark.parametrize("name", SYSTEMS)
def test_controller_gains_and_golden_closed_loop(name):
    """Set-up math (CARE, gains) and the fused closed-loop kernel against the reference's trajectories
    (config 1: double integrator + LQR for 5 s; energy shaping; hover LQR)."""
    g = load_golden(TRAJ[name])
    d, c = _ctrl_objects(name)
    K = c.K if name in ("linear", "quad2d", "nearhover") else c.get_lqr_term()[0]
    np.testing.assert_allclose(K, g["K"], rtol=1e-9, atol=1e-9)
    T = g["US"].shape[0]
    out = c.rollout(g["XS"][0][None, :], T)           # numpy f64 in -> f64 kernels
    dd = np.abs(wrapped_diff(out["traj"][:, 0], g["XS"], ANGLE_IDX[name]))
    du = np.abs(out["u"][:, 0] - g["US"].reshape(T, -1))
    if name in ("cartpole", "acrobot"):
        assert dd[:150].max() < 1e-7 and du[:150].max() < 2e-6 and dd.max() < 1e-3   # swing-up: errors grow along the unstable loop
    else:
        assert dd.max() < 1e-9 and du.max() < 1e-9
    assert int(out["done_step"][0]) == T
    # single-state call of the plugin method
    u0 = c.get_control_efforts(g["XS"][0])
    np.testing.assert_allclose(np.atleast_1d(u0), g["US"][0].reshape(-1), rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("name", SYSTEMS)
def test_rollout_feedback_vs_oracle(name, prec):
    """Fused T-step kernel == oracle closed loop, with costs, termination and integer done steps."""
    tdt, ndt, tol = DT[prec]
    d, c = _ctrl_objects(name)
    s = orc_system(name)
    cfg = make_vhjb_config(name)
    n, m = d.get_dimension()
    task = _abi.make_task(n, m, cfg.Q, cfg.R, np.eye(n) * 2.0, c.xf if hasattr(c, "xf") else cfg.xf, getattr(c, "uf", cfg.uf), cfg.obs_min,
                          cfg.obs_max, cfg.epsilon)
    B, T = 3001, 60
    rng = np.random.default_rng(21)
    xf = np.asarray(c.xf if hasattr(c, "xf") else cfg.xf, np.float64)
    x0 = xf + rng.uniform(-1, 1, (B, n)) * np.asarray(cfg.obs_max, np.float64).clip(max=3.0) * 0.9
    x0d = dev(x0, tdt); x0r = x0d.cpu().numpy().astype(np.float64)
    desc = c._descriptor()
    for terminate in (False, True):
        for integ in (_abi.EULER, _abi.RK4):
            got = _ops.rollout_feedback(d.system, desc, x0d, T, task=task, integrator=integ, terminate=terminate, log_u=True, log_cost=True)
            want = O.rollout_feedback(s, desc, x0r, T, task=task, integrator=integ, terminate=terminate)
            gs, ws = got["done_step"].cpu().numpy(), want["done_step"]
            if prec == "f64":
                assert np.array_equal(gs, ws)
                keep = np.ones(B, bool)
            else:
                # fp32 may cross a bound one step apart for envs within rounding of it: allow <1 %
                keep = gs == ws
                assert keep.mean() > 0.99 and np.abs(gs - ws).max() <= 1
            # compare a bounded prefix (error growth on the unstable plants is exponential in t).  Scales are PER ENVIRONMENT: the range of
            # that environment's own trajectory / costs over the prefix (a closed loop mixes the coordinates), +pi for wrapped angles.
            P = 12
            km = torch.as_tensor(keep)
            S_env = np.abs(want["traj"][:P]).max(axis=(0, 2))[None, :, None] + np.zeros((1, 1, n))
            S_env[..., ANGLE_IDX[name]] += np.pi
            S_u = np.abs(d.umax).astype(np.float64)[None, None, :]
            S_c = np.abs(want["cost"][:P]).max(axis=0)[None, :] + float(d.dt)
            if prec == "f64":
                check(got["traj"][:P, km], want["traj"][:P, keep], 1e-9, S_env[:, keep], angle_idx=ANGLE_IDX[name])
                check(got["u"][:P, km], want["u"][:P, keep], 1e-9, S_u)
                check(got["cost"][:P, km], want["cost"][:P, keep], 1e-9, S_c[:, keep])
            else:
                # float32: the CPU oracle compiled for float runs the same loop; the kernel's error distribution over the prefix must stay
                # within 2x its.  The energy-shaping laws switch branch discontinuously (a few environments next to the switching surface
                # take the other branch for a step in ANY float32 evaluation, then diverge): there the 98th percentile is compared, not the max.
                c32 = O.rollout_feedback(s, desc, x0r, T, task=task, integrator=integ, terminate=terminate, dtype=np.float32)
                both = keep & (c32["done_step"] == ws)
                qs = (0.5, 0.98) if name in ("cartpole", "acrobot") else (0.5, 0.999, 1.0)
                for key, S, ai_ in (("traj", S_env, ANGLE_IDX[name]), ("u", S_u, ()), ("cost", S_c, ())):
                    Sb = np.broadcast_to(S, want[key][:P].shape)[:, both]
                    eg = abs_err(got[key][:P].cpu().numpy()[:, both], want[key][:P, both], ai_) / Sb
                    ec = abs_err(c32[key][:P][:, both], want[key][:P, both], ai_) / Sb
                    for q in qs:
                        a, b = np.quantile(eg, q), np.quantile(ec, q)
                        assert a <= FACTOR * max(b, F32_ULP), f"{name} {key} terminate={terminate} integrator={integ}: q{q} of err / scale {a:.2e} vs CPU float32 {b:.2e}"
                    # and the stated tolerance: the typical element within 1e-5 of its environment's scale after 12 closed-loop steps
                    assert np.median(eg) <= tol
            if prec == "f64":
                # the whole 60-step horizon: last-bit differences (fma contraction, summation order) grow exponentially on the unstable
                # plants (the acrobot loop amplifies 1e-16 to ~1e-6 over 60 steps), hence 1e-5 here against 1e-9 for the prefix above
                S_all = np.abs(want["traj"]).max(axis=(0, 2))[None, :, None] + np.zeros((1, 1, n))       # per environment: its own range
                S_all[..., ANGLE_IDX[name]] += np.pi
                check(got["traj"], want["traj"], 1e-5, S_all, angle_idx=ANGLE_IDX[name])
                check(got["total_cost"], want["total_cost"], 1e-5, float(d.dt))                          # a sum of non-negative terms: relative
                check(got["x_final"], want["x_final"], 1e-5, S_all[0], angle_idx=ANGLE_IDX[name])


@pytest.mark.parametrize("name", ["linear", "cartpole", "quad2d", "nearhover"])
def test_rollout_feedback_equals_step_by_step(name):
    """Size-independent property at a larger batch: the fused kernel == controller + simulate kernels
    applied T times (bitwise, same arithmetic), f32, B = 2^16."""
    d, c = _ctrl_objects(name)
    B, T = 1 << 16, 25
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    x0 = d.get_initial_state(B, generator=gen)
    out = _ops.rollout_feedback(d.system, c._descriptor(), x0, T, log_u=True)
    x = x0.clone()
    for t in range(T):
        assert torch.equal(out["traj"][t], x)
        u = _ops.controller(d.system, c._descriptor(), x)
        assert torch.equal(out["u"][t], u)
        x = _ops.simulate(d.system, x, u)
    assert torch.equal(out["traj"][T], x) and torch.equal(out["x_final"], x)


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("name", SYSTEMS)
def test_vhjb_step_sequence(name, prec):
    """The per-step closed-loop kernel against the oracle, fed the same (arbitrary) value gradients:
    costs, done flags, held states and bit-equal done_step indices, including forced termination at T."""
    tdt, ndt, tol = DT[prec]
    d = make_dynamics(name)
    s = orc_system(name)
    cfg = make_vhjb_config(name)
    n, m = d.get_dimension()
    task = _abi.make_task(n, m, cfg.Q, cfg.R, np.eye(n) * 3.0, cfg.xf, cfg.uf, cfg.obs_min, cfg.obs_max, cfg.epsilon)
    import types
    tk = types.SimpleNamespace(R=cfg.R, R_inv=np.linalg.inv(np.asarray(cfg.R, np.float64)), uf=cfg.uf, epsilon=cfg.epsilon)
    B, T = 20001, 12
    rng = np.random.default_rng(33)
    xf = np.asarray(cfg.xf, np.float64)
    x = xf + rng.uniform(-1.05, 1.05, (B, n)) * np.asarray(cfg.obs_max, np.float64).clip(max=3.0)   # ~some start outside the box
    xd = dev(x, tdt); xo = xd.cpu().numpy().astype(np.float64)
    ds_d = torch.full((B,), -1, dtype=torch.int32, device="cuda"); ds_o = np.full(B, -1, np.int32)
    for t in range(T + 1):
        g = rng.standard_normal((B, n)) * 5
        gd = dev(g, tdt); gr = gd.cpu().numpy().astype(np.float64)
        xn = torch.empty_like(xd); c = torch.empty(B, dtype=tdt, device="cuda"); dn = torch.empty_like(c)
        uo = torch.empty((B, m), dtype=tdt, device="cuda"); rs = torch.empty(B, dtype=tdt, device="cuda")
        _ops.vhjb_step(d.system, task, t, T, xd, gd, xn, c, dn, ds_d, u_out=uo, resid_t=rs)
        ds_prev = ds_o
        oxn, ou, oc, od, ds_o, ors = O.vhjb_step(s, task, t, T, xo, gr, ds_o)
        if prec == "f64":
            assert np.array_equal(ds_d.cpu().numpy(), ds_o)
        else:
            agree = ds_d.cpu().numpy() == ds_o
            assert agree.mean() > 0.99
            # re-synchronise the few envs that sit within fp32 rounding of a bound
            ds_d = torch.as_tensor(ds_o, device="cuda"); xn = torch.where(torch.as_tensor(agree, device="cuda")[:, None], xn, dev(oxn, tdt))
            keep = agree
        keep = np.ones(B, bool) if prec == "f64" else keep
        km = torch.as_tensor(keep)
        # per element: tol x the element's own term scale (parity_util.step_term_scales; the gradient is an input here, so its term scale is
        # |g|), no factors; float32 also against the CPU oracle compiled for float
        S = step_term_scales(name, d, tk, s, xo, gr, np.abs(gr), ou, oc)
        lv = keep & (ds_o < 0)                                              # environments that took a live step
        check(xn[km], oxn[keep], tol, S["x_next"][keep], angle_idx=ANGLE_IDX[name])
        e_abs = np.abs(O.wrap(s, xo - xf[None, :]))
        S_term = 3.0 * (e_abs ** 2).sum(1) + 6.0 * (e_abs * (np.abs(xo) + np.abs(xf)[None, :])).sum(1)   # e'Pe, P = 3 I, e = wrap(x - xf) cancels
        check(c[km], oc[keep], tol, np.where(ds_o < 0, S["cost"], np.where(ds_o == t, S_term, 0.0))[keep])
        check(uo[km], ou[keep], tol, S["u"][keep])
        check(rs[km], ors[keep], tol, np.where(ds_o < 0, S["residual"], 0.0)[keep])     # fused HJB residual by-product
        if prec == "f32" and lv.sum() > 50:
            cxn, cu, cc, _, _, crs = O.vhjb_step(s, task, t, T, xo, gr, ds_prev, dtype=np.float32)
            for key, got_, cpu_, want_, ai_ in (("x_next", xn, cxn, oxn, ANGLE_IDX[name]), ("u", uo, cu, ou, ()), ("cost", c, cc, oc, ()), ("residual", rs, crs, ors, ())):
                assert_within_cpu_yardstick(f"step {t} {key}", got_, cpu_, want_, S[key], angle_idx=ai_, keep=lv)
        assert np.array_equal(dn[km].cpu().numpy().astype(np.float64), od[keep])
        xd = xn; xo = xn.cpu().numpy().astype(np.float64) if prec == "f32" else oxn
        if prec == "f64":
            xd = dev(oxn, tdt)
    assert (ds_o >= 0).all() and ds_o.max() <= T


def test_edge_cases_and_errors():
    d = make_dynamics("cartpole")
    # empty batch: a no-op that still returns well-formed tensors
    x = torch.empty((0, 4), device="cuda")
    assert _ops.simulate(d.system, x, torch.empty((0, 1), device="cuda")).shape == (0, 4)
    li, dg, sums = _ops.hjb_residual(d.system, task_for("cartpole", d), x, x.clone(), torch.empty((0,), device="cuda"))
    assert sums.tolist() == [0.0, 0.0, 0.0]
    # B = 1
    one = _ops.wrap(d.system, torch.tensor([[0.0, 7.0, 0.0, 0.0]], device="cuda"))
    assert abs(float(one[0, 1]) - (7.0 - 2 * np.pi)) < 1e-6
    # misaligned pointer -> ValueError from HJBX_EINVAL (a (B,4) f32 view shifted by one element)
    buf = torch.zeros(4 * 8 + 1, device="cuda")
    with pytest.raises(ValueError):
        _ops.wrap(d.system, buf[1:].view(8, 4), out=torch.empty((8, 4), device="cuda"))
    # wrong controller for the system
    c = _abi.make_controller(_abi.CTRL_ACROBOT_ENERGY, 4, 1, np.zeros((1, 4)))
    with pytest.raises(ValueError):
        _ops.controller(d.system, c, torch.zeros((4, 4), device="cuda"))
    # unsupported linear shape
    from q_learning_with_hjb_amd.configs.defaults import linear_dynamics_config
    from q_learning_with_hjb_amd.dynamics.linear import LinearDynamics
    big = LinearDynamics(linear_dynamics_config(A=np.eye(3).tolist(), B=np.ones((3, 1)).tolist(), x0_mean=[0] * 3, x0_std=[1] * 3))
    with pytest.raises(NotImplementedError):
        big.simulate(np.zeros(3), np.zeros(1))
    # NaN safety: a diverged NearHover env (tan at pi/2) must not poison its neighbours
    q = make_dynamics("nearhover")
    xs = torch.zeros((4, 10), device="cuda"); xs[1, 3] = float("nan")
    out = _ops.simulate(q.system, xs, torch.zeros((4, 3), device="cuda"))
    assert torch.isnan(out[1]).any() and not torch.isnan(out[[0, 2, 3]]).any()


def test_full_size_properties_cartpole():
    """BASELINE configs[1] size (B = 2^20, f32): size-independent properties of the step kernels."""
    d = make_dynamics("cartpole")
    B = 1 << 20
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    x = d.get_initial_state(B, generator=gen)
    lo = torch.as_tensor(d.x0_mean - d.x0_std, device="cuda"); hi = torch.as_tensor(d.x0_mean + d.x0_std, device="cuda")
    assert bool(((x[:, [0, 2, 3]] >= lo[[0, 2, 3]] - 1e-6) & (x[:, [0, 2, 3]] <= hi[[0, 2, 3]] + 1e-6)).all())
    w = _ops.wrap(d.system, x)
    assert torch.equal(_ops.wrap(d.system, w), w)                                # idempotent
    assert float(w[:, 1].min()) >= -np.pi - 1e-6 and float(w[:, 1].max()) < np.pi + 1e-6
    u = torch.zeros((B, 1), device="cuda")
    a = _ops.simulate(d.system, x, u); b = _ops.simulate(d.system, x, u)
    assert torch.equal(a, b)                                                    # deterministic
    big = _ops.simulate(d.system, x, u + 1e6)                                   # clipping: u=1e6 == u=umax
    assert torch.equal(big, _ops.simulate(d.system, x, u + float(d.umax[0])))
    # Euler step of the cart position is exactly x + dt * xdot
    assert torch.allclose(a[:, 0], x[:, 0] + d.dt * x[:, 2], rtol=0, atol=1e-6)
    # oracle spot-check on a strided sample of the big batch
    idx = torch.arange(0, B, 4099, device="cuda")
    want = O.simulate(orc_system("cartpole"), x[idx].cpu().numpy().astype(np.float64), np.zeros((idx.numel(), 1)))
    check(a[idx], want, 1e-5, 3.0, angle_idx=[1])


@pytest.mark.parametrize("name,B,T", [("acrobot", 1 << 20, 40), ("quad2d", 1 << 18, 60), ("nearhover", 1 << 18, 40)])
def test_full_size_closed_loops(name, B, T):
    """BASELINE configs[2] (acrobot energy shaping, B = 2^20) and configs[3] (Quadrotors2D hover, B = 2^18) at full
    size through size-independent properties, plus an oracle check on a strided sample."""
    d, c = _ctrl_objects(name)
    cfg = make_vhjb_config(name)
    n, m = d.get_dimension()
    task = _abi.make_task(n, m, cfg.Q, cfg.R, np.eye(n), c.xf, getattr(c, "uf", cfg.uf), cfg.obs_min, cfg.obs_max, cfg.epsilon)
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    x0 = d.get_initial_state(B, generator=gen)
    desc = c._descriptor()
    a = _ops.rollout_feedback(d.system, desc, x0, T, task=task, terminate=True, log_traj=True, log_u=False, log_cost=True)
    b = _ops.rollout_feedback(d.system, desc, x0, T, task=task, terminate=True, log_traj=False)
    assert torch.equal(a["done_step"], b["done_step"]) and torch.equal(a["x_final"], b["x_final"])      # logging changes nothing
    assert torch.equal(a["total_cost"], b["total_cost"])
    ds = a["done_step"].long()
    assert int(ds.min()) >= 0 and int(ds.max()) <= T
    # the state is held after termination; the final state equals the log at the terminal index
    idx = ds.clamp(max=T)
    gathered = a["traj"][idx, torch.arange(B, device="cuda")]
    assert torch.equal(gathered, a["x_final"])
    # per-step costs add up to total_cost (f32 summation order differs: loose tolerance), zero after termination
    tot = a["cost"].double().sum(0)
    assert torch.allclose(tot, a["total_cost"].double(), rtol=1e-4, atol=1e-4)
    after = torch.arange(T + 1, device="cuda")[:, None] > ds[None, :]
    assert float(a["cost"][after].abs().max()) == 0.0
    # oracle on a strided sample, short prefix (the closed loops amplify fp32 rounding with t)
    sel = torch.arange(0, B, B // 509, device="cuda")
    ref = O.rollout_feedback(orc_system(name), desc, x0[sel].cpu().numpy().astype(np.float64), T, task=task, terminate=True)
    gs, ws = a["done_step"][sel].cpu().numpy(), ref["done_step"]
    keep = gs == ws
    assert keep.mean() > 0.97
    mbf = 0.02 if name in ("acrobot", "cartpole") else 0.0
    check(a["traj"][:10, sel][:, torch.as_tensor(keep)], ref["traj"][:10, keep], 5e-4, np.abs(ref["traj"][:10]).max(), angle_idx=ANGLE_IDX[name],
          max_bad_frac=mbf)


def test_nan_state_terminates_instead_of_propagating():
    """A diverged environment (NaN state) is latched as done by the step kernel; its neighbours are unaffected."""
    d = make_dynamics("nearhover")
    cfg = make_vhjb_config("nearhover")
    task = _abi.make_task(10, 3, cfg.Q, cfg.R, np.eye(10), cfg.xf, cfg.uf, cfg.obs_min, cfg.obs_max, cfg.epsilon)
    x = torch.zeros((8, 10), device="cuda"); x[3, 4] = float("nan")
    g = torch.zeros_like(x); xn = torch.empty_like(x); c = torch.empty(8, device="cuda"); dn = torch.empty(8, device="cuda")
    ds = torch.full((8,), -1, dtype=torch.int32, device="cuda")
    _ops.vhjb_step(d.system, task, 5, 100, x, g, xn, c, dn, ds)
    assert ds.tolist() == [-1, -1, -1, 5, -1, -1, -1, -1] and dn.tolist() == [0, 0, 0, 1, 0, 0, 0, 0]
    assert not torch.isnan(xn[[0, 1, 2, 4, 5, 6, 7]]).any()


def test_abi_argument_validation():
    """Misuse is reported through status codes (mapped to Python exceptions), never a crash or a silent no-op."""
    import ctypes as C
    L = _abi.lib()
    d = make_dynamics("cartpole")
    x = torch.zeros((16, 4), device="cuda")
    # NULL array pointer
    rc = L.hjbx_wrap_f32(d.system.ptr, None, x.data_ptr(), 16, None)
    assert rc == _abi.EINVAL and "non-NULL" in _abi.last_error()
    # NULL system handle
    assert L.hjbx_wrap_f32(None, x.data_ptr(), x.data_ptr(), 16, None) == _abi.EINVAL
    # negative batch
    assert L.hjbx_simulate_f32(d.system.ptr, 0, x.data_ptr(), x.data_ptr(), x.data_ptr(), -1, None) == _abi.EINVAL
    # unknown integrator / residual mode
    u = torch.zeros((16, 1), device="cuda")
    with pytest.raises(ValueError, match="integrator"):
        _ops.simulate(d.system, x, u, integrator=7)
    with pytest.raises(ValueError, match="residual mode"):
        _ops.hjb_residual(d.system, task_for("cartpole", d), x, x.clone(), torch.zeros(16, device="cuda"), mode=5)
    # wrong dtype / device / shape are caught before the ABI
    with pytest.raises(TypeError):
        _ops.wrap(d.system, x.to(torch.float16))
    with pytest.raises(TypeError):
        _ops.wrap(d.system, x.cpu())
    with pytest.raises(ValueError):
        _ops.wrap(d.system, torch.zeros((16, 5), device="cuda"))
    # fused value network: only the reference's 128-128-64 widths, non-zero std
    from q_learning_with_hjb_amd.controller.vhjb import VHJBController
    ctl = VHJBController(d, make_vhjb_config("cartpole", features=[64, 64, 32]))
    with pytest.raises(NotImplementedError, match="128,128,64"):
        ctl.value_function_approximator.fused_value_grad(x)
    g = ctl.value_function_approximator.value_and_grad(x)[1]            # the PyTorch path still serves other widths
    assert g.shape == (16, 4)
    with pytest.raises(ValueError, match="normalization_std"):
        make_vhjb_config("cartpole", normalization_std=[1, 0, 1, 1])


def _double_integrator(dt=0.01, umax=1.0):
    from q_learning_with_hjb_amd.configs.defaults import linear_dynamics_config
    from q_learning_with_hjb_amd.dynamics.linear import LinearDynamics
    return LinearDynamics(linear_dynamics_config(dt=dt, umin=[-umax], umax=[umax]))


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_zoh_integrator_and_time_optimal_controller(prec):
    """HJBX_ZOH (exact discretisation of LinearDynamics) and the analytic minimum-time law of the double integrator:
    kernels vs oracle, and the closed loop against the reference's .mat ground truth."""
    tdt, ndt, tol = DT[prec]
    d = _double_integrator()
    d.integrator = _abi.ZOH
    s = O.System.from_dynamics(d)
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (1000, 2)); u = rng.uniform(-1.5, 1.5, (1000, 1))
    xd, ud = dev(x, tdt), dev(u, tdt)
    check(_ops.simulate(d.system, xd, ud, _abi.ZOH), O.simulate(s, xd.cpu().numpy().astype(np.float64), ud.cpu().numpy().astype(np.float64),
                                                              integrator=_abi.ZOH), tol)
    from q_learning_with_hjb_amd.controller.time_optimal import DoubleIntegratorTimeOptimalController
    c = DoubleIntegratorTimeOptimalController(d)
    desc = c._descriptor()
    check(_ops.controller(d.system, desc, xd), O.controller(s, desc, xd.cpu().numpy().astype(np.float64)), tol)
    g = load_golden("di_time_optimal")
    P, V = np.meshgrid(g["pos"], g["vel"], indexing="ij")
    x0 = np.stack([P.ravel(), V.ravel()], 1)
    t_reach = c.time_to_target(x0.astype(ndt), max_time=6.0).reshape(101, 101)      # numpy in -> numpy out, one launch
    ref = O.rollout_feedback(s, desc, x0.astype(ndt).astype(np.float64), 600, integrator=_abi.ZOH, stop_at_target=True, log=False)
    agree = (np.round(t_reach / 0.01).astype(np.int64).ravel() == ref["done_step"])
    assert agree.mean() > (0.999 if prec == "f64" else 0.9)       # bang-bang switching: fp32 can flip a sign next to the curve
    err = t_reach - g["attr"]
    assert np.abs(err).mean() < 0.12 and np.abs(err).max() < 0.45 and np.median(np.abs(err)) < 0.03
    # ZOH is refused where it does not exist
    cp = make_dynamics("cartpole")
    with pytest.raises(NotImplementedError, match="LINEAR"):
        _ops.simulate(cp.system, torch.zeros((4, 4), device="cuda", dtype=tdt), torch.zeros((4, 1), device="cuda", dtype=tdt), _abi.ZOH)
