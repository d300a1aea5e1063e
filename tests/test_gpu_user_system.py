"""User-defined `Dynamics` subclasses: the OPEN half of the reference's plugin surface.  The reference lets any subclass define
get_M / get_C / get_G / get_B and inherit get_control_affine_matrix (dynamics/dynamics_basic.py:64-94), or override
get_control_affine_matrix itself (dynamics/linear.py:20-22, quadrotors.py:17-46).  Here such a subclass hands the same per-state
methods over as a device-code snippet (`Dynamics.device_source`), the library compiles it at run time into its own streaming kernels
(hjbx_system_create_from_source, hiprtc), and the subclass is a first-class system: wrap / affine / dynamics_step / simulate /
vhjb_step / hjb_residual / rollout_feedback, float32 and float64, through the same C ABI.

Oracles: (1) a user-written cart-pole (other constants than the built-in one) against the oracle's generic manipulator path
(oracle/oracle_impl.h: M / C / G + explicit inverse, pinned by the reference's golden vectors); (2) a damped cart-pole -- a system the
library has never seen -- against a NumPy restatement of dynamics_basic.py:78-92 written in this file; (3) a user-written planar
quadrotor ("affine" kind) against the built-in kernels.  float64 at 1e-12 of each element's term scale.
"""
import numpy as np
import pytest
import torch

from conftest import make_vhjb_config, wrapped_diff
from oracle import oracle as O
from parity_util import check
from q_learning_with_hjb_amd import _abi, _ops
from q_learning_with_hjb_amd.configs import defaults as D
from q_learning_with_hjb_amd.dynamics.dynamics_basic import Dynamics

CARTPOLE_SRC = r"""
    // the reference's cart-pole (dynamics/cartpole.py:19-64) as a user system; p = mc, mp, l, g [, b_cart, b_pole]
    HJBX_DEV void wrap(T* x) const { x[1] = wrap_angle(x[1]); }
    HJBX_DEV void get_M(const T* x, T* Mq) const {
        T s, c; sincos_t(x[1], &s, &c);
        Mq[0] = p[0] + p[1]; Mq[1] = p[1] * p[2] * c; Mq[2] = p[1] * p[2] * c; Mq[3] = p[1] * p[2] * p[2];
    }
    HJBX_DEV void get_C(const T* x, T* Cq) const {
        T s, c; sincos_t(x[1], &s, &c);
        Cq[0] = DAMP0; Cq[1] = -p[1] * p[2] * x[3] * s; Cq[2] = T(0); Cq[3] = DAMP1;
    }
    HJBX_DEV void get_G(const T* x, T* Gq) const {
        T s, c; sincos_t(x[1], &s, &c);
        Gq[0] = T(0); Gq[1] = p[1] * p[3] * p[2] * s;
    }
    HJBX_DEV void get_B(T* Bq) const { Bq[0] = T(1); Bq[1] = T(0); }
"""

QUAD2D_SRC = r"""
    // the reference's planar quadrotor (dynamics/quadrotors.py:17-70) as an "affine" user system; p = m, r, I, g
    HJBX_DEV void wrap(T* x) const { x[2] = wrap_angle(x[2]); }
    HJBX_DEV void affine(const T* x, T* f1, T* f2) const {
        T s, c; sincos_t(x[2], &s, &c);
        f1[0] = x[3]; f1[1] = x[4]; f1[2] = x[5]; f1[3] = T(0); f1[4] = -p[3]; f1[5] = T(0);
        for (int i = 0; i < 6; ++i) f2[i] = T(0);
        f2[6] = -s / p[0]; f2[7] = -s / p[0]; f2[8] = c / p[0]; f2[9] = c / p[0]; f2[10] = p[1] / p[2]; f2[11] = -p[1] / p[2];
    }
"""


class UserCartpole(Dynamics):
    """A cart-pole the library has no built-in constants for, defined the reference's way: M, C, G, B."""

    def __init__(self, config, damping=(0.0, 0.0)):
        self.mc, self.mp, self.l, self.g = config.mc, config.mp, config.l, config.g
        self.damping = tuple(float(v) for v in damping)
        super().__init__(config)

    def device_source(self):
        damped = any(self.damping)
        src = CARTPOLE_SRC.replace("DAMP0", "p[4]" if damped else "T(0)").replace("DAMP1", "p[5]" if damped else "T(0)")
        return dict(kind="manipulator", source=src, params=[self.mc, self.mp, self.l, self.g] + (list(self.damping) if damped else []))

    # the host-side twins (what the reference's subclass would define; used by the NumPy restatement below)
    def get_M(self, x):
        c = np.cos(x[1])
        return np.array([[self.mc + self.mp, self.mp * self.l * c], [self.mp * self.l * c, self.mp * self.l ** 2]])

    def get_C(self, x):
        return np.array([[self.damping[0], -self.mp * self.l * x[3] * np.sin(x[1])], [0, self.damping[1]]])

    def get_G(self, x):
        return np.array([0, self.mp * self.g * self.l * np.sin(x[1])])

    def get_B(self):
        return np.array([1.0, 0.0])


class UserQuad2D(Dynamics):
    def __init__(self, config):
        self.cfg = config
        super().__init__(config)

    def device_source(self):
        return dict(kind="affine", source=QUAD2D_SRC, params=[self.cfg.m, self.cfg.r, self.cfg.I, self.cfg.g])


def reference_affine(d, x):
    """dynamics_basic.py:78-92, restated: f1 = [dq; -inv(M)(C dq + G)], f2 = [0; inv(M) B] for ONE state."""
    D_ = d.state_dim // 2
    dq = x[D_:]
    Mi = np.linalg.inv(d.get_M(x))
    f1 = np.hstack([dq, -Mi @ (d.get_C(x) @ dq + d.get_G(x))])
    f2 = np.vstack([np.zeros((D_, d.control_dim)), (Mi @ d.get_B()).reshape(-1, d.control_dim)])
    return f1, f2


CFG = dict(mc=2.0, mp=0.3, l=0.7, g=9.81, dt=0.02, umin=[-8], umax=[8], x0_mean=[0, 3.0, 0, 0], x0_std=[1.0, 0.6, 1.0, 1.0])


def _states(B, seed=0):
    rng = np.random.default_rng(seed)
    x = np.array([0, 3.0, 0, 0]) + rng.uniform(-1, 1, (B, 4)) * [2.0, 1.5, 3.0, 3.0]
    u = rng.uniform(-10, 10, (B, 1))
    return x, u


def test_user_system_compiles_without_a_gpu_and_reports_compile_errors():
    """CPU part (no `gpu` marker): hiprtc compiles the snippet into the library's kernels without a device; a snippet that does not compile
    raises ValueError with the compiler's log; argument errors map to ValueError."""
    d = UserCartpole(D.cartpole_dynamics_config(**CFG))
    assert d.system.kind == _abi.SYS_USER and d.get_dimension() == (4, 1)
    n, m = __import__("ctypes").c_int(), __import__("ctypes").c_int()
    assert _abi.lib().hjbx_dims(d.system.ptr, __import__("ctypes").byref(n), __import__("ctypes").byref(m)) == 0 and (n.value, m.value) == (4, 1)
    with pytest.raises(ValueError, match="compiler log"):
        _abi.SystemHandle.from_source(_abi.USER_MANIPULATOR, "HJBX_DEV void wrap(T* x) const { x[1] = no_such_function(x[1]); }", 4, 1, 0.02, [-1], [1], [1.0])
    assert "no_such_function" in _abi.compile_log()
    with pytest.raises(ValueError, match="even"):
        _abi.SystemHandle.from_source(_abi.USER_MANIPULATOR, CARTPOLE_SRC, 3, 1, 0.02, [-1], [1], [1.0])
    with pytest.raises(ValueError, match="parameters"):
        _abi.SystemHandle.from_source(_abi.USER_AFFINE, QUAD2D_SRC, 6, 2, 0.05, [-1, -1], [1, 1], np.ones(17))

    class NoKernel(Dynamics):       # a subclass without device_source keeps the documented error
        pass
    nk = NoKernel(D.cartpole_dynamics_config())
    with pytest.raises(NotImplementedError, match="device_source"):
        nk.system


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_user_cartpole_matches_the_oracles_generic_manipulator_path(prec):
    """A cart-pole written as M / C / G / B by the user (constants the built-in configuration does not have) == the oracle's generic
    manipulator path with the same constants: affine, dynamics_step, wrap, simulate (Euler, RK4), per element at 1e-12 (f64) / 1e-5 (f32)
    of the element's own term scale."""
    tdt, tol = (torch.float64, 1e-12) if prec == "f64" else (torch.float32, 1e-5)
    d = UserCartpole(D.cartpole_dynamics_config(**CFG))
    s = O.System(_abi.SYS_CARTPOLE, 4, 1, d.dt, d.umin, d.umax, [d.mc, d.mp, d.l, d.g])
    x, u = _states(5003)
    xd, ud = torch.as_tensor(x, dtype=tdt, device="cuda"), torch.as_tensor(u, dtype=tdt, device="cuda")
    xr, ur = xd.cpu().numpy().astype(np.float64), ud.cpu().numpy().astype(np.float64)
    f1, f2 = _ops.affine(d.system, xd)
    o1, o2 = O.affine(s, xr)
    row1 = np.abs(o1).max(1, keepdims=True)
    check(f1, o1, tol, row1)
    check(f2, o2, tol, np.abs(o2).reshape(len(x), -1).max(1)[:, None, None])
    S_xd = np.abs(o1) + np.einsum("bkj,bj->bk", np.abs(o2), np.abs(ur)) + row1
    check(_ops.dynamics_step(d.system, xd, ud), O.dynamics_step(s, xr, ur), tol, S_xd)
    check(_ops.wrap(d.system, xd), O.wrap(s, xr), tol, np.pi, angle_idx=[1])
    S_sim = np.abs(xr) + d.dt * S_xd
    S_sim[:, 1] += np.pi
    for integ in (_abi.EULER, _abi.RK4):
        check(_ops.simulate(d.system, xd, ud, integ), O.simulate(s, xr, ur, integ), tol, S_sim, angle_idx=[1])
    # the reference-shaped surface: numpy (n,) in, numpy out, through the same kernels
    f1s, f2s = d.get_control_affine_matrix(x[7])
    assert f1s.shape == (4,) and f2s.shape == (4, 1)
    np.testing.assert_allclose(f1s, o1[7] if prec == "f64" else O.affine(s, x[7:8])[0][0], rtol=1e-10, atol=1e-10)
    xn = d.simulate(x[7], u[7])
    assert np.abs(wrapped_diff(xn, O.simulate(s, x[7:8], u[7:8])[0], [1])).max() < 1e-10


@pytest.mark.gpu
def test_user_system_the_library_has_never_seen_matches_the_reference_formula():
    """A DAMPED cart-pole (viscous friction on cart and pole, in C): no built-in kernel, no oracle kind.  Against the NumPy restatement of
    the reference's generic manipulator form (dynamics_basic.py:78-92, np.linalg.inv) evaluated state by state with the subclass's own
    get_M / get_C / get_G / get_B, f64 at 1e-12 of the term scale; the Euler step and the wrap included."""
    d = UserCartpole(D.cartpole_dynamics_config(**CFG), damping=(0.4, 0.05))
    x, u = _states(400, seed=3)
    f1, f2 = d.get_control_affine_matrix(x)                              # numpy f64 in -> the f64 kernels
    want = [reference_affine(d, xi) for xi in x]
    w1, w2 = np.stack([w[0] for w in want]), np.stack([w[1] for w in want])
    row1 = np.abs(w1).max(1, keepdims=True)
    check(f1, w1, 1e-12, row1)
    check(f2, w2, 1e-12, np.abs(w2).reshape(len(x), -1).max(1)[:, None, None])
    uc = np.clip(u, d.umin, d.umax)
    xn = x + (w1 + np.einsum("bkj,bj->bk", w2, uc)) * d.dt               # dynamics_basic.py:118-120
    xn[:, 1] = np.remainder(xn[:, 1] + np.pi, 2 * np.pi) - np.pi         # cartpole.py:61-63
    S = np.abs(x) + d.dt * (np.abs(w1) + np.abs(w2[:, :, 0]) * np.abs(uc) + row1)
    S[:, 1] += np.pi
    check(d.simulate(x, u), xn, 1e-12, S, angle_idx=[1])
    # the damping really is in the kernels: the undamped system differs
    f1u, _ = UserCartpole(D.cartpole_dynamics_config(**CFG)).get_control_affine_matrix(x)
    assert np.abs(f1u - f1).max() > 1e-2


@pytest.mark.gpu
def test_user_affine_system_equals_the_builtin_kernels():
    """kind "affine": the planar quadrotor written by the user against the built-in Quadrotors2D kernels (f64: 1e-12; the closed loop under
    the hover LQR -- the fused rollout kernel compiled for the user system -- against the built-in fused kernel)."""
    from q_learning_with_hjb_amd.controller.quadrotors_model_based_controller import Quadrotors2DHoveringController
    from q_learning_with_hjb_amd.dynamics.quadrotors import Quadrotors2D
    cfg = D.quadrotors2d_dynamics_config()
    du, db = UserQuad2D(cfg), Quadrotors2D(cfg)
    rng = np.random.default_rng(1)
    x = rng.uniform(-1.5, 1.5, (3001, 6)); u = rng.uniform(-25, 25, (3001, 2))
    xd, ud = torch.as_tensor(x, device="cuda"), torch.as_tensor(u, device="cuda")
    for a, b in zip(_ops.affine(du.system, xd), _ops.affine(db.system, xd)):
        check(a, b.cpu().numpy(), 1e-12, 1.0)
    for integ in (_abi.EULER, _abi.RK4):
        check(_ops.simulate(du.system, xd, ud, integ), _ops.simulate(db.system, xd, ud, integ).cpu().numpy(), 1e-12, np.abs(x) + np.pi, angle_idx=[2])
    c = Quadrotors2DHoveringController(db, np.zeros(6), np.eye(6), np.eye(2))
    task = _abi.make_task(6, 2, np.eye(6), np.eye(2), np.eye(6), np.zeros(6), c.uf, [-2, -2, -1.5, -5, -5, -2], [2, 2, 1.5, 5, 5, 2], 1e-10)
    x0 = torch.as_tensor(rng.uniform(-0.8, 0.8, (2000, 6)), device="cuda")
    a = _ops.rollout_feedback(du.system, c._descriptor(), x0, 60, task=task, terminate=True, log_u=True, log_cost=True)
    b = _ops.rollout_feedback(db.system, c._descriptor(), x0, 60, task=task, terminate=True, log_u=True, log_cost=True)
    assert torch.equal(a["done_step"], b["done_step"])
    S = np.abs(b["traj"].cpu().numpy()).max(axis=(0, 2))[None, :, None] + np.pi
    check(a["traj"], b["traj"].cpu().numpy(), 1e-9, S, angle_idx=[2])
    check(a["total_cost"], b["total_cost"].cpu().numpy(), 1e-9, 0.05)


@pytest.mark.gpu
def test_user_system_vhjb_kernels_and_fused_rollout_properties():
    """The closed-loop side for a user system: vhjb_step and hjb_residual against the oracle (the user cart-pole == the oracle's cart-pole
    kind with the same constants), the fused T-step feedback rollout bit-identical to controller + simulate applied step by step (f32,
    B = 2^16: the size-independent property), and the MFMA entry points refusing a user handle."""
    d = UserCartpole(D.cartpole_dynamics_config(**CFG))
    s = O.System(_abi.SYS_CARTPOLE, 4, 1, d.dt, d.umin, d.umax, [d.mc, d.mp, d.l, d.g])
    cfg = make_vhjb_config("cartpole")
    task = _abi.make_task(4, 1, cfg.Q, cfg.R, np.eye(4) * 3.0, cfg.xf, cfg.uf, cfg.obs_min, cfg.obs_max, cfg.epsilon)
    rng = np.random.default_rng(5)
    B = 4001
    x = np.asarray(cfg.xf, np.float64) + rng.uniform(-1.05, 1.05, (B, 4)) * [4.8, 0.418, 3, 3]
    g = rng.standard_normal((B, 4)) * 5
    xd, gd = torch.as_tensor(x, device="cuda"), torch.as_tensor(g, device="cuda")
    ds = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    xn = torch.empty_like(xd); c = torch.empty(B, dtype=torch.float64, device="cuda"); dn = torch.empty_like(c); uo = torch.empty((B, 1), dtype=torch.float64, device="cuda")
    rs = torch.empty_like(c)
    for integ in (_abi.EULER, _abi.RK4):
        ds.fill_(-1)
        _ops.vhjb_step(d.system, task, 3, 200, xd, gd, xn, c, dn, ds, u_out=uo, integrator=integ, resid_t=rs)
        oxn, ou, oc, od, ods, ors = O.vhjb_step(s, task, 3, 200, x, g, np.full(B, -1, np.int32), integrator=integ)
        assert np.array_equal(ds.cpu().numpy(), ods) and np.array_equal(dn.cpu().numpy(), od) and 0 < (ods >= 0).sum() < B
        assert np.abs(wrapped_diff(xn.cpu().numpy(), oxn, [1])).max() < 1e-10 and np.abs(uo.cpu().numpy() - ou).max() < 1e-10
        assert np.abs(c.cpu().numpy() - oc).max() < 1e-9 * (1 + np.abs(oc).max()) and np.abs(rs.cpu().numpy() - ors).max() < 1e-8 * (1 + np.abs(ors).max())
    done = (rng.uniform(size=B) < 0.3).astype(np.float64)
    for mode in (_abi.RESIDUAL_NORMALISED, _abi.RESIDUAL_RAW):
        li, dg, sums = _ops.hjb_residual(d.system, task, xd, gd, torch.as_tensor(done, device="cuda"), mode)
        oli, odg, osums = O.hjb_residual(s, task, x, g, done, mode)
        assert np.abs(li.cpu().numpy() - oli).max() < 1e-9 * (1 + np.abs(oli).max()) and np.abs(dg.cpu().numpy() - odg).max() < 1e-9 * (1 + np.abs(odg).max())
        np.testing.assert_allclose(sums.cpu().numpy(), osums, rtol=1e-10)
    # fused rollout == step by step, bitwise (f32, 2^16 environments), under an LQR for the user system's own linearisation
    from q_learning_with_hjb_amd.utils.utils import linearize, solve_continuous_are
    xf = np.array([0, np.pi, 0, 0.0])
    A, Bm = linearize(d, xf, [0.0])
    P = solve_continuous_are(A, Bm, np.eye(4), np.eye(1))
    K = Bm.T @ P
    assert np.linalg.eigvals(A - Bm @ K).real.max() < 0                       # the linearisation of the USER system is stabilisable and stabilised
    ctrl = _abi.make_controller(_abi.CTRL_LINEAR_FEEDBACK, 4, 1, K, xf=xf)
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    x0 = d.get_initial_state(1 << 16, generator=gen)
    T = 25
    out = _ops.rollout_feedback(d.system, ctrl, x0, T, log_u=True)
    xx = x0.clone()
    for t in range(T):
        assert torch.equal(out["traj"][t], xx)
        uu = _ops.controller(d.system, ctrl, xx)
        assert torch.equal(out["u"][t], uu)
        xx = _ops.simulate(d.system, xx, uu)
    assert torch.equal(out["traj"][T], xx)
    # matrix-core entry points: built-in systems only
    from q_learning_with_hjb_amd.controller.vhjb import VHJBController
    ctl = VHJBController(d, make_vhjb_config("cartpole"))
    assert not ctl.fused_value_grad and not ctl.fused_param_grad
    with pytest.raises(NotImplementedError):
        _ops.value_grad(d.system, ctl.value_function_approximator.descriptor(), x0[:64].contiguous())
    # ... and the learner runs on the user system through the PyTorch network + the run-time compiled step / residual kernels
    ctl.epochs, ctl.num_of_trajectories_per_epoch = 2, 16
    lists = ctl.train()
    assert len(lists) == 6 and len(lists[0]) == 2 and all(np.isfinite(v) for v in lists[0])
