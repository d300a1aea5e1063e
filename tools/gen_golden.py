#!/usr/bin/env python3
"""Generate golden vectors from the reference's own NumPy (CPU) code path.

Runs ONLY in the build container (needs /root/reference); the fixtures it writes under
tests/golden/ are plain data (inputs + expected outputs) and are what travels to the GPU box.

How the reference is imported: every dynamics/controller module of the
reference does `import jax`, `import jax.numpy as jnp` and (for the config dataclasses)
`import gin` at module top, but its NumPy branches (`isinstance(x, jnp.ndarray)` false) never call
into them.  jax/gin are not installed and cannot be installed offline, so two *empty placeholder*
modules are registered in sys.modules: `jax.numpy.ndarray` is a class no array is an instance of and
`gin.configurable` is the identity decorator.  They implement no arithmetic; all numbers written
below are produced by the reference's unmodified NumPy/SciPy statements.  `controller/vhjb.py`
(JAX autodiff + Flax + optax) is NOT imported -- that part of the path stays "parity unpinned"
 and is checked by known-answer identities instead.

Usage:  python tools/gen_golden.py [--out tests/golden]
"""
import argparse
import json
import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference"


def _install_placeholders():
    class _NeverArray:  # no ndarray is an instance -> reference takes its NumPy branch
        pass

    jax = types.ModuleType("jax")
    jnp = types.ModuleType("jax.numpy")
    jnp.ndarray = _NeverArray
    jrandom = types.ModuleType("jax.random")
    jrandom.PRNGKey = lambda s: s
    jax.numpy = jnp
    jax.random = jrandom
    sys.modules["jax"] = jax
    sys.modules["jax.numpy"] = jnp
    sys.modules["jax.random"] = jrandom
    gin = types.ModuleType("gin")
    gin.configurable = lambda c=None, **kw: c if c is not None else (lambda k: k)
    sys.modules["gin"] = gin
    import matplotlib
    matplotlib.use("Agg")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)

    _install_placeholders()
    sys.path.insert(0, REF)
    import numpy as np
    from configs.dynamics.dynamics_config import (LinearDynamicsConfig, CartpoleDynamicsConfig,
                                                  Quadrotors2DConfig, NearHoverQuadcopterConfig)
    from dynamics.linear import LinearDynamics
    from dynamics.cartpole import Cartpole
    from dynamics.quadrotors import Quadrotors2D, NearHoverQuadcopter
    from dynamics.acrobot import Acrobot, p as acrobot_p
    from dynamics.dynamics_basic import Dynamics
    from controller.lqr import LQR
    from controller.cartpole_energy_shaping import CartpoleEnergyShapingController
    from controller.acrobot_energy_shaping import AcrobotEnergyShapingController
    from controller.quadrotors_model_based_controller import (Quadrotors2DHoveringController,
                                                              NearHoverQuadcopterHoveringController)
    from utils.utils import solve_continuous_are

    # ---- constants copied from the .gin files (gin itself is absent) -------------------------
    GIN = {
        "linear": dict(seed=0, dt=0.02, A=[[0, 1], [0, 0]], B=[[0], [1]], umin=[-5], umax=[5],
                       x0_mean=[0, 0], x0_std=[1, 1]),
        "cartpole": dict(seed=0, mc=1, mp=0.1, l=1, g=9.81, dt=0.02, x0_mean=[0, 3.14, 0, 0],
                         x0_std=[2.4, 0.05, 1, 0.05], umin=[-10], umax=[10]),
        "quad2d": dict(seed=0, m=1, r=0.25, g=9.81, I=0.0625, dt=0.05, x0_mean=[0] * 6,
                       x0_std=[1] * 6, umin=[-20, -20], umax=[20, 20]),
        "nearhover": dict(seed=0, dt=0.05, g=9.81, m=1, kT=0.91, n0=10, umin=[0, -10, -10],
                          umax=[14.715, 10, 10], x0_mean=[0] * 10,
                          x0_std=[1, 1, 1, 0.5, 0.5, 1, 1, 1, 0.5, 0.5]),
    }

    def make(name):
        if name == "linear":
            return LinearDynamics(LinearDynamicsConfig(**GIN["linear"]))
        if name == "cartpole":
            return Cartpole(CartpoleDynamicsConfig(**GIN["cartpole"]))
        if name == "quad2d":
            return Quadrotors2D(Quadrotors2DConfig(**GIN["quad2d"]))
        if name == "nearhover":
            return NearHoverQuadcopter(NearHoverQuadcopterConfig(**GIN["nearhover"]))
        if name == "acrobot":
            # Acrobot.__init__ raises TypeError (stale constructor): set the attributes
            # of acrobot.py:23-31 by hand and use the unmodified methods.
            a = object.__new__(Acrobot)
            a.dim = 2
            a.control_dim = 1
            a.state_dim = 4
            a.p = acrobot_p
            a.m1, a.m2, a.l1, a.l2, a.I1, a.I2, a.umax_scalar = (acrobot_p[k] for k in
                                                                ("m1", "m2", "l1", "l2", "I1", "I2", "umax"))
            a.g, a.dt = acrobot_p["g"], acrobot_p["dt"]
            a.umax = acrobot_p["umax"]  # the controller reads acrobot.umax as a scalar
            return a
        raise KeyError(name)

    spot = {}
    data = {}

    # ---- spot values recorded in SURVEY.md 8c: the generator must reproduce them -------------
    lin = make("linear")
    x0 = lin.get_initial_state()
    lqr = LQR(lin, np.eye(2), np.eye(1))
    u0 = lqr.get_control_efforts(x0)
    x1 = lin.simulate(x0, u0)
    assert np.allclose(x0, [0.09762701, 0.43037873], atol=1e-8), x0
    assert np.allclose(lqr.P, [[3 ** 0.5, 1], [1, 3 ** 0.5]], atol=1e-9)
    assert np.allclose(u0, [-0.84306484], atol=1e-8) and np.allclose(x1, [0.10623458, 0.41351744], atol=1e-8)
    spot["linear"] = dict(x0=x0.tolist(), P=lqr.P.tolist(), K=lqr.K.tolist(), u0=u0.tolist(), x1=x1.tolist())
    cp = make("cartpole")
    x0 = cp.get_initial_state()
    f1, f2 = cp.get_control_affine_matrix(x0)
    assert np.allclose(x0, [0.23430483, -3.12166627, 0.20552675, 0.00448832], atol=1e-8), x0
    assert np.allclose(f1, [0.20552675, 0.00448832, 0.0195418, 0.21500285], atol=1e-7), f1
    assert np.allclose(f2.ravel(), [0, 0, 0.9999603, 0.99976178], atol=1e-7), f2
    spot["cartpole"] = dict(x0=x0.tolist(), f1=f1.tolist(), f2=f2.ravel().tolist())
    ac = make("acrobot")
    xa = np.array([.3, -.2, .5, -.7])
    assert np.allclose(ac.get_M(xa), [[15.92026631, 9.96013316], [9.96013316, 8]], atol=1e-7)
    assert np.allclose(ac.get_G(xa), [21.72454907, 3.99333667], atol=1e-7)
    assert np.allclose(ac.energy(xa), -96.65636927, atol=1e-7)

    # ---- per-system pointwise vectors ---------------------------------------------------------
    rng = np.random.default_rng(20250212)
    NPTS = 96
    for name in ("linear", "cartpole", "acrobot", "quad2d", "nearhover"):
        d = make(name)
        n = d.state_dim
        m = d.control_dim
        if name == "acrobot":
            umin, umax = -np.array([25.0]), np.array([25.0])
            # base-class simulate needs these (acrobot.py's own 3-arg call site is stale)
            d_umin, d_umax = umin, umax
        else:
            umin, umax = d.get_control_limit()
            d_umin, d_umax = umin, umax
        X = rng.uniform(-1, 1, size=(NPTS, n)) * 4.0       # well beyond +-pi on the angle slots
        X[:8] *= 3.0                                       # several periods away
        X[8] = 0.0
        U = rng.uniform(-1.5, 1.5, size=(NPTS, m)) * np.maximum(np.abs(umax), np.abs(umin))  # some beyond limits
        F1 = np.zeros((NPTS, n)); F2 = np.zeros((NPTS, n, m)); XD = np.zeros((NPTS, n)); XN = np.zeros((NPTS, n))
        XW = np.zeros((NPTS, n))
        for i in range(NPTS):
            x = X[i].copy(); u = U[i].copy()
            f1, f2 = d.get_control_affine_matrix(x)
            F1[i] = f1; F2[i] = np.asarray(f2).reshape(n, m)
            XD[i] = d.dynamics_step(x, u)
            if name == "acrobot":
                uc = np.clip(u, d_umin, d_umax)
                XN[i] = d.states_wrap(x + d.dynamics_step(x, uc) * d.dt)   # == Dynamics.simulate body
            else:
                XN[i] = d.simulate(x.copy(), u)
            XW[i] = d.states_wrap(x.copy())
        rec = dict(X=X, U=U, F1=F1, F2=F2, XDOT=XD, XNEXT=XN, XWRAP=XW, umin=np.asarray(umin, float),
                   umax=np.asarray(umax, float), dt=np.float64(d.dt))
        # seam cases for the wrap (exact multiples of pi etc.)
        seam_vals = np.array([np.pi, -np.pi, 3 * np.pi, -3 * np.pi, 2 * np.pi, -2 * np.pi, 0.0,
                              np.nextafter(np.pi, 0), np.nextafter(np.pi, 4), np.nextafter(-np.pi, 0),
                              np.nextafter(-np.pi, -4), 1e-300, -1e-300, 100.0, -100.0, 7.0, -7.0])
        XS = np.tile(seam_vals[:, None], (1, n))
        XSW = np.stack([d.states_wrap(x.copy()) for x in XS])
        rec["XSEAM"] = XS; rec["XSEAMWRAP"] = XSW
        if name in ("cartpole", "acrobot"):
            rec["M"] = np.stack([d.get_M(x) for x in X]); rec["C"] = np.stack([d.get_C(x) for x in X])
            rec["G"] = np.stack([d.get_G(x) for x in X])
        if name == "acrobot":
            rec["E"] = np.array([d.energy(x) for x in X])
        if name != "acrobot":
            # reproducible initial states: the reference seeds np.random in Dynamics.__init__
            d2 = make(name)
            rec["X0SEQ"] = np.stack([d2.get_initial_state() for _ in range(8)])
            d2 = make(name)  # the raw uniforms behind them, so a kernel can be fed the same draws
            np.random.seed(0)
            rec["U01SEQ"] = np.stack([np.random.uniform(size=(n,)) for _ in range(8)])
        data[name] = rec

    # ---- closed-loop trajectories under the model-based controllers (a20 + a1) ----------------
    def rollout(d, ctrl, x0, steps, sim=None):
        xs = [np.array(x0, float)]; us = []
        for i in range(steps):
            u = np.atleast_1d(ctrl.get_control_efforts(xs[-1]))
            us.append(np.array(u, float))
            xs.append(sim(xs[-1], u) if sim else d.simulate(xs[-1], u))
        return np.array(xs), np.array(us)

    # C1: double integrator + LQR, seed 0, T = 5 s / 0.02 (scripts/test_vhjb_policy.py:132-154)
    lin = make("linear")
    lqr = LQR(lin, np.eye(2), np.eye(1))
    xs, us = rollout(lin, lqr, lin.get_initial_state(), 249)
    data["traj_linear_lqr"] = dict(XS=xs, US=us, K=lqr.K, P=lqr.P)

    # cartpole energy shaping (controller/cartpole_energy_shaping.py:113-125, __main__ config)
    cp = make("cartpole")
    ces = CartpoleEnergyShapingController(cp)
    K, P = ces.get_lqr_term()
    A_, B_ = ces.get_linearized_dynamics()
    xs, us = rollout(cp, ces, cp.get_initial_state(), 499)
    data["traj_cartpole_es"] = dict(XS=xs, US=us, K=K, P=P, Alin=A_, Blin=B_, Kes=np.asarray(ces.K, float))
    # swing-up from hanging with a few different starts (exercises the mode switch)
    starts = np.array([[0.0, 0.1, 0.0, 0.0], [0.5, -0.4, 0.2, 0.3], [-1.0, 2.0, 0.0, -1.0], [0.1, 3.0, 0.0, 0.2]])
    XS, US = [], []
    for s in starts:
        xs, us = rollout(cp, ces, s, 400)
        XS.append(xs); US.append(us)
    data["traj_cartpole_es_multi"] = dict(X0=starts, XS=np.array(XS), US=np.array(US))
    # pointwise controller outputs (both branches)
    Xp = data["cartpole"]["X"].copy()
    Xp[::3, 1] = np.pi + 0.05 * rng.standard_normal(Xp[::3, 1].shape)      # near upright -> LQR branch
    Xp[::3, 3] = 0.1 * rng.standard_normal(Xp[::3, 3].shape)
    data["ctrl_cartpole_es"] = dict(X=Xp, U=np.stack([np.atleast_1d(ces.get_control_efforts(x)) for x in Xp]))

    # acrobot energy shaping (controller/acrobot_energy_shaping.py:123-135): x0=[0.001,0,0,0], 25 s
    ac = make("acrobot")
    aes = AcrobotEnergyShapingController(ac)
    K, P = aes.get_lqr_term()
    A_, B_ = aes.get_linearized_dynamics()
    a_umin, a_umax = -np.array([25.0]), np.array([25.0])
    def ac_sim(x, u):
        uc = np.clip(u, a_umin, a_umax)
        return ac.states_wrap(x + ac.dynamics_step(x, uc) * ac.dt)
    xs, us = rollout(ac, aes, np.array([0.001, 0, 0, 0]), 499, sim=ac_sim)
    data["traj_acrobot_es"] = dict(XS=xs, US=us, K=K, P=P, Alin=A_, Blin=B_, Kes=np.asarray(aes.K, float),
                                   Exf=np.float64(ac.energy(aes.xf)))
    Xp = data["acrobot"]["X"].copy()
    Xp[::3, 0] = np.pi + 0.05 * rng.standard_normal(Xp[::3, 0].shape)
    Xp[::3, 1:] = 0.05 * rng.standard_normal(Xp[::3, 1:].shape)
    data["ctrl_acrobot_es"] = dict(X=Xp, U=np.stack([np.atleast_1d(aes.get_control_efforts(x)) for x in Xp]))

    # Quadrotors2D hover LQR (quadrotors_model_based_controller.py:7-38), gin task Q=I6 R=I2
    qd = make("quad2d")
    qc = Quadrotors2DHoveringController(qd, np.zeros(6), np.eye(6), np.eye(2))
    xs, us = rollout(qd, qc, qd.get_initial_state(), 199)
    data["traj_quad2d_hover"] = dict(XS=xs, US=us, K=qc.K, P=qc.P, A=qc.A, B=qc.B, uf=qc.uf)
    data["ctrl_quad2d_hover"] = dict(X=data["quad2d"]["X"],
                                     U=np.stack([qc.get_control_efforts(x.copy()) for x in data["quad2d"]["X"]]))

    # NearHover hover LQR (:40-75; __main__ :296-312: xf = x0_mean, Q=I10, R=I3, 5 s)
    nh = make("nearhover")
    nc = NearHoverQuadcopterHoveringController(nh, nh.x0_mean, np.eye(10), np.eye(3))
    xs, us = rollout(nh, nc, nh.get_initial_state(), 99)
    data["traj_nearhover_hover"] = dict(XS=xs, US=us, K=nc.K, P=nc.P, A=nc.A, B=nc.B, uf=nc.uf)
    data["ctrl_nearhover_hover"] = dict(X=data["nearhover"]["X"] * 0.25,
                                        U=np.stack([nc.get_control_efforts(x.copy()) for x in data["nearhover"]["X"] * 0.25]))

    # ---- CARE by ordered Schur (utils/utils.py:30-80) on the linearisations above -------------
    care = {}
    for tag, (A, B, Q, R) in {
        "linear": (np.array(GIN["linear"]["A"], float), np.array(GIN["linear"]["B"], float), np.eye(2), np.eye(1)),
        "cartpole": (data["traj_cartpole_es"]["Alin"], data["traj_cartpole_es"]["Blin"], np.eye(4), np.eye(1)),
        "acrobot": (data["traj_acrobot_es"]["Alin"], data["traj_acrobot_es"]["Blin"], np.eye(4), np.eye(1)),
        "quad2d": (qc.A, qc.B, np.eye(6), np.eye(2)),
        "nearhover": (nc.A, nc.B, np.eye(10), np.eye(3)),
        "eye2": (np.eye(2), np.eye(2), np.eye(2), np.eye(2)),
    }.items():
        care[tag + "_A"] = A; care[tag + "_B"] = B; care[tag + "_Q"] = Q; care[tag + "_R"] = R
        care[tag + "_P"] = solve_continuous_are(A, B, Q, R)
    assert np.allclose(care["eye2_P"], 2.41421356 * np.eye(2), atol=1e-7)   # notebook cell 4 anchor
    data["care"] = care

    # ---- time-optimal double integrator: the reference's own ground-truth data file ---------------------------
    # examples/data/...level_set_methods.mat (loaded in examples/double_integrator_optimal_time.ipynb cell 18): `attr` = analytic
    # minimum time to reach the target, `mttr` = the level-set solution, on the 101 x 101 grid [-1,1]^2 indexed [position, velocity].
    # Only the numeric arrays are copied (scipy.io.loadmat parses, it executes nothing).
    import scipy.io
    mat = scipy.io.loadmat(os.path.join(REF, "examples", "data",
                                        "time_optimal_control_for_double_integrator_results_from_level_set_methods.mat"))
    grid = mat["gridOut"]
    lo, hi, N = grid["min"][0][0].ravel(), grid["max"][0][0].ravel(), grid["N"][0][0].ravel()
    data["di_time_optimal"] = dict(attr=mat["attr"].astype(np.float64), mttr=mat["mttr"].astype(np.float64),
                                   pos=np.linspace(lo[0], hi[0], int(N[0])), vel=np.linspace(lo[1], hi[1], int(N[1])))

    for k, rec in data.items():
        np.savez(os.path.join(out, f"{k}.npz"), **{kk: np.asarray(vv) for kk, vv in rec.items()})
    with open(os.path.join(out, "spot_values.json"), "w") as f:
        json.dump(dict(spot=spot, gin=GIN, acrobot_p=acrobot_p,
                       note="generated by tools/gen_golden.py from the reference NumPy branch"), f, indent=1)
    tot = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))
    print(f"wrote {len(data)} fixtures to {out} ({tot/1024:.0f} KiB)")


if __name__ == "__main__":
    main()
