"""ctypes front-end of the CPU oracle (oracle/liborc.so).  TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product
package.  It borrows the descriptor ctypes structs from the product's ABI module (they mirror
include/hjbx.h, which the oracle also compiles against; `check_layout()` verifies the sizes agree).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from q_learning_with_hjb_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liborc.so")
MAXN, MAXM = _abi.HJBX_MAX_N, _abi.HJBX_MAX_M


class OrcSystem(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n", C.c_int32), ("m", C.c_int32), ("_pad", C.c_int32), ("dt", C.c_double),
                ("umin", C.c_double * MAXM), ("umax", C.c_double * MAXM), ("p", C.c_double * (2 * (MAXN * MAXN + MAXN * MAXM)))]


class OrcMlp(C.Structure):
    _fields_ = [("h1", C.c_int32), ("h2", C.c_int32), ("h3", C.c_int32), ("activation", C.c_int32), ("mean", C.c_double * MAXN),
                ("std", C.c_double * MAXN), ("xf", C.c_double * MAXN), ("eps_scalar", C.c_double)]


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("hjbx_oracle.c", "oracle_impl.h")] + [os.path.join(_HERE, "..", "include", "hjbx.h")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.run(["make", "-C", _HERE, "-B", "liborc.so"], check=True, stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        for f in ("orc_sizeof_system", "orc_sizeof_mlp", "orc_sizeof_task", "orc_sizeof_controller"):
            getattr(_lib, f).restype = C.c_size_t
        for s in ("f32", "f64"):
            getattr(_lib, f"orc_vhjb_rollout_{s}").restype = C.c_int64
            getattr(_lib, f"orc_rollout_feedback_{s}").restype = C.c_int64
        check_layout()
    return _lib


def check_layout():
    L = _lib
    assert L.orc_sizeof_system() == C.sizeof(OrcSystem)
    assert L.orc_sizeof_mlp() == C.sizeof(OrcMlp)
    assert L.orc_sizeof_task() == C.sizeof(_abi.HjbxTask), "HjbxTask ctypes layout != struct hjbx_task"
    assert L.orc_sizeof_controller() == C.sizeof(_abi.HjbxController), "HjbxController ctypes layout != struct hjbx_controller"


def has_openmp():
    return bool(lib().orc_has_openmp())


def threads(n=0):
    """Set (n > 0) and return the number of OpenMP threads the batch loops use."""
    return int(lib().orc_threads(int(n)))


class System:
    """Host description of one of the five systems (what Dynamics.__init__ stores)."""

    def __init__(self, kind, n, m, dt, umin, umax, params):
        self.kind, self.n, self.m, self.dt = int(kind), int(n), int(m), float(dt)
        s = OrcSystem()
        s.kind, s.n, s.m, s.dt = self.kind, self.n, self.m, self.dt
        _abi._fill(s.umin, umin)
        _abi._fill(s.umax, umax)
        _abi._fill(s.p, params)
        self.c = s

    @classmethod
    def from_dynamics(cls, d):
        """Build from a product `Dynamics` object's host-side attributes (no device involved)."""
        h = d.system
        return cls(h.kind, h.n, h.m, h.dt, h.umin, h.umax, h.params)


def make_mlp(features, mean, std, xf, eps_scalar, activation=0):
    p = OrcMlp()
    p.activation = {"relu": 0, "tanh": 1, "sin": 2}.get(activation, activation)
    p.h1, p.h2, p.h3 = (int(f) for f in features)
    _abi._fill(p.mean, mean)
    _abi._fill(p.std, std)
    _abi._fill(p.xf, xf)
    p.eps_scalar = float(eps_scalar)
    return p


def _dt(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "f64", np.float64
    if dtype == np.float32:
        return "f32", np.float32
    raise TypeError(dtype)


def _a(x, dt, shape=None):
    a = np.ascontiguousarray(x, dtype=dt)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _r(s):
    return C.byref(s) if s is not None else None


def affine(sys, x, dtype=np.float64):
    sfx, dt = _dt(dtype)
    x = _a(x, dt, (-1, sys.n)); B = x.shape[0]
    f1 = np.empty((B, sys.n), dt); f2 = np.empty((B, sys.n, sys.m), dt)
    getattr(lib(), f"orc_affine_{sfx}")(_r(sys.c), _p(x), _p(f1), _p(f2), C.c_int64(B))
    return f1, f2


def wrap(sys, x, dtype=np.float64):
    sfx, dt = _dt(dtype)
    x = _a(x, dt, (-1, sys.n)); out = np.empty_like(x)
    getattr(lib(), f"orc_wrap_{sfx}")(_r(sys.c), _p(x), _p(out), C.c_int64(x.shape[0]))
    return out


def manip(sys, x, dtype=np.float64):
    sfx, dt = _dt(dtype)
    x = _a(x, dt, (-1, 4)); B = x.shape[0]
    M = np.empty((B, 2, 2), dt); Cm = np.empty((B, 2, 2), dt); G = np.empty((B, 2), dt); E = np.zeros((B,), dt)
    getattr(lib(), f"orc_manip_{sfx}")(_r(sys.c), _p(x), _p(M), _p(Cm), _p(G), _p(E), C.c_int64(B))
    return M, Cm, G, E


def dynamics_step(sys, x, u, dtype=np.float64):
    sfx, dt = _dt(dtype)
    x = _a(x, dt, (-1, sys.n)); u = _a(u, dt, (-1, sys.m)); xd = np.empty_like(x)
    getattr(lib(), f"orc_dynamics_step_{sfx}")(_r(sys.c), _p(x), _p(u), _p(xd), C.c_int64(x.shape[0]))
    return xd


def simulate(sys, x, u, integrator=0, dtype=np.float64):
    sfx, dt = _dt(dtype)
    x = _a(x, dt, (-1, sys.n)); u = _a(u, dt, (-1, sys.m)); xn = np.empty_like(x)
    getattr(lib(), f"orc_simulate_{sfx}")(_r(sys.c), C.c_int(integrator), _p(x), _p(u), _p(xn), C.c_int64(x.shape[0]))
    return xn


def initial_state(sys, mean, std, u01, dtype=np.float64):
    sfx, dt = _dt(dtype)
    u01 = _a(u01, dt, (-1, sys.n)); x0 = np.empty_like(u01)
    mean = _a(mean, np.float64); std = _a(std, np.float64)
    getattr(lib(), f"orc_initial_state_{sfx}")(_r(sys.c), _p(mean), _p(std), _p(u01), _p(x0), C.c_int64(u01.shape[0]))
    return x0


def running_cost(sys, task, x, u, dtype=np.float64):
    sfx, dt = _dt(dtype)
    x = _a(x, dt, (-1, sys.n)); u = _a(u, dt, (-1, sys.m)); c = np.empty((x.shape[0],), dt)
    getattr(lib(), f"orc_running_cost_{sfx}")(_r(sys.c), _r(task), _p(x), _p(u), _p(c), C.c_int64(x.shape[0]))
    return c


def termination_cost(sys, task, x, dtype=np.float64):
    sfx, dt = _dt(dtype)
    x = _a(x, dt, (-1, sys.n)); c = np.empty((x.shape[0],), dt)
    getattr(lib(), f"orc_termination_cost_{sfx}")(_r(sys.c), _r(task), _p(x), _p(c), C.c_int64(x.shape[0]))
    return c


def control_from_grad(sys, task, x, g, dtype=np.float64):
    sfx, dt = _dt(dtype)
    x = _a(x, dt, (-1, sys.n)); g = _a(g, dt, (-1, sys.n)); u = np.empty((x.shape[0], sys.m), dt)
    getattr(lib(), f"orc_control_from_grad_{sfx}")(_r(sys.c), _r(task), _p(x), _p(g), _p(u), C.c_int64(x.shape[0]))
    return u


def hjb_residual(sys, task, x, g, done, mode=0, dtype=np.float64):
    """-> loss_i (B,), dloss_dgrad (B,n), sums (3,) float64"""
    sfx, dt = _dt(dtype)
    x = _a(x, dt, (-1, sys.n)); g = _a(g, dt, (-1, sys.n)); done = _a(done, dt, (-1,))
    B = x.shape[0]
    li = np.empty((B,), dt); dg = np.empty((B, sys.n), dt); sums = np.zeros(3, np.float64)
    getattr(lib(), f"orc_hjb_residual_{sfx}")(_r(sys.c), _r(task), C.c_int(mode), _p(x), _p(g), _p(done), _p(li), _p(dg), _p(sums),
                                             C.c_int64(B))
    return li, dg, sums


def termination_residual(eps, V, cost, done, dtype=np.float64):
    sfx, dt = _dt(dtype)
    V = _a(V, dt, (-1,)); cost = _a(cost, dt, (-1,)); done = _a(done, dt, (-1,))
    B = V.shape[0]
    li = np.empty((B,), dt); dv = np.empty((B,), dt); sums = np.zeros(3, np.float64)
    getattr(lib(), f"orc_termination_residual_{sfx}")(C.c_double(eps), _p(V), _p(cost), _p(done), _p(li), _p(dv), _p(sums), C.c_int64(B))
    return li, dv, sums


def controller(sys, ctrl, x, dtype=np.float64):
    sfx, dt = _dt(dtype)
    x = _a(x, dt, (-1, sys.n)); u = np.empty((x.shape[0], sys.m), dt)
    getattr(lib(), f"orc_controller_{sfx}")(_r(sys.c), _r(ctrl), _p(x), _p(u), C.c_int64(x.shape[0]))
    return u


def value_grad(sys, mlp, W1, W2, W3, x, dtype=np.float64):
    sfx, dt = _dt(dtype)
    x = _a(x, dt, (-1, sys.n)); B = x.shape[0]
    W1, W2, W3 = _a(W1, dt), _a(W2, dt), _a(W3, dt)
    V = np.empty((B,), dt); g = np.empty((B, sys.n), dt)
    getattr(lib(), f"orc_value_grad_{sfx}")(_r(sys.c), _r(mlp), _p(W1), _p(W2), _p(W3), _p(x), _p(V), _p(g), C.c_int64(B))
    return V, g


def vhjb_step(sys, task, step, T_max, x, g, done_step, integrator=0, dtype=np.float64):
    """-> x_next, u, cost_t, done_t, done_step(updated copy), resid_t"""
    sfx, dt = _dt(dtype)
    x = _a(x, dt, (-1, sys.n)); g = _a(g, dt, (-1, sys.n)); B = x.shape[0]
    ds = np.ascontiguousarray(done_step, np.int32).copy()
    xn = np.empty_like(x); u = np.empty((B, sys.m), dt); c = np.empty((B,), dt); d = np.empty((B,), dt); rs = np.empty((B,), dt)
    getattr(lib(), f"orc_vhjb_step_{sfx}")(_r(sys.c), _r(task), C.c_int(integrator), C.c_int(step), C.c_int(T_max), _p(x), _p(g), _p(xn),
                                          _p(u), _p(c), _p(d), _p(ds), _p(rs), C.c_int64(B))
    return xn, u, c, d, ds, rs


def vhjb_rollout(sys, task, mlp, W1, W2, W3, x0, T_max, integrator=0, dtype=np.float64, log=True):
    """The reference's execution model: env by env, value gradient per step. -> dict + live_steps"""
    sfx, dt = _dt(dtype)
    x0 = _a(x0, dt, (-1, sys.n)); B = x0.shape[0]
    W1, W2, W3 = _a(W1, dt), _a(W2, dt), _a(W3, dt)
    traj = np.empty((T_max + 1, B, sys.n), dt) if log else None
    cost = np.empty((T_max + 1, B), dt) if log else None
    ds = np.empty((B,), np.int32)
    live = getattr(lib(), f"orc_vhjb_rollout_{sfx}")(_r(sys.c), _r(task), _r(mlp), _p(W1), _p(W2), _p(W3), C.c_int(integrator), C.c_int(T_max),
                                                    _p(x0), _p(traj), _p(cost), _p(ds), C.c_int64(B))
    return dict(traj=traj, cost=cost, done_step=ds, live_steps=int(live))


def rollout_feedback(sys, ctrl, x0, T_steps, task=None, integrator=0, terminate=False, dtype=np.float64, log=True, stop_at_target=False):
    sfx, dt = _dt(dtype)
    x0 = _a(x0, dt, (-1, sys.n)); B = x0.shape[0]
    traj = np.empty((T_steps + 1, B, sys.n), dt) if log else None
    ulog = np.empty((T_steps, B, sys.m), dt) if log else None
    cost = np.empty((T_steps + 1, B), dt) if (log and task is not None) else None
    total = np.empty((B,), dt) if task is not None else None
    ds = np.empty((B,), np.int32); xf = np.empty_like(x0)
    live = getattr(lib(), f"orc_rollout_feedback_{sfx}")(_r(sys.c), _r(task), _r(ctrl), C.c_int(integrator), C.c_uint32((1 if terminate else 0) | (2 if stop_at_target else 0)),
                                                        C.c_int(T_steps), _p(x0), _p(traj), _p(ulog), _p(cost), _p(ds), _p(total), _p(xf),
                                                        C.c_int64(B))
    return dict(traj=traj, u=ulog, cost=cost, total_cost=total, done_step=ds, x_final=xf, live_steps=int(live))
