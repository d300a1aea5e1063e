"""ctypes binding of libhjbx.so (include/hjbx.h) -- the only way host code reaches the HIP kernels.

There is no CPU implementation behind this module: if the shared library is missing, or there is
no MI355X to run on, calls raise.  (The CPU oracle lives in /oracle and is test infrastructure.)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

import numpy as np

HJBX_MAX_N = 10
HJBX_MAX_M = 3

# enums of include/hjbx.h
SYS_LINEAR, SYS_CARTPOLE, SYS_ACROBOT, SYS_QUAD2D, SYS_NEARHOVER, SYS_USER = range(6)
USER_AFFINE, USER_MANIPULATOR = 0, 1
USER_MAX_PARAMS = 16
EULER, RK4, ZOH = 0, 1, 2
RESIDUAL_NORMALISED, RESIDUAL_RAW = 0, 1
CTRL_LINEAR_FEEDBACK, CTRL_CARTPOLE_ENERGY, CTRL_ACROBOT_ENERGY, CTRL_DI_TIME_OPTIMAL = 0, 1, 2, 3
LAW_QUADRATIC, LAW_BANGBANG = 0, 1
ACT_RELU, ACT_TANH, ACT_SIN = 0, 1, 2
ROLLOUT_TERMINATE = 1
ROLLOUT_STOP_AT_TARGET = 2
OPT_ROLLOUT_SCHEDULE, OPT_ROLLOUT_EXTRA_WORKGROUPS, OPT_STREAM_ROWS, OPT_MLP_ARITHMETIC, OPT_TRAIN_KERNEL = 0, 1, 2, 3, 4
OK, EINVAL, EUNSUPPORTED, EHIP, ENODEVICE = 0, -1, -2, -3, -4

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
_LIB_PATH = os.environ.get("HJBX_LIBRARY") or os.path.join(_CSRC, "libhjbx.so")   # (HJBX_LIBRARY: development builds of tools/dev)
# (source, extra flags, object): hjbx_mlp.hip is compiled once per activation (see the top of that file)
_UNITS = (("hjbx_kernels.hip", (), "hjbx_kernels.o"),
          ("hjbx_mlp.hip", ("-DHJBX_MLP_ACT=0",), "hjbx_mlp_relu.o"),
          ("hjbx_mlp.hip", ("-DHJBX_MLP_ACT=1",), "hjbx_mlp_tanh.o"),
          ("hjbx_mlp.hip", ("-DHJBX_MLP_ACT=2",), "hjbx_mlp_x3.o"),
          ("hjbx_mlp.hip", ("-DHJBX_MLP_ACT=3", "-fno-slp-vectorize"), "hjbx_mlp_h2.o"),
          ("hjbx_mlp.hip", ("-DHJBX_MLP_ACT=4",), "hjbx_mlp_sin.o"),
          ("hjbx_train.hip", ("-fno-slp-vectorize",), "hjbx_train.o"),
          ("hjbx_train_coop.hip", ("-fno-slp-vectorize",), "hjbx_train_coop.o"),
          ("hjbx_fit.hip", (), "hjbx_fit.o"),
          ("hjbx_user.hip", (f'-DHJBX_CSRC_DIR="{_CSRC}"',), "hjbx_user.o"))       # embeds three headers as text for hiprtc (.incbin)
_SOURCES = tuple(dict.fromkeys(u[0] for u in _UNITS))
_HEADERS = ("hjbx_systems.hpp", "hjbx_internal.hpp", "hjbx_host.hpp", "hjbx_mlp_core.hpp", "hjbx_mlp_x3.hpp", "hjbx_mlp_h2.hpp", "hjbx_stream_kernels.hpp",
            "hjbx_user_kernels.hpp", os.path.join("..", "..", "include", "hjbx.h"))


class HjbxAdamState(C.Structure):
    """struct hjbx_adam_state"""
    _fields_ = [("param", C.c_void_p * 3), ("exp_avg", C.c_void_p * 3), ("exp_avg_sq", C.c_void_p * 3), ("numel", C.c_int64 * 3),
                ("step", C.c_void_p * 3), ("ticket", C.c_void_p), ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double)]


class HjbxNextMinibatch(C.Structure):
    """struct hjbx_next_minibatch"""
    _fields_ = [("buf_x", C.c_void_p), ("buf_cost", C.c_void_p), ("buf_done", C.c_void_p), ("capacity", C.c_int64), ("n", C.c_int),
                ("perm", C.c_void_p), ("perm_len", C.c_int64), ("reg_table", C.c_void_p), ("table_len", C.c_int64), ("batch", C.c_int64),
                ("xs", C.c_void_p), ("costs", C.c_void_p), ("dones", C.c_void_p), ("reg_out", C.c_void_p)]


class HjbxTask(C.Structure):
    """struct hjbx_task"""
    _fields_ = [
        ("Q", C.c_double * (HJBX_MAX_N * HJBX_MAX_N)),
        ("R", C.c_double * (HJBX_MAX_M * HJBX_MAX_M)),
        ("Rinv", C.c_double * (HJBX_MAX_M * HJBX_MAX_M)),
        ("P", C.c_double * (HJBX_MAX_N * HJBX_MAX_N)),
        ("xf", C.c_double * HJBX_MAX_N),
        ("uf", C.c_double * HJBX_MAX_M),
        ("obs_min", C.c_double * HJBX_MAX_N),
        ("obs_max", C.c_double * HJBX_MAX_N),
        ("eps", C.c_double),
        ("law", C.c_int32),
        ("_pad", C.c_int32),
        ("target_r2", C.c_double),
    ]


class HjbxController(C.Structure):
    """struct hjbx_controller"""
    _fields_ = [
        ("kind", C.c_int32),
        ("wrap_error", C.c_int32),
        ("K", C.c_double * (HJBX_MAX_M * HJBX_MAX_N)),
        ("xf", C.c_double * HJBX_MAX_N),
        ("uf", C.c_double * HJBX_MAX_M),
        ("P", C.c_double * (HJBX_MAX_N * HJBX_MAX_N)),
        ("Kes", C.c_double * 3),
        ("eps_energy", C.c_double),
        ("eps_state", C.c_double),
        ("eps_region", C.c_double),
    ]


class HjbxMlp(C.Structure):
    """struct hjbx_mlp"""
    _fields_ = [
        ("W1", C.c_void_p),
        ("W2", C.c_void_p),
        ("W3", C.c_void_p),
        ("h1", C.c_int32),
        ("h2", C.c_int32),
        ("h3", C.c_int32),
        ("activation", C.c_int32),
        ("mean", C.c_double * HJBX_MAX_N),
        ("std", C.c_double * HJBX_MAX_N),
        ("xf", C.c_double * HJBX_MAX_N),
        ("eps_scalar", C.c_double),
    ]


def _fill(dst, src):
    a = np.asarray(src, dtype=np.float64).ravel()
    if a.size > len(dst):
        raise ValueError(f"descriptor field takes at most {len(dst)} values, got {a.size}")
    for i, v in enumerate(a):
        dst[i] = float(v)


def make_task(n, m, Q, R, P, xf, uf, obs_min, obs_max, eps, Rinv=None, law=0, target_r2=0.0) -> HjbxTask:
    """Pack the task part of VHJBControllerConfig (+P) into struct hjbx_task.  law = LAW_BANGBANG selects the
    time-optimal control law / unit running cost with the target ball e'e <= target_r2."""
    Q = np.asarray(Q, np.float64).reshape(n, n)
    R = np.asarray(R, np.float64).reshape(m, m)
    t = HjbxTask()
    _fill(t.Q, Q)
    _fill(t.R, R)
    _fill(t.Rinv, np.linalg.inv(R) if Rinv is None else np.asarray(Rinv, np.float64).reshape(m, m))
    _fill(t.P, np.zeros((n, n)) if P is None else np.asarray(P, np.float64).reshape(n, n))
    _fill(t.xf, np.asarray(xf, np.float64).reshape(n))
    _fill(t.uf, np.asarray(uf, np.float64).reshape(m))
    _fill(t.obs_min, np.full(n, -np.inf) if obs_min is None else np.asarray(obs_min, np.float64).reshape(n))
    _fill(t.obs_max, np.full(n, np.inf) if obs_max is None else np.asarray(obs_max, np.float64).reshape(n))
    t.eps = float(eps)
    t.law = int(law)
    t.target_r2 = float(target_r2)
    return t


def make_controller(kind, n, m, K, xf=None, uf=None, wrap_error=True, P=None, Kes=None, eps_energy=0.0,
                    eps_state=0.0, eps_region=0.0) -> HjbxController:
    c = HjbxController()
    c.kind = int(kind)
    c.wrap_error = 1 if wrap_error else 0
    _fill(c.K, np.asarray(K, np.float64).reshape(m, n))
    _fill(c.xf, np.zeros(n) if xf is None else np.asarray(xf, np.float64).reshape(n))
    _fill(c.uf, np.zeros(m) if uf is None else np.asarray(uf, np.float64).reshape(m))
    if P is not None:
        _fill(c.P, np.asarray(P, np.float64).reshape(n, n))
    if Kes is not None:
        _fill(c.Kes, np.asarray(Kes, np.float64).reshape(3))
    c.eps_energy = float(eps_energy)
    c.eps_state = float(eps_state)
    c.eps_region = float(eps_region)
    return c


# ------------------------------------------------------------------------------------------------
# build + load
# ------------------------------------------------------------------------------------------------
def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into csrc/libhjbx.so (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, s) for s in _SOURCES]
    deps = srcs + [os.path.normpath(os.path.join(_CSRC, h)) for h in _HEADERS]
    if not force and os.path.exists(_LIB_PATH):
        if all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(d) for d in deps if os.path.exists(d)):
            return _LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -ffp-contract=on: fuse a*b+c only inside one source expression, so every kernel that inlines the same
    # device function rounds identically (fused rollout == step-by-step kernels, bit for bit)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-fPIC"] + os.environ.get("HJBX_EXTRA_FLAGS", "").split()
    # the translation units compile side by side (each MFMA object holds 30 kernel instantiations, ~55 s), then link
    objs, procs = [], []
    for src, extra, objname in _UNITS:
        obj = os.path.join(_CSRC, objname)
        cmd = [hipcc] + flags + list(extra) + ["-c", os.path.join(_CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        objs.append(obj)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", _LIB_PATH] + objs
    if verbose:
        print(" ".join(link))
    subprocess.run(link, check=True)
    return _LIB_PATH


_lib = None
_lib_lock = threading.Lock()

_I64, _I32, _U32, _VP, _DBL = C.c_int64, C.c_int, C.c_uint32, C.c_void_p, C.c_double


def _typed_signatures():
    P = _VP
    return {
        "affine": [P, P, P, P, _I64, P],
        "wrap": [P, P, P, _I64, P],
        "dynamics_step": [P, P, P, P, _I64, P],
        "simulate": [P, _I32, P, P, P, _I64, P],
        "initial_state": [P, P, P, P, P, _I64, P],
        "running_cost": [P, P, P, P, P, _I64, P],
        "termination_cost": [P, P, P, P, _I64, P],
        "control_from_grad": [P, P, P, P, P, _I64, P],
        "hjb_residual": [P, P, _I32, P, P, P, P, P, P, P, _I64, P],
        "termination_residual": [_DBL, P, P, P, P, P, P, P, _I64, P],
        "vhjb_step": [P, P, _I32, _I32, _I32, P, P, P, P, P, P, P, P, _I64, P],
        "controller": [P, P, P, P, _I64, P],
        "rollout_feedback": [P, P, P, _I32, _U32, _I32, P, P, P, P, P, P, P, _I64, P],
    }


EXPORTED_SYMBOLS = (
    ["hjbx_version", "hjbx_last_error", "hjbx_device_count", "hjbx_set_option", "hjbx_system_create", "hjbx_system_create_from_source",
     "hjbx_last_compile_log", "hjbx_system_destroy", "hjbx_dims",
     "hjbx_reduce_workspace_bytes", "hjbx_rollout_workspace_bytes", "hjbx_value_grad_f32", "hjbx_vhjb_rollout_f32",
     "hjbx_value_loss_grad_workspace_bytes", "hjbx_value_loss_grad_f32", "hjbx_mix_gradients_f32", "hjbx_mix_adam_f32", "hjbx_replay_gather_f32",
     "hjbx_value_loss_adam_workspace_bytes", "hjbx_value_loss_adam_f32"]
    + [f"hjbx_{k}_{s}" for k in _typed_signatures() for s in ("f32", "f64")]
)


def lib() -> C.CDLL:
    """Load libhjbx.so (raises RuntimeError when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"hjbx: {_LIB_PATH} is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the hot path)")
        # PyTorch first: its wheel carries its own HIP runtime, and whichever libamdhip64 the process loads first serves both.  With
        # libhjbx.so loaded first (e.g. build() then smoke() in one process) torch ends up on the system runtime and hipGetDevice
        # reports no device; the other way round both use torch's copy and share its streams and allocations.
        import torch  # noqa: F401
        L = C.CDLL(_LIB_PATH)
        L.hjbx_version.restype = C.c_int
        L.hjbx_last_error.restype = C.c_size_t
        L.hjbx_last_error.argtypes = [C.c_char_p, C.c_size_t]
        L.hjbx_device_count.restype = C.c_int
        L.hjbx_reduce_workspace_bytes.restype = C.c_size_t
        L.hjbx_rollout_workspace_bytes.restype = C.c_size_t
        L.hjbx_set_option.restype = C.c_int
        L.hjbx_set_option.argtypes = [C.c_int, C.c_int]
        L.hjbx_system_create.restype = C.c_int
        L.hjbx_system_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, _VP, _VP, _VP, C.c_int, C.POINTER(_VP)]
        L.hjbx_system_create_from_source.restype = C.c_int
        L.hjbx_system_create_from_source.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_double, _VP, _VP, _VP, C.c_int, C.POINTER(_VP)]
        L.hjbx_last_compile_log.restype = C.c_size_t
        L.hjbx_last_compile_log.argtypes = [C.c_char_p, C.c_size_t]
        L.hjbx_system_destroy.restype = None
        L.hjbx_system_destroy.argtypes = [_VP]
        L.hjbx_dims.restype = C.c_int
        L.hjbx_dims.argtypes = [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.hjbx_value_grad_f32.restype = C.c_int
        L.hjbx_value_grad_f32.argtypes = [_VP, _VP, _VP, _VP, _VP, _I64, _VP]
        L.hjbx_vhjb_rollout_f32.restype = C.c_int
        L.hjbx_vhjb_rollout_f32.argtypes = [_VP, _VP, _VP, _I32, _I32, _I32, _I32, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I64, _VP, _VP]
        L.hjbx_value_loss_grad_workspace_bytes.restype = C.c_size_t
        L.hjbx_value_loss_grad_workspace_bytes.argtypes = [_I64]
        L.hjbx_value_loss_grad_f32.restype = C.c_int
        L.hjbx_value_loss_grad_f32.argtypes = [_VP, _VP, _VP, _I32, _VP, _VP, _VP, _VP, _VP, _I64, _VP]
        L.hjbx_mix_gradients_f32.restype = C.c_int
        L.hjbx_mix_gradients_f32.argtypes = [_VP, _I64, _VP, _DBL, _DBL, _VP, _VP, _VP, _VP, _VP]
        L.hjbx_mix_adam_f32.restype = C.c_int
        L.hjbx_mix_adam_f32.argtypes = [_VP, _VP, _DBL, _DBL, C.POINTER(HjbxAdamState), _VP, _VP, _VP, _VP]
        L.hjbx_value_loss_adam_workspace_bytes.restype = C.c_size_t
        L.hjbx_value_loss_adam_workspace_bytes.argtypes = [_I64]
        L.hjbx_value_loss_adam_f32.restype = C.c_int
        L.hjbx_value_loss_adam_f32.argtypes = [_VP, _VP, _VP, _I32, _VP, _VP, _VP, _VP, _DBL, _DBL, C.POINTER(HjbxAdamState), _VP, _VP,
                                               _VP, C.POINTER(HjbxNextMinibatch), _VP, _I64, _VP]
        L.hjbx_replay_gather_f32.restype = C.c_int
        L.hjbx_replay_gather_f32.argtypes = [_VP, _VP, _VP, _I64, _I32, _VP, _I64, _VP, _VP, _I64, _I64, _VP, _VP, _VP, _VP, _VP]
        for name, sig in _typed_signatures().items():
            for sfx in ("f32", "f64"):
                fn = getattr(L, f"hjbx_{name}_{sfx}")
                fn.restype = C.c_int
                fn.argtypes = sig
        _lib = L
        # HJBX_MLP_ARITHMETIC=f32|bf16x3|f16x2 in the environment selects the value-network arithmetic without code changes
        want = os.environ.get("HJBX_MLP_ARITHMETIC", "").strip().lower()
        if want:
            modes = {"f32": 0, "bf16x3": 1, "f16x2": 2}
            if want not in modes:
                raise RuntimeError(f"HJBX_MLP_ARITHMETIC must be one of {sorted(modes)}, got {want!r}")
            L.hjbx_set_option(OPT_MLP_ARITHMETIC, modes[want])
        return L


class HjbxError(RuntimeError):
    pass


def set_option(option: int, value: int) -> int:
    """hjbx_set_option: sets a process-wide knob, returns the previous value (value < 0: query)."""
    rc = lib().hjbx_set_option(int(option), int(value))
    if rc < 0 and value >= 0 or rc == EINVAL:
        check(EINVAL)
    return rc


def last_error() -> str:
    buf = C.create_string_buffer(512)
    lib().hjbx_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def compile_log() -> str:
    """The hiprtc log of the calling thread's last hjbx_system_create_from_source."""
    n = lib().hjbx_last_compile_log(None, 0)
    buf = C.create_string_buffer(n + 1)
    lib().hjbx_last_compile_log(buf, n + 1)
    return buf.value.decode("utf-8", "replace")


def check(rc: int):
    """Turn an hjbx_status into the exception the Python surface promises (Errors)."""
    if rc == OK:
        return
    msg = last_error()
    if rc == EINVAL:
        raise ValueError(f"hjbx: {msg}")
    if rc == EUNSUPPORTED:
        raise NotImplementedError(f"hjbx: {msg}")
    raise HjbxError(f"hjbx (status {rc}): {msg}")


class SystemHandle:
    """Owns an hjbx_system*; immutable after creation."""

    def __init__(self, kind, n, m, dt, umin, umax, params):
        self.kind, self.n, self.m, self.dt = int(kind), int(n), int(m), float(dt)
        self.umin = np.ascontiguousarray(umin, np.float64).reshape(m)
        self.umax = np.ascontiguousarray(umax, np.float64).reshape(m)
        self.params = np.ascontiguousarray(params, np.float64).ravel()
        h = _VP()
        check(lib().hjbx_system_create(self.kind, self.n, self.m, self.dt, self.umin.ctypes.data, self.umax.ctypes.data,
                                       self.params.ctypes.data, int(self.params.size), C.byref(h)))
        self._h = h

    @classmethod
    def from_source(cls, user_kind, device_source: str, n, m, dt, umin, umax, params):
        """hjbx_system_create_from_source: a user-defined system compiled at run time into the library's streaming kernels
        (include/hjbx.h; the snippet contract is at the top of csrc/hjbx_user_kernels.hpp).  ValueError with the compiler's log when the
        source does not compile."""
        self = cls.__new__(cls)
        self.kind, self.n, self.m, self.dt = SYS_USER, int(n), int(m), float(dt)
        self.umin = np.ascontiguousarray(umin, np.float64).reshape(m)
        self.umax = np.ascontiguousarray(umax, np.float64).reshape(m)
        self.params = np.ascontiguousarray(params, np.float64).ravel()
        self.user_kind, self.device_source = int(user_kind), str(device_source)
        h = _VP()
        rc = lib().hjbx_system_create_from_source(self.user_kind, self.device_source.encode(), self.n, self.m, self.dt, self.umin.ctypes.data,
                                                  self.umax.ctypes.data, self.params.ctypes.data if self.params.size else None,
                                                  int(self.params.size), C.byref(h))
        if rc == EINVAL and compile_log():
            raise ValueError(f"hjbx: {last_error()}\n--- compiler log ---\n{compile_log()}")
        check(rc)
        self._h = h
        return self

    @property
    def ptr(self):
        return self._h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None:
            try:
                _lib.hjbx_system_destroy(h)
            except Exception:
                pass
            self._h = None


def ref(struct):
    """pointer to a host descriptor struct (or None)"""
    return None if struct is None else C.cast(C.pointer(struct), _VP)
