"""`Dynamics` -- plugin surface #1 of the reference (dynamics/dynamics_basic.py:7-122), backed by the
gfx950 kernels of libhjbx.so.

Same attributes and method names as the reference; every array method additionally accepts a
leading batch dimension and torch tensors:

* numpy in -> numpy out (float64 stays float64 and runs the f64 kernels, like the reference's CPU
  rollout state; float32 runs the f32 kernels);
* torch CUDA tensor in -> torch CUDA tensor out, zero copies, enqueued on the current stream.

All arithmetic happens on the device.  Without an MI355X these methods raise RuntimeError.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

from .. import _abi, _ops


def _to_device(a, like_dtype=None):
    """-> (2-D CUDA tensor, was_1d, kind) where kind in {'numpy', 'cpu', 'cuda'}"""
    dev = _ops.require_device()
    if isinstance(a, torch.Tensor):
        kind = "cuda" if a.is_cuda else "cpu"
        t = a
    else:
        kind = "numpy"
        arr = np.asarray(a)
        if arr.dtype not in (np.float32, np.float64):
            arr = arr.astype(np.float64)
        t = torch.from_numpy(np.ascontiguousarray(arr))
    if t.dtype not in (torch.float32, torch.float64):
        t = t.to(torch.float64)
    if like_dtype is not None and t.dtype != like_dtype:
        t = t.to(like_dtype)
    was_1d = t.dim() == 1
    if was_1d:
        t = t.unsqueeze(0)
    t = t.to(dev).contiguous()
    return t, was_1d, kind


def _from_device(t, was_1d, kind):
    if was_1d:
        t = t.squeeze(0)
    if kind == "cuda":
        return t
    t = t.cpu()
    return t.numpy() if kind == "numpy" else t


class Dynamics:
    """x_dot = f_1(x) + f_2(x) u  with a forward-Euler `simulate` (RK4 optional)."""

    dt: float
    state_dim: int
    control_dim: int
    x0_mean: np.ndarray
    x0_std: np.ndarray
    umin: np.ndarray
    umax: np.ndarray

    _KIND = None  # hjbx_system_kind of the subclass

    def __init__(self, config) -> None:
        # dynamics_basic.py:17-26
        self.state_dim = config.state_dim
        self.control_dim = config.control_dim
        self.dt = config.dt
        self.umin = config.umin
        self.umax = config.umax
        self.x0_mean = config.x0_mean
        self.x0_std = config.x0_std
        self.seed = config.seed
        np.random.seed(config.seed)  # the reference seeds NumPy's global generator here
        self.integrator = _abi.EULER  # parity mode; set to _abi.RK4 for the 4th-order integrator
        self._sys = None
        if self._KIND is not None:
            self._sys = _abi.SystemHandle(self._KIND, self.state_dim, self.control_dim, self.dt, self.umin, self.umax,
                                          self._system_params(config))
        else:
            # a user-defined subclass (the reference lets any subclass define get_M / get_C / get_G / get_B and inherit
            # get_control_affine_matrix, dynamics_basic.py:64-94): it hands the same per-state methods over as device code
            src = self.device_source()
            if src is not None:
                kind = {"affine": _abi.USER_AFFINE, "manipulator": _abi.USER_MANIPULATOR}[src["kind"]]
                self._sys = _abi.SystemHandle.from_source(kind, src["source"], self.state_dim, self.control_dim, self.dt, self.umin, self.umax,
                                                          np.asarray(src.get("params", ()), np.float64))

    # -- subclass hooks ---------------------------------------------------------------------------
    def _system_params(self, config) -> np.ndarray:
        raise NotImplementedError

    def device_source(self):
        """Hook for user-defined subclasses: return dict(kind="manipulator" | "affine", source=<device code>, params=[...]) to get the
        library's streaming kernels compiled for this system at run time (hjbx_system_create_from_source; contract of the snippet:
        csrc/hjbx_user_kernels.hpp).  kind "manipulator": the snippet defines wrap, get_M, get_C, get_G, get_B and the generic
        manipulator form of dynamics_basic.py:64-94 is supplied; kind "affine": it defines wrap and affine (f1, f2) itself.
        Default: None -- such a subclass has no kernels and its compute methods raise NotImplementedError."""
        return None

    @property
    def system(self) -> _abi.SystemHandle:
        if self._sys is None:
            raise NotImplementedError(f"{type(self).__name__} has no device kernel: a custom Dynamics subclass gets its kernels by defining "
                                      "device_source() (see Dynamics.device_source)")
        return self._sys

    # -- reference API ----------------------------------------------------------------------------
    def get_initial_state(self, batch_size=None, *, generator=None, dtype=None):
        """wrap(U(-x0_std, x0_std) + x0_mean)  (dynamics_basic.py:28-29).

        Default: one state from NumPy's global RNG, float64 -- the reference's stream, draw for draw.
        `batch_size=B` draws B states from the same stream ((B, n) numpy).  With a torch
        `generator` the uniforms come from the device RNG and a (B, n) CUDA tensor is returned."""
        B = 1 if batch_size is None else int(batch_size)
        dev = _ops.require_device()
        if generator is not None:
            dt = dtype or torch.float32
            u01 = torch.rand((B, self.state_dim), generator=generator, device=dev, dtype=dt)
            return _ops.initial_state(self.system, self.x0_mean, self.x0_std, u01)
        u01 = np.random.uniform(size=(B, self.state_dim))
        t = torch.from_numpy(u01).to(dev)
        if dtype is not None:
            t = t.to(dtype)
        x0 = _ops.initial_state(self.system, self.x0_mean, self.x0_std, t).cpu().numpy()
        return x0[0] if batch_size is None else x0

    def get_dimension(self) -> Tuple[int, int]:
        return self.state_dim, self.control_dim

    def get_control_limit(self) -> Tuple[np.ndarray, np.ndarray]:
        return self.umin, self.umax

    def get_M(self, x):
        raise NotImplementedError

    def get_C(self, x):
        raise NotImplementedError

    def get_G(self, x):
        raise NotImplementedError

    def get_B(self):
        raise NotImplementedError

    def states_wrap(self, x):
        """x: (n,) or (B, n).  Out of place (the reference's NumPy branch wraps in place; jnp does not)."""
        t, one, kind = _to_device(x)
        self._check_state(t)
        return _from_device(_ops.wrap(self.system, t), one, kind)

    def get_control_affine_matrix(self, x):
        """-> f_1 (n,) / (B, n), f_2 (n, m) / (B, n, m)"""
        t, one, kind = _to_device(x)
        self._check_state(t)
        f1, f2 = _ops.affine(self.system, t)
        return _from_device(f1, one, kind), _from_device(f2, one, kind)

    def dynamics_step(self, x, u):
        """x_dot for (x, u); u is NOT clipped here (dynamics_basic.py:96-105)."""
        t, one, kind = _to_device(x)
        self._check_state(t)
        ut = self._control_like(u, t)
        return _from_device(_ops.dynamics_step(self.system, t, ut), one, kind)

    def simulate(self, x, u):
        """One step: u clipped to [umin, umax], integrate, wrap (dynamics_basic.py:107-122)."""
        t, one, kind = _to_device(x)
        self._check_state(t)
        ut = self._control_like(u, t)
        return _from_device(_ops.simulate(self.system, t, ut, self.integrator), one, kind)

    # -- helpers ----------------------------------------------------------------------------------
    def _check_state(self, t):
        assert t.shape[-1] == self.state_dim, f"state has {t.shape[-1]} entries, expected {self.state_dim}"

    def _control_like(self, u, x_t):
        if not isinstance(u, torch.Tensor):
            u = np.asarray(u, dtype=np.float64)
            if u.ndim == 0:
                u = u.reshape(1)
        ut, _, _ = _to_device(u, like_dtype=x_t.dtype)
        if ut.shape[0] != x_t.shape[0]:
            ut = ut.reshape(x_t.shape[0], self.control_dim)
        assert ut.shape[-1] == self.control_dim, f"control has {ut.shape[-1]} entries, expected {self.control_dim}"
        return ut.contiguous()
