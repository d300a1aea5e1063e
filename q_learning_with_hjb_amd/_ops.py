"""Typed, tensor-level wrappers over the C ABI: torch CUDA tensors in, torch CUDA tensors out.

PyTorch is plumbing here (device memory + the current HIP stream); every computation below is one
of the hand-written gfx950 kernels in csrc/.  Inputs must be float32/float64 CUDA tensors; the
public `Dynamics` / `Controller` classes do the numpy <-> device marshalling around these.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _abi
from ._abi import check, lib, ref


def require_device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("hjbx: no HIP device visible -- the batched rollout / HJB path only runs on an MI355X "
                           "(there is deliberately no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _sfx(t: torch.Tensor) -> str:
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.float64:
        return "f64"
    raise TypeError(f"hjbx kernels take float32 or float64 tensors, got {t.dtype}")


def _chk(t: torch.Tensor, name: str, shape, dtype=None):
    if not t.is_cuda:
        raise TypeError(f"{name} must be a CUDA (HIP) tensor")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name} has dtype {t.dtype}, expected {dtype}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


_ws = {}


def _workspace(device) -> torch.Tensor:
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    w = _ws.get(key)
    if w is None:
        # zero-filled ONCE: the reducing kernels keep their arrival counters in it and leave them at zero (include/hjbx.h)
        w = torch.zeros((lib().hjbx_reduce_workspace_bytes() + 15) // 16 * 2, dtype=torch.float64, device=device)
        _ws[key] = w
    return w


_rws = {}


def _rollout_workspace(device) -> torch.Tensor:
    """Work-distribution words of the persistent rollout kernel: zero-filled once, left zeroed by every launch; one per stream."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    w = _rws.get(key)
    if w is None:
        w = torch.zeros((lib().hjbx_rollout_workspace_bytes() + 15) // 16 * 4, dtype=torch.int32, device=device)
        _rws[key] = w
    return w


def _fn(name, t):
    return getattr(lib(), f"hjbx_{name}_{_sfx(t)}")


def affine(sys, x):
    B = x.shape[0]
    _chk(x, "x", (B, sys.n))
    f1 = torch.empty_like(x)
    f2 = torch.empty((B, sys.n, sys.m), dtype=x.dtype, device=x.device)
    check(_fn("affine", x)(sys.ptr, _p(x), _p(f1), _p(f2), B, _stream()))
    return f1, f2


def wrap(sys, x, out=None):
    B = x.shape[0]
    _chk(x, "x", (B, sys.n))
    out = torch.empty_like(x) if out is None else out
    _chk(out, "out", (B, sys.n), x.dtype)
    check(_fn("wrap", x)(sys.ptr, _p(x), _p(out), B, _stream()))
    return out


def dynamics_step(sys, x, u):
    B = x.shape[0]
    _chk(x, "x", (B, sys.n))
    _chk(u, "u", (B, sys.m), x.dtype)
    xd = torch.empty_like(x)
    check(_fn("dynamics_step", x)(sys.ptr, _p(x), _p(u), _p(xd), B, _stream()))
    return xd


def simulate(sys, x, u, integrator=_abi.EULER, out=None):
    B = x.shape[0]
    _chk(x, "x", (B, sys.n))
    _chk(u, "u", (B, sys.m), x.dtype)
    out = torch.empty_like(x) if out is None else out
    _chk(out, "out", (B, sys.n), x.dtype)
    check(_fn("simulate", x)(sys.ptr, int(integrator), _p(x), _p(u), _p(out), B, _stream()))
    return out


def initial_state(sys, x0_mean, x0_std, u01):
    import numpy as np
    B = u01.shape[0]
    _chk(u01, "u01", (B, sys.n))
    mean = np.ascontiguousarray(x0_mean, np.float64).reshape(sys.n)
    std = np.ascontiguousarray(x0_std, np.float64).reshape(sys.n)
    x0 = torch.empty_like(u01)
    check(_fn("initial_state", u01)(sys.ptr, mean.ctypes.data, std.ctypes.data, _p(u01), _p(x0), B, _stream()))
    return x0


def running_cost(sys, task, x, u):
    B = x.shape[0]
    _chk(x, "x", (B, sys.n))
    _chk(u, "u", (B, sys.m), x.dtype)
    c = torch.empty((B,), dtype=x.dtype, device=x.device)
    check(_fn("running_cost", x)(sys.ptr, ref(task), _p(x), _p(u), _p(c), B, _stream()))
    return c


def termination_cost(sys, task, x):
    B = x.shape[0]
    _chk(x, "x", (B, sys.n))
    c = torch.empty((B,), dtype=x.dtype, device=x.device)
    check(_fn("termination_cost", x)(sys.ptr, ref(task), _p(x), _p(c), B, _stream()))
    return c


def control_from_grad(sys, task, x, grad_v):
    B = x.shape[0]
    _chk(x, "x", (B, sys.n))
    _chk(grad_v, "grad_v", (B, sys.n), x.dtype)
    u = torch.empty((B, sys.m), dtype=x.dtype, device=x.device)
    check(_fn("control_from_grad", x)(sys.ptr, ref(task), _p(x), _p(grad_v), _p(u), B, _stream()))
    return u


def hjb_residual(sys, task, x, grad_v, done, mode=_abi.RESIDUAL_NORMALISED, want_loss=True, want_grad=True, want_sums=True):
    """-> (loss_i (B,) | None, dloss_dgrad (B,n) | None, sums (3,) | None)"""
    B = x.shape[0]
    _chk(x, "x", (B, sys.n))
    _chk(grad_v, "grad_v", (B, sys.n), x.dtype)
    _chk(done, "done", (B,), x.dtype)
    li = torch.empty((B,), dtype=x.dtype, device=x.device) if want_loss else None
    dg = torch.empty_like(x) if want_grad else None
    sums = torch.empty((3,), dtype=x.dtype, device=x.device) if want_sums else None
    ws = _workspace(x.device) if want_sums else None
    check(_fn("hjb_residual", x)(sys.ptr, ref(task), int(mode), _p(x), _p(grad_v), _p(done), _p(li), _p(dg), _p(sums),
                                 _p(ws), B, _stream()))
    return li, dg, sums


def termination_residual(eps, V, cost, done, want_loss=True, want_grad=True, want_sums=True):
    B = V.shape[0]
    _chk(V, "V", (B,))
    _chk(cost, "cost", (B,), V.dtype)
    _chk(done, "done", (B,), V.dtype)
    li = torch.empty_like(V) if want_loss else None
    dv = torch.empty_like(V) if want_grad else None
    sums = torch.empty((3,), dtype=V.dtype, device=V.device) if want_sums else None
    ws = _workspace(V.device) if want_sums else None
    check(_fn("termination_residual", V)(float(eps), _p(V), _p(cost), _p(done), _p(li), _p(dv), _p(sums), _p(ws), B, _stream()))
    return li, dv, sums


def vhjb_step(sys, task, t, T_max, x, grad_v, x_next, cost_t, done_t, done_step, u_out=None, integrator=_abi.EULER, resid_t=None):
    B = x.shape[0]
    _chk(x, "x", (B, sys.n))
    _chk(grad_v, "grad_v", (B, sys.n), x.dtype)
    _chk(x_next, "x_next", (B, sys.n), x.dtype)
    _chk(cost_t, "cost_t", (B,), x.dtype)
    _chk(done_t, "done_t", (B,), x.dtype)
    _chk(done_step, "done_step", (B,), torch.int32)
    if u_out is not None:
        _chk(u_out, "u_out", (B, sys.m), x.dtype)
    if resid_t is not None:
        _chk(resid_t, "resid_t", (B,), x.dtype)
    check(_fn("vhjb_step", x)(sys.ptr, ref(task), int(integrator), int(t), int(T_max), _p(x), _p(grad_v), _p(x_next), _p(u_out),
                              _p(cost_t), _p(done_t), _p(done_step), _p(resid_t), B, _stream()))


def controller(sys, ctrl, x):
    B = x.shape[0]
    _chk(x, "x", (B, sys.n))
    u = torch.empty((B, sys.m), dtype=x.dtype, device=x.device)
    check(_fn("controller", x)(sys.ptr, ref(ctrl), _p(x), _p(u), B, _stream()))
    return u


def rollout_feedback(sys, ctrl, x0, T_steps, task=None, integrator=_abi.EULER, terminate=False, log_traj=True, log_u=False,
                     log_cost=False, stop_at_target=False):
    """Whole closed loop in one kernel launch.  Returns a dict of device tensors (time-major)."""
    B = x0.shape[0]
    _chk(x0, "x0", (B, sys.n))
    dt, dev = x0.dtype, x0.device
    traj = torch.empty((T_steps + 1, B, sys.n), dtype=dt, device=dev) if log_traj else None
    ulog = torch.empty((T_steps, B, sys.m), dtype=dt, device=dev) if log_u else None
    cost = torch.empty((T_steps + 1, B), dtype=dt, device=dev) if (log_cost and task is not None) else None
    total = torch.empty((B,), dtype=dt, device=dev) if task is not None else None
    done_step = torch.empty((B,), dtype=torch.int32, device=dev)
    x_final = torch.empty_like(x0)
    flags = (_abi.ROLLOUT_TERMINATE if terminate else 0) | (_abi.ROLLOUT_STOP_AT_TARGET if stop_at_target else 0)
    check(_fn("rollout_feedback", x0)(sys.ptr, ref(task), ref(ctrl), int(integrator), flags, int(T_steps), _p(x0), _p(traj),
                                      _p(ulog), _p(cost), _p(done_step), _p(total), _p(x_final), B, _stream()))
    return dict(traj=traj, u=ulog, cost=cost, done_step=done_step, total_cost=total, x_final=x_final)


def value_grad(sys, mlp_desc, x, want_v=True, want_grad=True):
    """Fused MFMA value network forward + input gradient (f32 only)."""
    B = x.shape[0]
    _chk(x, "x", (B, sys.n), torch.float32)
    V = torch.empty((B,), dtype=x.dtype, device=x.device) if want_v else None
    g = torch.empty_like(x) if want_grad else None
    check(lib().hjbx_value_grad_f32(sys.ptr, ref(mlp_desc), _p(x), _p(V), _p(g), B, _stream()))
    return V, g


def vhjb_rollout(sys, task, mlp_desc, x, n_steps, T_max, done_step, t_first=0, integrator=_abi.EULER, log_traj=True, log_u=False,
                 log_residual=False, want_x_out=False, env_order=None, out=None):
    """`n_steps` closed-loop VHJB steps (value gradient + step) in ONE kernel launch (f32).  `done_step` (B,) int32 is
    updated in place.  Returns a dict of time-major device tensors: traj (n_steps+1,B,n) | None, cost, done
    (n_steps,B), u (n_steps,B,m) | None, residual (n_steps,B) | None, x_out (B,n) | None.
    `env_order` (B,) int32: permutation packing the environments into the kernel's tiles (live ones first = compaction).
    `out`: dict of preallocated contiguous slabs to write into (keys traj / u / cost / done / residual, e.g. time slices of a
    whole-horizon log) instead of fresh tensors."""
    B = x.shape[0]
    _chk(x, "x", (B, sys.n), torch.float32)
    _chk(done_step, "done_step", (B,), torch.int32)
    dev = x.device
    out = out or {}

    def slab(key, shape, wanted):
        t = out.get(key)
        if t is not None:
            _chk(t, key, shape, torch.float32)
            return t
        return torch.empty(shape, dtype=torch.float32, device=dev) if wanted else None

    traj = slab("traj", (n_steps + 1, B, sys.n), log_traj)
    ulog = slab("u", (n_steps, B, sys.m), log_u)
    cost = slab("cost", (n_steps, B), True)
    done = slab("done", (n_steps, B), True)
    resid = slab("residual", (n_steps, B), log_residual)
    x_out = torch.empty_like(x) if want_x_out else None
    if env_order is not None:
        _chk(env_order, "env_order", (B,), torch.int32)
    check(lib().hjbx_vhjb_rollout_f32(sys.ptr, ref(task), ref(mlp_desc), int(integrator), int(t_first), int(n_steps), int(T_max), _p(x),
                                      _p(traj), _p(ulog), _p(cost), _p(done), _p(resid), _p(done_step), _p(x_out), _p(env_order), B,
                                      _p(_rollout_workspace(dev)), _stream()))
    return dict(traj=traj, u=ulog, cost=cost, done=done, residual=resid, x_out=x_out)


_tws = {}


def value_loss_grad(sys, task, mlp_desc, x, cost, done, mode=_abi.RESIDUAL_NORMALISED, out=None):
    """Fused parameter gradient of the value-learning step (f32, relu): -> flat (2P + 4,) =
    [d sum(hjb)/dW1 | dW2 | dW3 | d sum(termination)/dW1 | dW2 | dW3 | sum hjb, sum termination, #interior, #done]."""
    B = x.shape[0]
    _chk(x, "x", (B, sys.n), torch.float32)
    _chk(cost, "cost", (B,), torch.float32)
    _chk(done, "done", (B,), torch.float32)
    P = sys.n * mlp_desc.h1 + mlp_desc.h1 * mlp_desc.h2 + mlp_desc.h2 * mlp_desc.h3
    flat = torch.empty((2 * P + 4,), dtype=torch.float32, device=x.device) if out is None else out
    _chk(flat, "out", (2 * P + 4,), torch.float32)
    ws = _train_workspace(x.device, lib().hjbx_value_loss_grad_workspace_bytes(B))
    check(lib().hjbx_value_loss_grad_f32(sys.ptr, ref(task), ref(mlp_desc), int(mode), _p(x), _p(cost), _p(done), _p(flat), _p(ws), B, _stream()))
    return flat


def _train_workspace(device, need):
    """The per-(device, stream) scratch of the parameter-gradient entry points: grown on demand, a much larger one given back."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    ws = _tws.get(key)
    if ws is None or ws.numel() < need or ws.numel() > 4 * need + (64 << 20):
        ws = None
        _tws.pop(key, None)
        ws = torch.empty((need + 255) // 256 * 256, dtype=torch.uint8, device=device)
        _tws[key] = ws
    return ws


def _adam_struct(params, exp_avgs, exp_avg_sqs, steps, ticket, lr, beta1, beta2, adam_eps):
    st = _abi.HjbxAdamState()
    for i, (p, m, v, k) in enumerate(zip(params, exp_avgs, exp_avg_sqs, steps)):
        for t, nm in ((p, "param"), (m, "exp_avg"), (v, "exp_avg_sq")):
            _chk(t, nm, tuple(p.shape), torch.float32)
        _chk(k, "step", (), torch.float32)
        st.param[i], st.exp_avg[i], st.exp_avg_sq[i], st.numel[i], st.step[i] = p.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), k.data_ptr()
    _chk(ticket, "ticket", (1,), torch.int32)
    st.ticket = ticket.data_ptr()
    st.lr, st.beta1, st.beta2, st.eps = float(lr), float(beta1), float(beta2), float(adam_eps)
    return st


def _reg_args(regularization):
    if torch.is_tensor(regularization):
        _chk(regularization, "regularization", (), torch.float32)
        return _p(regularization), 0.0
    return None, float(regularization)


def next_minibatch(buf_x, buf_cost, buf_done, perm, reg_table, xs, costs, dones, reg_out):
    """struct hjbx_next_minibatch for value_loss_adam: what replay_gather would be called with for the update after this one."""
    batch, n = xs.shape
    _chk(buf_x, "buf_x", (buf_x.shape[0], n), torch.float32)
    _chk(buf_cost, "buf_cost", (buf_x.shape[0],), torch.float32)
    _chk(buf_done, "buf_done", (buf_x.shape[0],), torch.float32)
    _chk(perm, "perm", (perm.shape[0],), torch.int32)
    _chk(costs, "costs", (batch,), torch.float32)
    _chk(dones, "dones", (batch,), torch.float32)
    _chk(reg_out, "reg_out", (), torch.float32)
    _chk(reg_table, "reg_table", (reg_table.shape[0],), torch.float32)
    nx = _abi.HjbxNextMinibatch()
    nx.buf_x, nx.buf_cost, nx.buf_done, nx.capacity, nx.n = buf_x.data_ptr(), buf_cost.data_ptr(), buf_done.data_ptr(), buf_x.shape[0], n
    nx.perm, nx.perm_len, nx.reg_table, nx.table_len, nx.batch = perm.data_ptr(), perm.shape[0], reg_table.data_ptr(), reg_table.shape[0], batch
    nx.xs, nx.costs, nx.dones, nx.reg_out = xs.data_ptr(), costs.data_ptr(), dones.data_ptr(), reg_out.data_ptr()
    nx._keep = (buf_x, buf_cost, buf_done, perm, reg_table, xs, costs, dones, reg_out)       # the struct holds raw pointers
    return nx


def value_loss_adam(sys, task, mlp_desc, x, cost, done, mode, regularization, eps, params, exp_avgs, exp_avg_sqs, steps, ticket, lr, beta1, beta2,
                    adam_eps, loss_accum=None, step_counter=None, next_mb=None):
    """hjbx_value_loss_adam_f32: params_update in one call (gradient, counts, mix, losses, Adam) for a single process; the flat gradient buffer is
    not materialised on the default path.  next_mb (next_minibatch(...)): the epilogue also assembles the minibatch of update step_counter + 1.
    -> losses (3,) = [total, hjb, termination]"""
    B = x.shape[0]
    _chk(x, "x", (B, sys.n), torch.float32)
    _chk(cost, "cost", (B,), torch.float32)
    _chk(done, "done", (B,), torch.float32)
    st = _adam_struct(params, exp_avgs, exp_avg_sqs, steps, ticket, lr, beta1, beta2, adam_eps)
    losses = torch.empty((3,), dtype=torch.float32, device=x.device)
    reg_dev, reg = _reg_args(regularization)
    if loss_accum is not None:
        _chk(loss_accum, "loss_accum", (3,), torch.float32)
    if step_counter is not None:
        _chk(step_counter, "step_counter", (1,), torch.int32)
    ws = _train_workspace(x.device, lib().hjbx_value_loss_adam_workspace_bytes(B))
    check(lib().hjbx_value_loss_adam_f32(sys.ptr, ref(task), ref(mlp_desc), int(mode), _p(x), _p(cost), _p(done), reg_dev, reg, float(eps), C.byref(st), _p(losses),
                                         _p(loss_accum), _p(step_counter), None if next_mb is None else C.byref(next_mb), _p(ws), B, _stream()))
    return losses


def release_workspaces(stream=None):
    """Drop the cached workspaces (reduce tickets, rollout flags, parameter-gradient scratch) of one stream handle, or of every stream
    (`stream=None`).  They are re-created -- zero-filled where the kernels need that -- on next use.  Call it when a stream goes away, or
    after a launch was aborted (a device fault, a killed kernel): the ticket / flag words are only guaranteed to be zero after launches
    that ran to completion."""
    for cache in (_ws, _rws, _tws):
        for key in [k for k in cache if stream is None or k[1] == stream]:
            del cache[key]


def mix_gradients(flat, n_params, regularization, eps, loss_accum=None, step_counter=None):
    """hjbx_mix_gradients_f32: -> (mixed (P,), losses (3,) = [total, hjb, termination]).  `regularization`: float or 0-dim float32 CUDA tensor.
    loss_accum (3,) float32: the losses are added to it on the device; step_counter (1,) int32: incremented on the device."""
    _chk(flat, "flat", (2 * n_params + 4,), torch.float32)
    mixed = torch.empty((n_params,), dtype=torch.float32, device=flat.device)
    losses = torch.empty((3,), dtype=torch.float32, device=flat.device)
    if torch.is_tensor(regularization):
        _chk(regularization, "regularization", (), torch.float32)
        reg_dev, reg = _p(regularization), 0.0
    else:
        reg_dev, reg = None, float(regularization)
    if loss_accum is not None:
        _chk(loss_accum, "loss_accum", (3,), torch.float32)
    if step_counter is not None:
        _chk(step_counter, "step_counter", (1,), torch.int32)
    check(lib().hjbx_mix_gradients_f32(_p(flat), int(n_params), reg_dev, reg, float(eps), _p(mixed), _p(losses), _p(loss_accum), _p(step_counter), _stream()))
    return mixed, losses


def mix_adam(flat, regularization, eps, params, exp_avgs, exp_avg_sqs, steps, ticket, lr, beta1, beta2, adam_eps, loss_accum=None, step_counter=None):
    """hjbx_mix_adam_f32: mix of the two gradients + one Adam step on the three weight matrices, in place, one launch.  -> losses (3,)"""
    P = sum(p.numel() for p in params)
    _chk(flat, "flat", (2 * P + 4,), torch.float32)
    st = _adam_struct(params, exp_avgs, exp_avg_sqs, steps, ticket, lr, beta1, beta2, adam_eps)
    losses = torch.empty((3,), dtype=torch.float32, device=flat.device)
    reg_dev, reg = _reg_args(regularization)
    if loss_accum is not None:
        _chk(loss_accum, "loss_accum", (3,), torch.float32)
    if step_counter is not None:
        _chk(step_counter, "step_counter", (1,), torch.int32)
    check(lib().hjbx_mix_adam_f32(_p(flat), reg_dev, reg, float(eps), C.byref(st), _p(losses), _p(loss_accum), _p(step_counter), _stream()))
    return losses


def replay_gather(buf_x, buf_cost, buf_done, perm, step_counter, reg_table, xs, costs, dones, reg_out):
    """hjbx_replay_gather_f32: minibatch number step_counter[0] (read on the device) of the epoch's permutation into xs / costs / dones, and the
    regularisation weight of that update into reg_out (0-dim float32)."""
    batch, n = xs.shape
    _chk(buf_x, "buf_x", (buf_x.shape[0], n), torch.float32)
    _chk(perm, "perm", (perm.shape[0],), torch.int32)
    _chk(step_counter, "step_counter", (1,), torch.int32)
    _chk(costs, "costs", (batch,), torch.float32)
    _chk(dones, "dones", (batch,), torch.float32)
    _chk(reg_out, "reg_out", (), torch.float32)
    _chk(reg_table, "reg_table", (reg_table.shape[0],), torch.float32)
    check(lib().hjbx_replay_gather_f32(_p(buf_x), _p(buf_cost), _p(buf_done), int(buf_x.shape[0]), int(n), _p(perm), int(perm.shape[0]), _p(step_counter),
                                       _p(reg_table), int(reg_table.shape[0]), int(batch), _p(xs), _p(costs), _p(dones), _p(reg_out), _stream()))
