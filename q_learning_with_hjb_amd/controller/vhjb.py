"""VHJBController -- value-function learning with the HJB residual (reference controller/vhjb.py).

Split of labour (BASELINE.json north_star):
  * HIP kernels (libhjbx.so): control law, running/terminal cost, the closed-loop rollout step, the HJB
    and termination residuals with their analytic gradients, angle wrap, initial states, and -- for
    float32 -- the value network forward + input gradient fused on the matrix cores.
  * PyTorch-ROCm: the value-network parameters, their gradients (autograd through a few matmuls) and
    Adam; torch.distributed (RCCL) for ONE flat all-reduce per optimiser step.

Rollouts are batched: B independent copies of the reference's batch-1 loop (vhjb.py:171-193) advance
in lock-step, state resident in HBM, time-major logs `(T+1, B, ...)`.
"""
from __future__ import annotations

import math
import weakref
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _abi, _ops
from ..dynamics.dynamics_basic import Dynamics, _from_device, _to_device
from ..utils.utils import linearize, solve_continuous_are
from .controller_basic import Controller


# ------------------------------------------------------------------------------------------------
# value network
# ------------------------------------------------------------------------------------------------
def lecun_normal_(w: torch.Tensor, generator=None):
    """Flax's default Dense init: truncated normal (+-2 sigma), variance 1/fan_in.
    `w` has the Flax kernel layout (in, out)."""
    fan_in = w.shape[0]
    std = math.sqrt(1.0 / fan_in) / 0.87962566103423978
    with torch.no_grad():
        torch.nn.init.trunc_normal_(w, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=generator)
        w.mul_(std)
    return w


class ValueFunctionApproximator(torch.nn.Module):
    """V(x) = ||MLP((e - mean)/std)||^2 + eps_s ||e||^2,  e = wrap(x - xf); bias-free Dense layers
    with ReLU between them so that V(xf) = 0 (reference vhjb.py:17-60).  Kernels are stored
    (in, out) and applied as `x @ W`, like Flax.  `activation` "sin" / "tanh" give the notebook variants
    (examples/double_integrator_optimal_time.ipynb cell 5, examples/cartpole_balancing.ipynb): odd functions
    with act(0) = 0, so V(xf) = 0 still holds; the fused MFMA kernels exist for all three."""

    FUSED_ACTIVATIONS = ("relu", "tanh", "sin")   # hjbx_value_grad_f32 / hjbx_vhjb_rollout_f32 exist for these
    _ACT = {"relu": (torch.relu, lambda a: (a > 0).to(a.dtype)),
            "sin": (torch.sin, torch.cos),
            "tanh": (torch.tanh, lambda a: 1.0 - torch.tanh(a) ** 2)}

    def __init__(self, dynamics: Dynamics, features: Sequence[int], mean, std, xf, epsilon_scalar: float,
                 using_batch_norm: bool = False, dtype=torch.float32, device=None, generator=None, activation: str = "relu"):
        super().__init__()
        if activation not in self._ACT:
            raise ValueError(f"activation must be one of {sorted(self._ACT)}, got {activation!r}")
        self.activation = activation
        if using_batch_norm:
            raise NotImplementedError("BatchNorm is disabled in every reference config and is not implemented")
        if len(features) != 3:
            raise NotImplementedError("the fused kernel and this module take exactly three Dense layers")
        self.dynamics = dynamics
        self.features = tuple(int(f) for f in features)
        n = dynamics.state_dim
        dims = (n,) + self.features
        self.weights = torch.nn.ParameterList()
        for i in range(3):
            w = torch.empty((dims[i], dims[i + 1]), dtype=dtype, device=device)
            self.weights.append(torch.nn.Parameter(lecun_normal_(w, generator)))
        self.register_buffer("mean", torch.as_tensor(np.asarray(mean, np.float64), dtype=dtype, device=device))
        self.register_buffer("std", torch.as_tensor(np.asarray(std, np.float64), dtype=dtype, device=device))
        self.register_buffer("xf", torch.as_tensor(np.asarray(xf, np.float64), dtype=dtype, device=device))
        self.epsilon_scalar = float(epsilon_scalar)
        self._np = dict(mean=np.asarray(mean, np.float64), std=np.asarray(std, np.float64), xf=np.asarray(xf, np.float64))

    # error coordinates are data: nothing is differentiated through the wrap
    def error_coords(self, x: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            return _ops.wrap(self.dynamics.system, (x - self.xf).contiguous())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        V, _ = self.value_and_grad(x, want_grad=False)
        return V

    def value_and_grad(self, x: torch.Tensor, want_grad: bool = True, weights=None):
        """V (B,) and dV/dx (B, n) as differentiable functions of the weights (`weights`: stand-ins for the three
        parameters, e.g. detached views that act as fresh autograd leaves).

        The input gradient is written out as reverse mode by hand (three transposed matmuls with the
        ReLU masks), which keeps d(grad)/d(weights) a plain first-order autograd graph -- the
        "double back-prop" of vhjb.py:282 without create_graph."""
        W1, W2, W3 = self.weights if weights is None else weights
        e = self.error_coords(x)
        z = (e - self.mean) / self.std
        act, dact = self._ACT[self.activation]
        a1 = z @ W1
        h1 = act(a1)
        a2 = h1 @ W2
        h2 = act(a2)
        y = h2 @ W3
        V = (y * y).sum(-1) + self.epsilon_scalar * (e * e).sum(-1)
        if not want_grad:
            return V, None
        d2 = ((2.0 * y) @ W3.t()) * dact(a2)
        d1 = (d2 @ W2.t()) * dact(a1)
        g = (d1 @ W1.t()) / self.std + (2.0 * self.epsilon_scalar) * e
        return V, g

    @torch.no_grad()
    def load_quadratic(self, P, noise: float = 0.0, generator=None):
        """Set the weights so that V(x) = e'Pe + eps_s |e|^2 exactly (P symmetric positive definite), plus
        `noise` x a fresh lecun-normal draw on every entry.  With P from the CARE this is the value
        function the training converges to near xf: a synthetic "trained" network for benchmarks and
        tests.  Construction: q = L'e with P = LL'; ReLU pairs carry (q, -q) through both hidden layers."""
        n = self.dynamics.state_dim
        h1, h2, h3 = self.features
        assert 2 * n <= min(h1, h2) and n <= h3
        assert float(np.abs(self._np["mean"]).max()) == 0.0 and float(np.abs(self._np["std"] - 1).max()) == 0.0
        L = torch.as_tensor(np.linalg.cholesky(np.asarray(P, np.float64)), dtype=self.weights[0].dtype, device=self.weights[0].device)
        eye = torch.eye(n, dtype=L.dtype, device=L.device)
        W1, W2, W3 = (torch.zeros_like(w) for w in self.weights)
        W1[:, :n], W1[:, n:2 * n] = L, -L
        W2[:n, :n], W2[n:2 * n, :n], W2[:n, n:2 * n], W2[n:2 * n, n:2 * n] = eye, -eye, -eye, eye
        W3[:n, :n], W3[n:2 * n, :n] = eye, -eye
        for w, t in zip(self.weights, (W1, W2, W3)):
            if noise:
                t += noise * lecun_normal_(torch.empty_like(t), generator)
            w.copy_(t)

    def descriptor(self) -> _abi.HjbxMlp:
        d = _abi.HjbxMlp()
        W1, W2, W3 = self.weights
        d.W1, d.W2, d.W3 = W1.data_ptr(), W2.data_ptr(), W3.data_ptr()
        d.h1, d.h2, d.h3 = self.features
        d.activation = {"relu": _abi.ACT_RELU, "tanh": _abi.ACT_TANH, "sin": _abi.ACT_SIN}[self.activation]
        _abi._fill(d.mean, self._np["mean"])
        _abi._fill(d.std, self._np["std"])
        _abi._fill(d.xf, self._np["xf"])
        d.eps_scalar = self.epsilon_scalar
        return d

    @torch.no_grad()
    def fused_value_grad(self, x: torch.Tensor, want_v=True, want_grad=True):
        """Inference-only V and dV/dx from the fused MFMA kernel (float32)."""
        if self.activation not in self.FUSED_ACTIVATIONS:
            raise NotImplementedError(f"no fused value-gradient kernel for the {self.activation} activation")
        for w in self.weights:
            assert w.is_contiguous() and w.dtype == torch.float32
        return _ops.value_grad(self.dynamics.system, self.descriptor(), x, want_v, want_grad)


# ------------------------------------------------------------------------------------------------
# autograd bridges to the residual kernels (forward value + analytic first derivative)
# ------------------------------------------------------------------------------------------------
class _HJBResidualSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grad_v, x, done, sys, task, mode):
        _, dg, sums = _ops.hjb_residual(sys, task, x, grad_v.detach().contiguous(), done, mode, want_loss=False)
        ctx.save_for_backward(dg)
        ctx.mark_non_differentiable(sums)
        return sums[0].clone(), sums

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out, _grad_sums):
        (dg,) = ctx.saved_tensors
        return grad_out * dg, None, None, None, None, None


class _TerminationResidualSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, V, cost, done, eps):
        _, dv, sums = _ops.termination_residual(eps, V.detach().contiguous(), cost, done, want_loss=False)
        ctx.save_for_backward(dv)
        ctx.mark_non_differentiable(sums)
        return sums[0].clone(), sums

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out, _grad_sums):
        (dv,) = ctx.saved_tensors
        return grad_out * dv, None, None, None


def sgdr_schedule(step: int, init_value: float, peak_value: float, end_value: float, warmup_steps: int, decay_steps: int,
                  num_cycles: int) -> float:
    """optax.sgdr_schedule of `num_cycles` identical warm-up + cosine-decay cycles (vhjb.py:122-126;
    semantics assumed in SURVEY A.4): linear init->peak over `warmup_steps`, cosine peak->end over
    the remaining `decay_steps - warmup_steps`; after the last cycle it holds `end_value`."""
    c = min(int(step) // decay_steps, num_cycles - 1)
    local = int(step) - c * decay_steps
    if local < warmup_steps:
        return init_value + (peak_value - init_value) * local / warmup_steps
    span = decay_steps - warmup_steps
    k = min(local - warmup_steps, span)
    cos = 0.5 * (1.0 + math.cos(math.pi * k / span))
    alpha = end_value / peak_value if peak_value != 0 else 0.0
    return peak_value * ((1 - alpha) * cos + alpha)


class ReplayBuffer:
    """Device-resident FIFO of (x, cost, done) records (reference StatesDataset: deque(maxlen),
    vhjb.py:62-73) with shuffle-without-replacement, drop-last minibatches (the DataLoader of :154)."""

    def __init__(self, n: int, capacity: int, dtype, device):
        self.capacity = int(capacity)
        self.x = torch.empty((self.capacity, n), dtype=dtype, device=device)
        self.cost = torch.empty((self.capacity,), dtype=dtype, device=device)
        self.done = torch.empty((self.capacity,), dtype=dtype, device=device)
        self.size = 0
        self.head = 0  # next write slot

    def __len__(self):
        return self.size

    def extend(self, x: torch.Tensor, cost: torch.Tensor, done: torch.Tensor):
        k = x.shape[0]
        if k == 0:
            return
        if k >= self.capacity:
            x, cost, done = x[-self.capacity:], cost[-self.capacity:], done[-self.capacity:]
            k = self.capacity
        idx = (self.head + torch.arange(k, device=self.x.device)) % self.capacity
        self.x[idx] = x
        self.cost[idx] = cost
        self.done[idx] = done
        self.head = (self.head + k) % self.capacity
        self.size = min(self.capacity, self.size + k)

    def num_batches(self, batch_size: int) -> int:
        return self.size // batch_size

    def batches(self, batch_size: int, generator=None, limit: Optional[int] = None, out=None):
        """Minibatches (xs, costs, dones) of one pass over a fresh permutation, drop-last.  `out` = (xs, costs, dones) buffers to gather
        into (the static inputs of a captured update graph: saves three device copies per step); the same buffers are yielded each time."""
        nb = self.num_batches(batch_size)
        if limit is not None:
            nb = min(nb, limit)
        if nb == 0:
            return
        perm = torch.randperm(self.size, device=self.x.device, generator=generator)
        for b in range(nb):
            idx = perm[b * batch_size:(b + 1) * batch_size]
            if out is not None:
                torch.index_select(self.x, 0, idx, out=out[0])
                torch.index_select(self.cost, 0, idx, out=out[1])
                torch.index_select(self.done, 0, idx, out=out[2])
                yield out
            else:
                yield self.x[idx], self.cost[idx], self.done[idx]


class VHJBController(Controller):

    def __init__(self, dynamics: Dynamics, config, device=None, dtype=torch.float32, process_group=None,
                 residual_mode=_abi.RESIDUAL_NORMALISED, fused_value_grad: Optional[bool] = None,
                 graph_updates: Optional[bool] = None, activation: str = "relu", fused_param_grad: Optional[bool] = None) -> None:
        super().__init__()
        self.device = torch.device(device) if device is not None else _ops.require_device()
        self.dtype = dtype
        self.process_group = process_group
        self.world_size = torch.distributed.get_world_size(process_group) if self._distributed() else 1
        self.rank = torch.distributed.get_rank(process_group) if self._distributed() else 0

        # seeds (vhjb.py:80-83); ranks > 0 offset theirs so the shards roll out different environments
        torch.manual_seed(config.seed + self.rank)
        np.random.seed(config.seed + self.rank)
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(config.seed + self.rank)
        self._init_gen = torch.Generator(device=self.device)
        self._init_gen.manual_seed(config.seed)  # identical initial weights on every rank

        self.epsilon = float(config.epsilon)
        self.dynamics = dynamics
        self.state_dim, self.control_dim = dynamics.get_dimension()
        self.umin, self.umax = dynamics.get_control_limit()
        assert self.umin.shape[0] == self.control_dim
        assert self.umax.shape[0] == self.control_dim
        self.Q, self.R = config.Q, config.R
        self.R_inv = np.linalg.inv(np.asarray(self.R, np.float64))
        self.xf, self.uf = config.xf, config.uf
        self.obs_min, self.obs_max = config.obs_min, config.obs_max
        self.residual_mode = residual_mode

        self.system_additional_init()
        self._task = _abi.make_task(self.state_dim, self.control_dim, self.Q, self.R, self.P, self.xf, self.uf, self.obs_min,
                                    self.obs_max, self.epsilon, Rinv=self.R_inv)

        self.value_function_approximator = ValueFunctionApproximator(
            dynamics, config.features, config.normalization_mean, config.normalization_std, self.xf, config.epsilon_scalar,
            config.using_batch_norm, dtype=dtype, device=self.device, generator=self._init_gen, activation=activation)
        # activation: "relu" = controller/vhjb.py; "tanh" / "sin" = the notebooks' networks
        # the matrix-core kernels carry the five built-in systems; a user-defined system (Dynamics.device_source) runs the value network
        # through PyTorch and its own run-time compiled step / residual kernels
        builtin = dynamics.system.kind != _abi.SYS_USER
        fusable = activation in ValueFunctionApproximator.FUSED_ACTIVATIONS and builtin
        self.fused_value_grad = (dtype == torch.float32 and fusable) if fused_value_grad is None else bool(fused_value_grad)
        if self.fused_value_grad and not fusable:
            raise NotImplementedError(f"no fused value-gradient kernel for the {activation} activation")
        # the parameter gradient of the optimiser step: hand-written MFMA kernels (hjbx_value_loss_grad_f32: forward, input gradient,
        # residuals and the second-order reverse sweep in closed form) for the float32 ReLU network of controller/vhjb.py, the tanh network
        # of examples/cartpole_balancing.ipynb and (state dimension <= 4) the sin network of examples/double_integrator_optimal_time.ipynb;
        # anything else (float64, HJBX_FUSED_PARAM_GRAD=0) goes through PyTorch autograd
        can_fuse_pg = (dtype == torch.float32 and (activation in ("relu", "tanh") or (activation == "sin" and self.state_dim <= 4))
                       and tuple(config.features) == (128, 128, 64) and self.device.type == "cuda" and not config.using_batch_norm and builtin)
        if fused_param_grad is None:
            fused_param_grad = can_fuse_pg and os.environ.get("HJBX_FUSED_PARAM_GRAD", "1") != "0"
        if fused_param_grad and not can_fuse_pg:
            raise NotImplementedError("the fused parameter-gradient kernels exist for float32 ReLU / tanh / sin (n <= 4) networks with features "
                                      "[128, 128, 64] on the built-in systems only")
        self.fused_param_grad = bool(fused_param_grad)
        # fused rollouts of big batches re-pack live environments every `compaction_interval` steps (0 = never)
        self.compaction_interval, self.compaction_min_batch = 16, 8192
        self.train_mode = False
        # the optimiser step of `train` is replayed from a hipGraph (about 100 launch-bound kernels at batch 256); the
        # data-parallel path keeps eager launches around its all-reduce
        if graph_updates is None:   # HJBX_GRAPH_UPDATES=0 forces eager launches (debugging aid)
            graph_updates = self.device.type == "cuda" and not self._distributed() and os.environ.get("HJBX_GRAPH_UPDATES", "1") != "0"
        self.graph_updates = bool(graph_updates)
        self._graphed_update = None
        self._fit_graph = None
        self._reg_buf = None
        # optax.adam(lr) (vhjb.py:120): b1 0.9, b2 0.999, eps 1e-8.  On the device the fused implementation (one kernel for the three
        # weight matrices instead of ~10 foreach launches); HJBX_FUSED_ADAM=0 or a PyTorch without it falls back to the default
        adam_kw = dict(lr=config.lr, betas=(0.9, 0.999), eps=1e-8, capturable=self.graph_updates and self.device.type == "cuda")
        self.optimizer = None
        if self.device.type == "cuda" and os.environ.get("HJBX_FUSED_ADAM", "1") != "0":
            try:
                self.optimizer = torch.optim.Adam(self.value_function_approximator.parameters(), fused=True, **adam_kw)
            except (RuntimeError, TypeError, ValueError):
                self.optimizer = None
        fused_adam = self.optimizer is not None
        if self.optimizer is None:
            self.optimizer = torch.optim.Adam(self.value_function_approximator.parameters(), **adam_kw)
        # on the fused parameter-gradient path the Adam step rides in the library's mix kernel (hjbx_mix_adam_f32), working on this
        # optimiser's state tensors in place: the other paths (autograd, HJBX_FUSED_ADAM=0) and state_dict() see one and the same state
        self._native_adam = self.fused_param_grad and fused_adam
        self._adam_ticket = torch.zeros((1,), dtype=torch.int32, device=self.device) if self._native_adam else None
        self._sched = dict(init_value=config.regularization_init_value, peak_value=config.regularization_peak_value,
                           end_value=config.regularization_end_value, warmup_steps=config.regularization_warmup_steps_per_cycle,
                           decay_steps=config.regularization_total_steps_per_cycle, num_cycles=config.regularization_num_of_cycles)
        self.update_counter = 0
        self.regularization = self.regularization_scheduler(self.update_counter)
        self.epochs = config.epochs
        self.batch_size = config.batch_size

        self.maximum_timestep = config.maximum_step
        self.num_of_trajectories_per_epoch = config.num_of_trajectories_per_epoch

        # seed data set: interior points (cost 0, done 0) and boundary points (cost=min(e'Pe, clip), done 1), vhjb.py:136-150
        n = self.state_dim

        def draw(count, mean, std):
            pts = np.stack([np.random.uniform(low=-1, high=1, size=n) * std + mean for _ in range(count)]) if count else np.zeros((0, n))
            return self._dev(pts)

        interior = draw(config.num_of_interior_data, config.interior_states_mean, config.interior_states_std)
        boundary = draw(config.num_of_boundary_data, config.boundary_states_mean, config.boundary_states_std)
        self.replay_buffer = ReplayBuffer(n, config.maximum_buffer_size, dtype, self.device)
        if interior.shape[0]:
            interior = _ops.wrap(dynamics.system, interior)
            z = torch.zeros(interior.shape[0], dtype=dtype, device=self.device)
            self.replay_buffer.extend(interior, z, z)
        if boundary.shape[0]:
            boundary = _ops.wrap(dynamics.system, boundary)
            bc = torch.clamp(_ops.termination_cost(dynamics.system, self._task, boundary), max=float(config.boundary_cost_clip))
            self.replay_buffer.extend(boundary, bc, torch.ones_like(bc))

    # ---------------------------------------------------------------------------------------------
    def _distributed(self) -> bool:
        return torch.distributed.is_available() and torch.distributed.is_initialized() and \
            (self.process_group is not None or torch.distributed.get_world_size() > 1)

    def _dev(self, a) -> torch.Tensor:
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=self.dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a, np.float64), dtype=self.dtype, device=self.device).contiguous()

    def regularization_scheduler(self, step: int) -> float:
        return sgdr_schedule(step, **self._sched)

    def system_additional_init(self) -> None:
        """Linearise about (xf, uf), assume f(xf, uf) = 0, solve the CARE for the terminal cost P (vhjb.py:156-160)."""
        Alin, Blin = linearize(self.dynamics, self.xf, self.uf)
        self.Alin, self.Blin = Alin, Blin
        self.P = solve_continuous_are(Alin, Blin, self.Q, self.R)

    # -- costs (vhjb.py:162-169), batched ---------------------------------------------------------------
    def running_cost(self, x, u):
        t, one, kind = _to_device(x)
        ut = self.dynamics._control_like(u, t)
        return _from_device(_ops.running_cost(self.dynamics.system, self._task, t, ut), one, kind)

    def termination_cost(self, x):
        t, one, kind = _to_device(x)
        return _from_device(_ops.termination_cost(self.dynamics.system, self._task, t), one, kind)

    # -- control law (vhjb.py:201-225) -----------------------------------------------------------------
    @torch.no_grad()
    def get_v_gradient(self, x: torch.Tensor) -> torch.Tensor:
        """dV/dx for a (B, n) device batch (inference)."""
        if self.fused_value_grad and x.dtype == torch.float32:
            return self.value_function_approximator.fused_value_grad(x, want_v=False)[1]
        return self.value_function_approximator.value_and_grad(x)[1]

    @torch.no_grad()
    def get_control_efforts_with_additional_term(self, x) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (u, v_gradient) on the device for x (B, n) / (n,)"""
        t, one, kind = _to_device(x, like_dtype=self.dtype)
        g = self.get_v_gradient(t)
        u = _ops.control_from_grad(self.dynamics.system, self._task, t, g)
        return _from_device(u, one, kind), _from_device(g, one, kind)

    def get_control_efforts(self, x):
        return self.get_control_efforts_with_additional_term(x)[0]

    # -- rollouts --------------------------------------------------------------------------------------
    @torch.no_grad()
    def rollout_batch(self, x0: torch.Tensor, max_steps: Optional[int] = None, log_u: bool = False, log_residual: bool = False):
        """B closed loops in lock-step (the batch twin of rollout_trajectory, vhjb.py:171-193).

        x0 (B, n) on the device.  Returns time-major device tensors: traj (T+1, B, n), cost (T+1, B),
        done (T+1, B), done_step (B,) int32 [index of each env's terminal tuple], u (T, B, m) | None,
        residual (T+1, B) | None [signed normalised HJB residual along the trajectories, a by-product of the step kernel].
        Tuple t of env b is valid iff t <= done_step[b]."""
        T = self.maximum_timestep if max_steps is None else int(max_steps)
        x0 = x0.to(dtype=self.dtype, device=self.device).contiguous()
        B, n = x0.shape
        sysh, task, integ = self.dynamics.system, self._task, self.dynamics.integrator
        if self.fused_value_grad and self.dtype == torch.float32:
            done_step = torch.full((B,), -1, dtype=torch.int32, device=self.device)
            desc = self.value_function_approximator.descriptor()
            chunk = self.compaction_interval if (self.compaction_interval and B >= self.compaction_min_batch) else 0
            if not chunk or chunk >= T + 1:
                # the whole loop (T live steps + the forced terminal iteration) in one persistent kernel launch
                out = _ops.vhjb_rollout(sysh, task, desc, x0, T + 1, T, done_step, integrator=integ,
                                        log_traj=True, log_u=log_u, log_residual=log_residual)
                return dict(traj=out["traj"][:T + 1], cost=out["cost"], done=out["done"], done_step=done_step,
                            u=None if out["u"] is None else out["u"][:T], residual=out["residual"])
            # large batches: launches of `chunk` steps; between them the environments are re-packed into the kernel's tiles
            # with the live ones first (device-side argsort, no host sync), so tiles of finished environments skip the network
            S = T + 1
            traj = torch.empty((S + 1, B, n), dtype=self.dtype, device=self.device)
            cost = torch.empty((S, B), dtype=self.dtype, device=self.device)
            done = torch.empty_like(cost)
            ulog = torch.empty((S, B, self.control_dim), dtype=self.dtype, device=self.device) if log_u else None
            resid = torch.empty_like(cost) if log_residual else None
            t0, order = 0, None
            x_cur = x0
            while t0 < S:
                k = min(chunk, S - t0)
                slabs = dict(traj=traj[t0:t0 + k + 1], cost=cost[t0:t0 + k], done=done[t0:t0 + k])
                if log_u:
                    slabs["u"] = ulog[t0:t0 + k]
                if log_residual:
                    slabs["residual"] = resid[t0:t0 + k]
                _ops.vhjb_rollout(sysh, task, desc, x_cur, k, T, done_step, t_first=t0, integrator=integ, log_traj=True, log_u=log_u,
                                  log_residual=log_residual, env_order=order, out=slabs)
                t0 += k
                x_cur = traj[t0]
                if t0 < S:
                    order = torch.argsort((done_step >= 0).to(torch.int8), stable=True).to(torch.int32)
            return dict(traj=traj[:T + 1], cost=cost, done=done, done_step=done_step, u=None if ulog is None else ulog[:T], residual=resid)
        traj = torch.empty((T + 2, B, n), dtype=self.dtype, device=self.device)  # slot T+1 is scratch for the last call
        cost = torch.empty((T + 1, B), dtype=self.dtype, device=self.device)
        done = torch.empty((T + 1, B), dtype=self.dtype, device=self.device)
        ulog = torch.empty((T, B, self.control_dim), dtype=self.dtype, device=self.device) if log_u else None
        resid = torch.empty((T + 1, B), dtype=self.dtype, device=self.device) if log_residual else None
        done_step = torch.full((B,), -1, dtype=torch.int32, device=self.device)
        traj[0].copy_(x0)
        g = None
        for t in range(T + 1):
            if t < T or g is None:
                g = self.get_v_gradient(traj[t])
            _ops.vhjb_step(sysh, task, t, T, traj[t], g, traj[t + 1], cost[t], done[t], done_step,
                           u_out=(ulog[t] if (log_u and t < T) else None), integrator=integ,
                           resid_t=(resid[t] if log_residual else None))
        return dict(traj=traj[:T + 1], cost=cost, done=done, done_step=done_step, u=ulog, residual=resid)

    @torch.no_grad()
    def warm_start(self, controller, num_of_trajectories: int, max_steps: Optional[int] = None, x0=None):
        """Seed the replay buffer with closed loops of a model-based controller instead of the (untrained) learned
        policy -- BASELINE configs[2] "energy-shaping warm-start + vhjb".  New behaviour: the reference has the
        controllers (controller/acrobot_energy_shaping.py:74-121, cartpole_energy_shaping.py:65-110) and the buffer
        (vhjb.py:62-73, 308) but no code joining them.  `controller` is any DeviceFeedbackController; its rollout runs in
        the fused hjbx_rollout_feedback kernel under this controller's task with rollout_trajectory's termination
        rules (observation box, forced terminal tuple at `max_steps`), and every emitted (x, cost, done) tuple is
        appended trajectory by trajectory, exactly as `train` appends its own rollouts.
        Returns dict(records, average_trajectory_cost, average_trajectory_length, done_step)."""
        T = self.maximum_timestep if max_steps is None else int(max_steps)
        if x0 is None:
            x0 = self.dynamics.get_initial_state(batch_size=int(num_of_trajectories))
        x0 = self._dev(np.atleast_2d(x0) if isinstance(x0, np.ndarray) else x0).contiguous()
        B = x0.shape[0]
        out = _ops.rollout_feedback(self.dynamics.system, controller._descriptor(), x0, T, task=self._task,
                                    integrator=self.dynamics.integrator, terminate=True, log_traj=True, log_u=False, log_cost=True)
        ds = out["done_step"].long()
        steps = torch.arange(T + 1, device=self.device)[:, None]
        valid = steps <= ds[None, :]                                  # (T+1, B): tuples up to and including the terminal one
        done = (steps == ds[None, :]).to(self.dtype)
        vm = valid.t().reshape(-1)
        self.replay_buffer.extend(out["traj"].transpose(0, 1).reshape(-1, self.state_dim)[vm], out["cost"].t().reshape(-1)[vm],
                                  done.t().reshape(-1)[vm])
        costs = (out["cost"] * valid).sum(0)
        return dict(records=int(vm.sum().item()), average_trajectory_cost=float(costs.mean().item()),
                    average_trajectory_length=float((ds + 1).double().mean().item()), done_step=out["done_step"])

    def rollout_trajectory(self) -> List[Tuple[np.ndarray, float, float]]:
        """One trajectory as the reference returns it: a list of (x, cost, done) tuples."""
        x0 = self._dev(self.dynamics.get_initial_state(batch_size=1))
        out = self.rollout_batch(x0)
        L = int(out["done_step"][0].item()) + 1
        xs = out["traj"][:L, 0].cpu().numpy()
        cs = out["cost"][:L, 0].cpu().numpy()
        ds = out["done"][:L, 0].cpu().numpy()
        return [(xs[i], float(cs[i]), float(ds[i])) for i in range(L)]

    def get_trajectory_cost(self, trajectory):
        total_cost = 0.0
        for x, cost, done in trajectory:
            total_cost += cost
        return total_cost

    # -- losses (vhjb.py:227-253) ----------------------------------------------------------------------
    def _hjb_sums(self, xs, dones):
        _, g = self.value_function_approximator.value_and_grad(xs)
        return _HJBResidualSum.apply(g, xs, dones, self.dynamics.system, self._task, self.residual_mode)

    def _termination_sums(self, xs, dones, costs):
        V, _ = self.value_function_approximator.value_and_grad(xs, want_grad=False)
        return _TerminationResidualSum.apply(V, costs, dones, self.epsilon)

    def hjb_loss(self, xs, dones):
        xs, dones = self._dev(xs), self._dev(dones)
        s, sums = self._hjb_sums(xs, dones)
        return s / (sums[1] + self.epsilon)

    def termination_loss(self, xs, dones, costs):
        xs, dones, costs = self._dev(xs), self._dev(dones), self._dev(costs)
        s, sums = self._termination_sums(xs, dones, costs)
        return s / (sums[2] + self.epsilon)

    def params_update(self, xs, dones, costs, regularization):
        """One optimiser step (vhjb.py:255-288): grad(hjb) + regularization * grad(termination), Adam.

        Data parallel: every rank contributes the gradient of its loss SUMS and its counts through
        ONE flat all-reduce; the division by the global counts happens afterwards, so the result equals
        the single-process update on the concatenated minibatch.  Returns (total, hjb, termination) losses."""
        return self._update_core(self._dev(xs), self._dev(dones), self._dev(costs), regularization)

    def value_loss_gradient(self, xs, dones, costs) -> torch.Tensor:
        """flat = [d sum(hjb_loss)/dW1 | dW2 | dW3 | d sum(termination_loss)/dW1 | dW2 | dW3 | sum hjb, sum termination, #interior, #done] of this rank's
        samples from the fused MFMA kernels (hjbx_value_loss_grad_f32), summed over the ranks by the ONE all-reduce of the data-parallel step
        (RCCL under the nccl backend).  The division by the GLOBAL counts and the mix follow in hjbx_mix_gradients_f32 (vhjb.py:241, 253, 284)."""
        flat = _ops.value_loss_grad(self.dynamics.system, self._task, self.value_function_approximator.descriptor(), xs, costs, dones,
                                    self.residual_mode)
        if self._distributed():
            torch.distributed.all_reduce(flat, group=self.process_group)
        return flat

    def _adam_state(self, params):
        return adam_state(self.optimizer, params)

    def _one_call_update(self) -> bool:
        """params_update as ONE C-ABI call (hjbx_value_loss_adam_f32): the library's Adam, one process, value_loss_gradient not overridden"""
        return (self.fused_param_grad and self._native_adam and not self._distributed()
                and type(self).value_loss_gradient is VHJBController.value_loss_gradient and "value_loss_gradient" not in self.__dict__)

    def _update_core(self, xs, dones, costs, regularization, loss_accum=None, step_counter=None, next_mb=None):
        """`regularization` is a float, or a 0-dim device tensor when the step is being captured into a graph.  loss_accum / step_counter
        (fused path only): device-side `total_losses += ...` and `update_counter += 1` of train (vhjb.py:320-323)."""
        model_params = list(self.value_function_approximator.parameters())
        if self.fused_param_grad:
            # one C-ABI call: [d sum(hjb)/dW | d sum(termination)/dW | sum hjb, sum termination, #interior, #done] -- exactly the buffer
            # the data-parallel step all-reduces once; then the division by the (global) counts and the mix (vhjb.py:241, 253, 284)
            if self._one_call_update():
                # one process: gradient, counts, mix, the three losses and optax.adam's step (vhjb.py:120, 262-263) in ONE C-ABI call of two
                # launches -- the flat buffer is not even materialised (hjbx_value_loss_adam_f32), on the optimiser's own state tensors; in the fit
                # graph its epilogue also gathers the next update's minibatch (next_mb)
                m, v, step = self._adam_state(model_params)
                g = self.optimizer.param_groups[0]
                losses = _ops.value_loss_adam(self.dynamics.system, self._task, self.value_function_approximator.descriptor(), xs, costs, dones,
                                              self.residual_mode, regularization, self.epsilon, [p.data for p in model_params], m, v, step,
                                              self._adam_ticket, g["lr"], g["betas"][0], g["betas"][1], g["eps"], loss_accum, step_counter, next_mb)
                return losses[0], losses[1], losses[2]
            flat = self.value_loss_gradient(xs, dones, costs)
            if self._native_adam:
                # counts, mix, the three losses AND optax.adam's step (vhjb.py:120, 262-263) in one launch, on the optimiser's own state tensors
                m, v, step = self._adam_state(model_params)
                g = self.optimizer.param_groups[0]
                losses = _ops.mix_adam(flat, regularization, self.epsilon, [p.data for p in model_params], m, v, step, self._adam_ticket, g["lr"],
                                       g["betas"][0], g["betas"][1], g["eps"], loss_accum, step_counter)
                return losses[0], losses[1], losses[2]
            P = sum(p.numel() for p in model_params)
            mixed, losses = _ops.mix_gradients(flat, P, regularization, self.epsilon, loss_accum, step_counter)   # counts, mix, losses: one launch
            grads, off = [], 0
            for p in model_params:
                grads.append(mixed[off:off + p.numel()].view_as(p))
                off += p.numel()
            for p, gr in zip(model_params, grads):
                p.grad = gr
            self.optimizer.step()
            return losses[0], losses[1], losses[2]
        else:
            # differentiate w.r.t. fresh leaves that alias the parameters: their grad accumulators are created on the stream this
            # step runs on, so a hipGraph capture cannot be joined to the stream of an older, still-alive autograd graph of the
            # same parameters (that unjoined cross-stream wait crashes hipStreamEndCapture)
            params = [p.detach().requires_grad_(True) for p in model_params]
            V, g = self.value_function_approximator.value_and_grad(xs, weights=params)
            h_sum, h_sums = _HJBResidualSum.apply(g, xs, dones, self.dynamics.system, self._task, self.residual_mode)
            t_sum, _ = _TerminationResidualSum.apply(V, costs, dones, self.epsilon)
            if self._distributed():
                g_h = torch.autograd.grad(h_sum, params, retain_graph=True, allow_unused=True)
                g_t = torch.autograd.grad(t_sum, params, allow_unused=True)
                grads, hjb_loss, termination_loss = allreduce_and_mix(g_h, g_t, (h_sum.detach(), t_sum.detach(), h_sums[1], h_sums[2]), params,
                                                                      regularization, self.epsilon, self.process_group)
            else:
                # one process: the normalisers are known before the backward pass, so ONE reverse sweep of the mixed loss gives
                # grad(hjb) + regularization * grad(termination) (vhjb.py:282-284) in half the kernels
                hjb_t = h_sum / (h_sums[1] + self.epsilon)
                term_t = t_sum / (h_sums[2] + self.epsilon)
                grads = torch.autograd.grad(hjb_t + regularization * term_t, params, allow_unused=True)
                grads = [torch.zeros_like(p) if gr is None else gr for p, gr in zip(params, grads)]
                hjb_loss, termination_loss = hjb_t.detach(), term_t.detach()
        for p, gr in zip(model_params, grads):
            p.grad = gr
        self.optimizer.step()
        return hjb_loss + regularization * termination_loss, hjb_loss, termination_loss

    def params_update_graphed(self, xs, dones, costs, regularization):
        """`params_update` replayed from a hipGraph captured on first use for this minibatch shape: the same kernels in the
        same order, without ~100 host-side launches per step.  The returned losses are views of the graph's static outputs:
        consume them before the next call."""
        if not (self._graphed_update is not None and xs is self._graphed_update.inputs[0]):
            xs, dones, costs = self._dev(xs), self._dev(dones), self._dev(costs)
        if self._reg_buf is None:
            self._reg_buf, self._reg_val = torch.zeros((), dtype=self.dtype, device=self.device), 0.0
        if float(regularization) != self._reg_val:          # (one launch saved whenever the weight did not move)
            self._reg_buf.fill_(float(regularization))
            self._reg_val = float(regularization)
        if self._graphed_update is None or not self._graphed_update.matches(xs, dones, costs, self._reg_buf):
            self._graphed_update = GraphedStep(self.optimizer, list(self.value_function_approximator.parameters()), self._update_core,
                                               (xs, dones, costs, self._reg_buf))
        return self._graphed_update(xs, dones, costs, self._reg_buf)

    # -- the fit phase of one epoch, driven from the device ------------------------------------------------
    def _fit_graph_usable(self) -> bool:
        """One process, the fused float32 parameter gradient, graphs enabled, and params_update not replaced by a subclass or a test double
        (those keep the per-minibatch loop, which calls params_update like the reference does, vhjb.py:314-319)."""
        return (self.graph_updates and self.fused_param_grad and not self._distributed() and self.device.type == "cuda"
                and "params_update" not in self.__dict__ and type(self).params_update is VHJBController.params_update
                and type(self)._update_core is VHJBController._update_core)

    def _fit_epoch_graphed(self, batch: int, nb: int):
        """`for xs, costs, dones in dataloader: params_update(...)` of train (vhjb.py:314-324) with every per-update decision taken on the device:
        minibatch k = rows perm[k batch .. (k + 1) batch) of the replay buffer, gathered by hjbx_replay_gather_f32 with k read from a device
        counter; the regularisation weight of update k read from a table of the schedule uploaded once per epoch; the three running loss sums
        and the counter advanced by hjbx_mix_gradients_f32.  The host replays ONE hipGraph (gather -> parameter gradient -> mix -> Adam) nb
        times and reads the loss sums back once.  -> (sum total, sum hjb, sum termination) as floats."""
        fg = self._fit_graph
        if fg is None or fg.batch != batch:
            fg = self._fit_graph = FitGraph(self, batch)
        rb = self.replay_buffer
        perm = torch.randperm(rb.size, device=self.device, generator=self._gen)       # the same draw as ReplayBuffer.batches
        fg.perm[:rb.size].copy_(perm)
        table = torch.tensor([self.regularization_scheduler(self.update_counter + k) for k in range(nb)], dtype=torch.float32)
        fg.reg_table[:nb].copy_(table, non_blocking=False)
        fg.step.zero_()
        fg.loss_accum.zero_()
        fg.begin_epoch()
        fg.replay(nb)
        sums = fg.loss_accum.cpu().tolist()
        self.update_counter += nb
        self.regularization = self.regularization_scheduler(self.update_counter)
        return sums[0], sums[1], sums[2]

    # -- training loop (vhjb.py:290-343) ---------------------------------------------------------------
    def train(self):
        average_trajectory_cost_list = []
        std_trajectory_cost_list = []
        average_trajectory_length_list = []
        average_total_loss_list = []
        average_hjb_loss_list = []
        average_termination_loss_list = []

        per_rank_batch = max(1, self.batch_size // self.world_size)
        for epoch in range(self.epochs):
            self.train_mode = False
            ntraj = self.num_of_trajectories_per_epoch
            if ntraj > 0:
                x0 = self._dev(self.dynamics.get_initial_state(batch_size=ntraj))
                out = self.rollout_batch(x0)
                ds = out["done_step"].long()
                valid = (torch.arange(self.maximum_timestep + 1, device=self.device)[:, None] <= ds[None, :])  # (T+1, B)
                traj_costs = (out["cost"] * valid).sum(0).double().cpu().numpy()
                trajectory_lengths = int((ds + 1).sum().item())
                vm = valid.t().reshape(-1)  # trajectory-major, like extending the deque trajectory by trajectory
                self.replay_buffer.extend(out["traj"].transpose(0, 1).reshape(-1, self.state_dim)[vm],
                                          out["cost"].t().reshape(-1)[vm], out["done"].t().reshape(-1)[vm])
            # fit the value function
            total_losses = hjb_losses = termination_losses = 0.0
            self.train_mode = True
            nb = self.replay_buffer.num_batches(per_rank_batch)
            if self._distributed():
                nbt = torch.tensor([nb], device=self.device)
                torch.distributed.all_reduce(nbt, op=torch.distributed.ReduceOp.MIN, group=self.process_group)
                nb = int(nbt.item())
            if nb and self._fit_graph_usable():
                # the whole fit phase driven from the device: ONE graph replay per update and nothing else on the host
                total_losses, hjb_losses, termination_losses = self._fit_epoch_graphed(per_rank_batch, nb)
            else:
                # with a captured update graph the minibatch is gathered straight into the graph's static input buffers
                gu = self._graphed_update if self.graph_updates else None
                static = (gu.inputs[0], gu.inputs[2], gu.inputs[1]) if gu is not None and gu.inputs[0].shape[0] == per_rank_batch else None
                for xs, costs, dones in self.replay_buffer.batches(per_rank_batch, generator=self._gen, limit=nb, out=static):
                    update = self.params_update_graphed if self.graph_updates else self.params_update
                    total_loss, hjb_loss, termination_loss = update(xs, dones, costs, self.regularization)
                    total_losses = total_losses + total_loss
                    hjb_losses = hjb_losses + hjb_loss
                    termination_losses = termination_losses + termination_loss
                    self.update_counter += 1
                    self.regularization = self.regularization_scheduler(self.update_counter)

            if ntraj > 0:
                average_trajectory_cost_list.append(float(traj_costs.sum() / ntraj))
                std_trajectory_cost_list.append(float(np.var(traj_costs) ** 0.5))
                average_trajectory_length_list.append(trajectory_lengths / ntraj)
            if nb != 0:
                average_total_loss_list.append(float(total_losses / nb))
                average_hjb_loss_list.append(float(hjb_losses / nb))
                average_termination_loss_list.append(float(termination_losses / nb))
            if (epoch + 1) % 10 == 0 and self.rank == 0:
                if ntraj > 0:
                    print(f"epoch:{epoch+1}, average trajectory cost:{average_trajectory_cost_list[-1]:.2f}, "
                          f"average trajectory length:{average_trajectory_length_list[-1]:.2f}")
                if nb != 0:
                    print(f"epoch:{epoch+1}, total loss:{average_total_loss_list[-1]:.5f}, regulation: {self.regularization:.1e},"
                          f"hjb loss:{average_hjb_loss_list[-1]:.5f}, termination loss:{average_termination_loss_list[-1]:.5f}")

        return (average_trajectory_cost_list, std_trajectory_cost_list, average_trajectory_length_list,
                average_total_loss_list, average_hjb_loss_list, average_termination_loss_list)


class GraphedStep:
    """One optimiser step captured in a hipGraph (torch.cuda.CUDAGraph on ROCm).

    `core(*inputs)` must run forward, backward and `optimizer.step()` with every tensor it reads being one of `inputs`, a
    parameter or optimiser state.  The inputs become static buffers that are filled before each replay; parameters, Adam
    moments and the step counter live at fixed addresses and are updated in place by the replayed kernels (Adam must be
    built `capturable`).  The warm-up steps PyTorch requires before a capture run on a side stream and are undone afterwards
    (parameters and optimiser state restored in place), so building the graph does not move the weights."""

    def __init__(self, optimizer, params, core, example_inputs):
        self.inputs = [t.detach().clone() for t in example_inputs]
        self.shapes = [tuple(t.shape) for t in self.inputs]
        self._stream, self.graph, self.out = capture_step(optimizer, params, lambda: core(*self.inputs), self.inputs[0].device)

    def __del__(self):
        # the workspaces cached for this step's private stream die with it (5 KB of scratch per sample for the parameter gradient)
        st = getattr(self, "_stream", None)
        if st is not None:
            try:
                _ops.release_workspaces(st.cuda_stream)
            except Exception:
                pass

    def matches(self, *inputs) -> bool:
        return [tuple(t.shape) for t in inputs] == self.shapes

    def __call__(self, *inputs):
        for buf, t in zip(self.inputs, inputs):
            if t is not buf:                                   # (the caller may have filled the static buffer in place)
                buf.copy_(t)
        self.graph.replay()
        return self.out


def adam_state(optimizer, params):
    """(exp_avg list, exp_avg_sq list, step list) of a torch.optim.Adam for `params`, created the way its fused / capturable implementation
    creates them if no step has run yet (zeros; the step counts float32 device scalars) -- the tensors hjbx_mix_adam_f32 updates in place."""
    st = optimizer.state
    for p in params:
        if len(st[p]) == 0:
            st[p]["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st[p]["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st[p]["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
    return [st[p]["exp_avg"] for p in params], [st[p]["exp_avg_sq"] for p in params], [st[p]["step"] for p in params]


def capture_step(optimizer, params, step_fn, dev, before_each=None):
    """Warm `step_fn` up three times and capture it into a hipGraph, leaving parameters and optimiser state as they were.
    -> (the private stream the graph was captured on, the graph, step_fn's captured outputs)"""
    saved_p = [p.detach().clone() for p in params]
    saved_s = {p: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in optimizer.state[p].items()} for p in params if p in optimizer.state}
    # warm-up AND capture run on this one side stream: the library's workspaces (_ops: reduce tickets, rollout flags, the parameter-
    # gradient scratch) are cached per (device, stream), so the buffers the warm-up allocated -- and zero-filled, outside any capture --
    # are exactly the ones the captured launches use (capturing on torch's default capture stream allocated every workspace a second
    # time, in the graph's pool, and recorded their zero-fill as memset nodes)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(3):                      # allocates Adam's state, rocBLAS workspaces and this library's workspaces
            if before_each is not None:
                before_each()
            step_fn()
        if before_each is not None:
            before_each()
    torch.cuda.current_stream(dev).wait_stream(side)
    with torch.no_grad():
        for p, q in zip(params, saved_p):
            p.copy_(q)
            for k, v in optimizer.state[p].items():
                if torch.is_tensor(v):
                    v.copy_(saved_s[p][k]) if p in saved_s else v.zero_()
            p.grad = None
    graph = torch.cuda.CUDAGraph()
    # no cyclic garbage collection while the stream is capturing: an unreachable CUDAGraph of an earlier controller collected in the middle
    # of this capture calls hipGraphDestroy, which HIP refuses during a capture ("operation not permitted when stream is capturing") -- from
    # a destructor, i.e. the process dies.  (torch.cuda.graph collects once on entry; allocations inside the capture can trigger it again.)
    import gc
    was_enabled = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        with torch.cuda.graph(graph, stream=side):
            out = step_fn()
    finally:
        if was_enabled:
            gc.enable()
    return side, graph, out


class FitGraph:
    """The optimiser step of the fit phase with its minibatch selection inside the graph (VHJBController._fit_epoch_graphed):
    hjbx_value_loss_adam_f32 (gradient kernel + one epilogue: reduce / mix / Adam / gather of the NEXT minibatch; the first minibatch of an epoch
    by hjbx_replay_gather_f32).  Static buffers: perm (replay capacity, int32), reg_table (one
    entry per update of an epoch), step (device update counter within the epoch), loss_accum (3 running loss sums).  Because nothing on the
    host changes between two updates, UNROLL consecutive updates are also captured as one graph: a launch of that graph pays the
    graph-to-graph gap (about 8 us on MI355X) once per UNROLL updates."""

    UNROLL = int(os.environ.get("HJBX_FIT_UNROLL", "8"))     # measured (cartpole, 256 samples): 4: 0.0444, 8: 0.0438, 16 / 32: 0.0430 ms per update

    def __init__(self, ctl, batch: int):
        rb, dev = ctl.replay_buffer, ctl.device
        assert rb.x.dtype == torch.float32
        self.batch = int(batch)
        n = rb.x.shape[1]
        # zeros: the warm-up launches gather row 0; sized so that UNROLL warm-up updates stay inside
        self.perm = torch.zeros((max(rb.capacity, self.UNROLL * batch),), dtype=torch.int32, device=dev)
        self.reg_table = torch.zeros((rb.capacity // batch + self.UNROLL + 1,), dtype=torch.float32, device=dev)
        self.step = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.loss_accum = torch.zeros((3,), dtype=torch.float32, device=dev)
        self.xs = torch.empty((batch, n), dtype=torch.float32, device=dev)
        self.costs = torch.empty((batch,), dtype=torch.float32, device=dev)
        self.dones = torch.empty((batch,), dtype=torch.float32, device=dev)
        self.reg = torch.zeros((), dtype=torch.float32, device=dev)
        self._ctl_ref = weakref.ref(ctl)      # (no reference cycle: the graphs die with the controller, not at some later garbage collection)
        self._streams, self._graphs = [], {}
        # two kernels per update when the one-call update is in use: its epilogue gathers the next minibatch
        self.fused_next = ctl._one_call_update()
        self._next_mb = _ops.next_minibatch(rb.x, rb.cost, rb.done, self.perm, self.reg_table, self.xs, self.costs, self.dones, self.reg) if self.fused_next else None
        self._capture(1)

    def _gather(self):
        rb = self._ctl_ref().replay_buffer
        _ops.replay_gather(rb.x, rb.cost, rb.done, self.perm, self.step, self.reg_table, self.xs, self.costs, self.dones, self.reg)

    def _one_update(self):
        ctl = self._ctl_ref()
        if self.fused_next:
            # the previous update's epilogue (or _gather at the start of the epoch) has assembled this minibatch; this update's assembles the next
            return ctl._update_core(self.xs, self.dones, self.costs, self.reg, loss_accum=self.loss_accum, step_counter=self.step, next_mb=self._next_mb)
        self._gather()
        return ctl._update_core(self.xs, self.dones, self.costs, self.reg, loss_accum=self.loss_accum, step_counter=self.step)

    def begin_epoch(self):
        """(after perm and reg_table are filled and step is zeroed)"""
        if self.fused_next:
            self._gather()

    def _capture(self, count: int):
        ctl = self._ctl_ref()

        def step_fn():
            for _ in range(count):
                out = self._one_update()
            return out

        saved = (self.step.clone(), self.loss_accum.clone())
        def before_each():
            self.step.zero_()
            self.begin_epoch()

        stream, graph, out = capture_step(ctl.optimizer, list(ctl.value_function_approximator.parameters()), step_fn, ctl.device, before_each=before_each)
        self.step.copy_(saved[0])
        self.loss_accum.copy_(saved[1])
        self.begin_epoch()                     # (the warm-up runs left another minibatch in the input buffers: gather the current one again)
        self._streams.append(stream)
        self._graphs[count] = (graph, out)

    def replay(self, count: int):
        if count >= self.UNROLL and self.UNROLL not in self._graphs:
            self._capture(self.UNROLL)
        many = self._graphs.get(self.UNROLL)
        while many is not None and count >= self.UNROLL:
            many[0].replay()
            count -= self.UNROLL
        for _ in range(count):
            self._graphs[1][0].replay()

    def __del__(self):
        for st in getattr(self, "_streams", ()):
            try:
                _ops.release_workspaces(st.cuda_stream)
            except Exception:
                pass



# ------------------------------------------------------------------------------------------------
# flat gradient buffer for the single all-reduce
# ------------------------------------------------------------------------------------------------
def allreduce_and_mix(g_h, g_t, scalars, params, regularization, epsilon, process_group=None):
    """The data-parallel step of params_update.  Inputs are THIS rank's gradients of the loss SUMS
    (hjb, termination) and `scalars` = (sum hjb, sum termination, #interior, #done) of its shard.
    One flat all-reduce (sum) over [g_h | g_t | scalars], then the division by the GLOBAL counts
    (vhjb.py:241, 253) -- per-rank means would be wrong when shards hold different numbers of done states.
    `process_group=False` skips the collective (single process).  -> (mixed grads, hjb_loss, termination_loss)"""
    flat = pack_flat(g_h, g_t, params, scalars)
    if process_group is not False and torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.all_reduce(flat, group=process_group)
    g_h, g_t, (hs, ts, n_int, n_done) = unpack_flat(flat, params)
    hjb_loss = hs / (n_int + epsilon)
    termination_loss = ts / (n_done + epsilon)
    grads = [a / (n_int + epsilon) + regularization * (b / (n_done + epsilon)) for a, b in zip(g_h, g_t)]
    return grads, hjb_loss, termination_loss


def mix_flat(flat: torch.Tensor, params, regularization, epsilon):
    """[g_h | g_t | sum hjb, sum termination, #interior, #done] (already summed over the ranks) -> (mixed grads as views of ONE buffer,
    hjb_loss, termination_loss): grad = g_h / (#interior + eps) + regularization * g_t / (#done + eps), vhjb.py:241, 253, 284."""
    P = sum(p.numel() for p in params)
    hs, ts, n_int, n_done = flat[2 * P], flat[2 * P + 1], flat[2 * P + 2], flat[2 * P + 3]
    ih, it = 1.0 / (n_int + epsilon), 1.0 / (n_done + epsilon)
    mixed = flat[:P] * ih + flat[P:2 * P] * (regularization * it)
    grads, off = [], 0
    for p in params:
        grads.append(mixed[off:off + p.numel()].view_as(p))
        off += p.numel()
    return grads, hs * ih, ts * it


def pack_flat(g_h, g_t, params, scalars) -> torch.Tensor:
    """[grad of sum(hjb) | grad of sum(termination) | scalars...] as one contiguous buffer."""
    parts = []
    for gs in (g_h, g_t):
        for p, g in zip(params, gs):
            parts.append((torch.zeros_like(p) if g is None else g).reshape(-1))
    parts.append(torch.stack([s.reshape(()).to(parts[0].dtype) for s in scalars]))
    return torch.cat(parts)


def unpack_flat(flat: torch.Tensor, params):
    out, off = [], 0
    for _ in range(2):
        gs = []
        for p in params:
            k = p.numel()
            gs.append(flat[off:off + k].view_as(p))
            off += k
        out.append(gs)
    return out[0], out[1], tuple(flat[off + i] for i in range(flat.numel() - off))
