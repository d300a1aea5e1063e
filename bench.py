#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched VHJB closed loop on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1]): cartpole balancing, VHJB controller with the
4->128->128->64 value network, B = 2^20 float32 environments PER GPU (weak scaling, the default) or 2^20 in
total sharded over the ranks (`--scaling strong`); environments are independent, so the rollout has no
data-path collective.  One "step" = one closed-loop environment step for the whole batch, i.e. one iteration
of rollout_trajectory's loop (reference controller/vhjb.py:175-186) for B environments: value gradient of the
current states -> HJB-optimal control -> running cost -> bounds / termination -> forward-Euler step -> log
(x, cost, done).  States stay resident in HBM.

The value network carries synthetic "trained" weights (the LQR value function embedded exactly, plus
5 % dense lecun-normal noise so no weight is zero), so environments stay inside the observation box like
the reference's trained policy (average trajectory length 200/200, examples/cartpole_balancing.ipynb
cell 10); `value` counts LIVE environment steps only.

Timing contract: W untimed warm-up steps, then a block of EXACTLY K steps bracketed by barrier +
torch.cuda.synchronize() on both sides; that block is repeated `--reps` times from the same post-warm-up state
(identical work every time), each repetition's time is the MAX over ranks, and `ms_per_step` / `value` come from
the MEDIAN repetition (min / max are printed beside it).

Launching: `python bench.py --gpus N` starts N fresh rank processes itself (one per GPU, before anything in
the parent touches the GPU); under `python -m torch.distributed.run ... bench.py --gpus N` it uses the
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* of the launcher.  Rank 0 prints ONE JSON line.

`roofline` is for the dominant hand-written kernel, timed live with HIP events on the launch stream;
`cpu_baseline` is the CPU oracle (oracle/, the port of the reference's batch-1 loop) on this box's host cores
over a bounded sample of the same workload; `secondary` carries the other BASELINE configs' closed loops, the
HBM-bound entry points (buffers rotated through a pool larger than the 256 MB Infinity Cache) and the
optimiser step, each measured in the same run.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0         # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3  # dense f32-input MFMA peak (same guide)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (same guide; AMD's 5 PF figure includes 2:1 sparsity)
ARITHMETIC = "f32"              # set from --arithmetic in main() (f32 = the library default = the reference's arithmetic)
HEADLINE_BATCH = {"cartpole": 1 << 20, "acrobot": 1 << 20, "quad2d": 1 << 18, "nearhover": 1 << 20, "linear": 1 << 20}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--reps", type=int, default=10, help="repetitions of the timed K-step block (median reported)")
    ap.add_argument("--batch", type=int, default=0, help="environments per GPU (weak) or in total (strong); 0 = the BASELINE size of --system")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch environments per GPU; strong: --batch environments in total, sharded contiguously over the ranks")
    ap.add_argument("--system", default="cartpole", choices=["cartpole", "acrobot", "quad2d", "nearhover", "linear"],
                    help="default cartpole = BASELINE configs[1]; quad2d / nearhover = the VHJB loops of configs[3] / configs[4]")
    ap.add_argument("--integrator", default="euler", choices=["euler", "rk4"], help="euler = the reference's integrator (parity mode)")
    ap.add_argument("--activation", default="relu", choices=["relu", "tanh"], help="relu = controller/vhjb.py (the BASELINE workload); tanh = the cartpole notebook's network")
    ap.add_argument("--arithmetic", default="f32", choices=["f32", "bf16x3", "f16x2"],
                    help="value-network arithmetic of the fused kernels (HJBX_OPT_MLP_ARITHMETIC): f32 (default, the reference's arithmetic) = float32 MFMA (an fmaf chain, bitwise); opt-in: bf16x3 = every "
                         "float32 operand split exactly into three bfloat16 pieces, six piece products on the bf16 matrix cores; f16x2 = operands scaled per "
                         "environment and rounded to two float16 pieces (22 bits), three piece products on the f16 matrix cores")
    ap.add_argument("--chunk", type=int, default=0, help="steps per persistent launch (0 = all K steps in one launch)")
    ap.add_argument("--prewarm", type=float, default=0.3, help="seconds of untimed clock pre-warm on scratch state before the W warm-up steps (0 for counter runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (other configs, HBM-bound entry points, optimiser step)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the multi-rank path on one GPU)")
    ap.add_argument("--torch-mlp", action="store_true", help="value gradient through PyTorch matmuls instead of the fused kernel")
    ap.add_argument("--stepwise", action="store_true", help="two launches per step (value_grad + vhjb_step) instead of the persistent rollout kernel")
    ap.add_argument("--cpu-sample-envs", type=int, default=0, help="0 = auto (about 10-20 s of CPU work)")
    ap.add_argument("--dry-run", action="store_true", help="launcher check only: start the ranks, rendezvous over gloo on the CPU, all-reduce the "
                    "rank ids and print the line's launch fields; no GPU work, no measurement (used by the CPU test of the N > 1 launch path)")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torchrun
# ----------------------------------------------------------------------------------------------------------------------
def self_launch(args) -> int:
    """Start N rank processes (one per GPU) and wait for them.  Runs in a parent that has made no HIP call: the children
    are fresh interpreters (subprocess, no fork of an initialised runtime, no exec of a process that touched the GPU)."""
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        out = None if r == 0 else subprocess.DEVNULL          # rank 0's stdout (the JSON line) is this process's stdout
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    rc = 0
    try:
        for p in procs:
            rc = rc or p.wait()
            if rc:
                break
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc


# ----------------------------------------------------------------------------------------------------------------------
# workloads
# ----------------------------------------------------------------------------------------------------------------------
def make_workload(system, integrator, activation, B, seed, torch_mlp=False):
    from q_learning_with_hjb_amd import _abi
    from q_learning_with_hjb_amd.configs import defaults as D
    from q_learning_with_hjb_amd.controller.vhjb import VHJBController
    from q_learning_with_hjb_amd.dynamics.acrobot import Acrobot
    from q_learning_with_hjb_amd.dynamics.cartpole import Cartpole
    from q_learning_with_hjb_amd.dynamics.linear import LinearDynamics
    from q_learning_with_hjb_amd.dynamics.quadrotors import NearHoverQuadcopter, Quadrotors2D
    dyn, ccfg, label = {
        "cartpole": lambda: (Cartpole(D.cartpole_dynamics_config()), D.cartpole_vhjb_config(), "cartpole balancing + vhjb controller (BASELINE configs[1])"),
        "acrobot": lambda: (Acrobot(D.acrobot_dynamics_config(x0_mean=[np.pi, 0, 0, 0], x0_std=[0.5, 0.5, 1, 1])), D.acrobot_vhjb_config(),
                            "acrobot at the upright + vhjb controller (BASELINE configs[2], closed-loop part)"),
        "quad2d": lambda: (Quadrotors2D(D.quadrotors2d_dynamics_config()), D.quadrotors2d_vhjb_config(), "Quadrotors2D hovering + vhjb controller (BASELINE configs[3])"),
        "nearhover": lambda: (NearHoverQuadcopter(D.near_hover_dynamics_config()), D.near_hover_vhjb_config(), "10-D near-hover quadcopter + vhjb controller (BASELINE configs[4])"),
        "linear": lambda: (LinearDynamics(D.linear_dynamics_config()), D.linear_vhjb_config(), "double integrator + vhjb controller"),
    }[system]()
    if integrator == "rk4":
        dyn.integrator = _abi.RK4
    ctl = VHJBController(dyn, ccfg, fused_value_grad=not torch_mlp, activation=activation)
    vf = ctl.value_function_approximator
    wgen = torch.Generator(device="cuda"); wgen.manual_seed(1234)
    vf.load_quadratic(ctl.P, noise=0.05, generator=wgen)
    gen = torch.Generator(device="cuda"); gen.manual_seed(seed)
    x0 = dyn.get_initial_state(B, generator=gen)
    if system in ("quad2d", "nearhover"):
        x0 = (x0 * 0.5).contiguous()             # start inside the observation box (the stock x0 box is as wide as it)
    n, m = dyn.get_dimension()
    return dict(system=system, dyn=dyn, ctl=ctl, vf=vf, x0=x0, n=n, m=m, B=B, label=label,
                flops_per_env=4.0 * (n * 128 + 128 * 128 + 128 * 64))   # value net fwd + input-grad MACs x 2


T_INF = 1 << 30   # no forced termination inside a timed region


class FusedLoop:
    """The persistent rollout kernel (hjbx_vhjb_rollout_f32) driven from a saved post-warm-up state."""

    def __init__(self, wl, chunk):
        from q_learning_with_hjb_amd import _ops
        self.ops, self.wl, self.chunk = _ops, wl, chunk
        self.desc = wl["vf"].descriptor()
        self.done_step = torch.full((wl["B"],), -1, dtype=torch.int32, device="cuda")
        self.x, self.t = wl["x0"], 0

    def prewarm(self, seconds=0.3):
        """~0.3 s of the same kernel on scratch state so the clocks have ramped before the W warm-up steps start (the first
        launches of a fresh process run ~15 % below the settled rate)."""
        wl = self.wl
        ds = self.done_step.clone()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            self.ops.vhjb_rollout(wl["dyn"].system, wl["ctl"]._task, self.desc, wl["x0"], 25, T_INF, ds, integrator=wl["dyn"].integrator, log_traj=False)
            torch.cuda.synchronize()

    def run(self, nsteps, events=None):
        wl = self.wl
        left = nsteps
        while left > 0:
            k = min(self.chunk or left, left)
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            out = self.ops.vhjb_rollout(wl["dyn"].system, wl["ctl"]._task, self.desc, self.x, k, T_INF, self.done_step, t_first=self.t,
                                        integrator=wl["dyn"].integrator, log_traj=True, want_x_out=True)
            if events is not None:
                e1.record()
                events.append((e0, e1, k))
            self.x = out["x_out"]
            self.t += k
            left -= k

    def save(self):
        return self.x.clone(), self.done_step.clone(), self.t

    def restore(self, st):
        self.x, self.t = st[0], st[2]
        self.done_step.copy_(st[1])


def time_fused(wl, K, W, reps, chunk, barrier, prewarm=0.3):
    """-> (per-rep wall seconds, per-launch (ms, steps) list, live env-steps of one K-step block on this rank)"""
    loop = FusedLoop(wl, chunk)
    if prewarm:
        loop.prewarm(prewarm)
    loop.run(W)
    barrier()
    state = loop.save()
    walls, launches = [], []
    for _ in range(reps):
        loop.restore(state)
        evs = []
        barrier()
        t0 = time.perf_counter()
        loop.run(K, evs)
        barrier()
        walls.append(time.perf_counter() - t0)
        launches += [(a.elapsed_time(b), k) for a, b, k in evs]
    ds = loop.done_step.long()
    live = int(torch.where(ds < 0, torch.full_like(ds, K), (ds - W).clamp(min=0, max=K)).sum().item())
    return walls, launches, live


def rollout_traffic_model(n, m, B, steps_per_launch):
    """HBM bytes of one launch of k_vhjb_rollout_mfma.  Algorithmic: per environment-step the log row (x', cost, done) = 4(n+2) bytes
    written; per launch and environment: x read (4n), done_step read + written (8), traj slab 0 + x_out written (8n).  `measured` uses
    the per-step and per-launch coefficients fitted to rocprofv3 PMC passes (FETCH_SIZE x 2 + WRITE_SIZE at two launch lengths,
    profiles/traffic.json) instead, when present."""
    alg = B * (steps_per_launch * 4.0 * (n + 2) + 4.0 * (3 * n + 2))
    measured = None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile):
        with open(tfile) as f:
            rec = json.load(f).get(f"k_vhjb_rollout_mfma/n{n}")
        if rec:
            measured = B * (steps_per_launch * rec["bytes_per_env_step"] + rec["bytes_per_env_launch"])
    return alg, measured


def mfma_roofline(wl, launches, B):
    ms = float(np.median([a / k for a, k in launches]))            # ms per step inside a launch
    steps_per_launch = float(np.mean([k for _, k in launches]))
    launch_ms = ms * steps_per_launch
    ach = wl["flops_per_env"] * B / (ms * 1e-3) / 1e12
    alg, meas = rollout_traffic_model(wl["n"], wl["m"], B, steps_per_launch)
    if ARITHMETIC != "f32":
        # executed matrix-core work: layers 2, 3 forward and backward as six bf16 (three f16) piece products each (layer 1 stays on the f32
        # MFMA and is not counted); priced against the dense bf16 / f16 peak
        pieces = 6 if ARITHMETIC == "bf16x3" else 3
        executed = pieces * 4.0 * (128 * 128 + 128 * 64)
        ach16 = executed * B / (ms * 1e-3) / 1e12
        return dict(bound="mfma", kernel=f"k_vhjb_rollout_mfma<{ARITHMETIC}> (hjbx_vhjb_rollout_f32, HJBX_OPT_MLP_ARITHMETIC={1 if pieces == 6 else 2})",
                    achieved=ach16, peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s", frac=ach16 / MFMA_BF16_PEAK_TFLOPS, traffic=meas, traffic_algorithmic=alg,
                    avg_launch_ms=launch_ms, steps_per_launch=steps_per_launch, flop_per_env_step=executed,
                    note=f"achieved = EXECUTED 16-bit MFMA flops ({pieces} piece products per float32 product) against the dense bf16 / f16 peak; "
                         "the float32-equivalent rate is f32_equivalent",
                    f32_equivalent=ach, f32_equivalent_over_f32_mfma_peak=ach / MFMA_F32_PEAK_TFLOPS)
    return dict(bound="mfma", kernel="k_vhjb_rollout_mfma (hjbx_vhjb_rollout_f32)", achieved=ach, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
                frac=ach / MFMA_F32_PEAK_TFLOPS, traffic=meas, traffic_algorithmic=alg, avg_launch_ms=launch_ms, steps_per_launch=steps_per_launch,
                flop_per_env_step=wl["flops_per_env"])


# ----------------------------------------------------------------------------------------------------------------------
# secondary measurements
# ----------------------------------------------------------------------------------------------------------------------
def timed_rotating(fn, sets, reps=60, warm=12):
    """Average duration of fn(set) over `reps` launches, cycling through `sets` so that consecutive launches touch different
    buffers (a pool > 2 x the 256 MB Infinity Cache makes every read come from HBM)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for r in range(warm):
        fn(sets[r % len(sets)])
    e0.record()
    for r in range(reps):
        fn(sets[(warm + r) % len(sets)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def hbm_entry_points(system="cartpole", B=1 << 20, pool_bytes=640 << 20):
    """simulate / vhjb_step / hjb_residual at B = 2^20: achieved GB/s of the ALGORITHMIC bytes."""
    from q_learning_with_hjb_amd import _abi, _ops
    wl = make_workload(system, "euler", "relu", B, 7)
    d, ctl, n, m = wl["dyn"], wl["ctl"], wl["n"], wl["m"]
    task = ctl._task
    gen = torch.Generator(device="cuda").manual_seed(3)
    rows = []

    def pool(per_set_bytes, make):
        k = max(2, int(np.ceil(pool_bytes / per_set_bytes)))
        return [make() for _ in range(k)]

    def row(name, secs, bytes_per_env, nsets):
        gbs = bytes_per_env * B / secs / 1e9
        rows.append(dict(name=f"{name} ({system}, B=2^{int(np.log2(B))})", us=secs * 1e6, bytes_per_env=bytes_per_env, achieved=gbs, unit="GB/s",
                         frac=gbs / HBM_PEAK_GBS, bound="hbm", rotating_sets=nsets))

    def st():
        x = (wl["x0"] + 0.01 * torch.randn((B, n), generator=gen, device="cuda")).contiguous()
        return dict(x=x, u=torch.randn((B, m), generator=gen, device="cuda"), g=torch.randn((B, n), generator=gen, device="cuda"),
                    xn=torch.empty_like(x), c=torch.empty(B, device="cuda"), dn=torch.zeros(B, device="cuda"),
                    ds=torch.full((B,), -1, dtype=torch.int32, device="cuda"))
    sets = pool(4 * B * (3 * n + m + 3), st)
    row("simulate euler", timed_rotating(lambda s: _ops.simulate(d.system, s["x"], s["u"], _abi.EULER, out=s["xn"]), sets), 4 * (2 * n + m), len(sets))
    row("simulate rk4", timed_rotating(lambda s: _ops.simulate(d.system, s["x"], s["u"], _abi.RK4, out=s["xn"]), sets), 4 * (2 * n + m), len(sets))
    row("vhjb_step euler", timed_rotating(lambda s: _ops.vhjb_step(d.system, task, 0, T_INF, s["x"], s["g"], s["xn"], s["c"], s["dn"], s["ds"]), sets),
        4 * (3 * n + 2) + 4, len(sets))
    row("hjb_residual fwd+bwd+sums", timed_rotating(lambda s: _ops.hjb_residual(d.system, task, s["x"], s["g"], s["dn"], want_loss=False), sets),
        4 * (3 * n + 1), len(sets))
    return rows


def param_gradient_kernels(system="nearhover", B=1 << 20):
    """hjbx_value_loss_grad_f32 (the fused MFMA parameter gradient of the value-learning step) on a full batch: samples/s and the
    fraction of the f32 MFMA peak, counting ALGORITHMIC flops (2 x MACs of the products in the header of hjbx_train.hip:
    6 S + 128^2 + 128 x 64 MACs per sample, S = 128 n + 128^2 + 128 x 64)."""
    from q_learning_with_hjb_amd import _ops
    wl = make_workload(system, "euler", "relu", B, 17)
    d, ctl, n = wl["dyn"], wl["ctl"], wl["n"]
    gen = torch.Generator(device="cuda").manual_seed(9)
    dones = (torch.rand(B, generator=gen, device="cuda") < 0.1).float()
    costs = torch.rand(B, generator=gen, device="cuda") * 5
    desc = wl["vf"].descriptor()
    out = None
    for _ in range(3):
        out = _ops.value_loss_grad(d.system, ctl._task, desc, wl["x0"], costs, dones, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _ops.value_loss_grad(d.system, ctl._task, desc, wl["x0"], costs, dones, out=out)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    t = float(np.median(ts))
    S = 128 * n + 128 * 128 + 128 * 64
    flops = 2.0 * (6 * S + 128 * 128 + 128 * 64)
    ach = flops * B / t / 1e12
    from q_learning_with_hjb_amd import _abi
    pair = ARITHMETIC == "f16x2" or _abi.set_option(_abi.OPT_TRAIN_KERNEL, -1) == 1      # (set_option(-1) only reads the option)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r03_train_coop_summary.json")
    if not pair and os.path.exists(tpath):
        with open(tpath) as f:
            rec = json.load(f).get("k_train_coop<0, 0, PS, hjbx::%s<float>>" % dict(cartpole="Cartpole", nearhover="NearHover").get(system, ""), {})
        traffic = rec.get("bytes_per_sample")
    return dict(name=f"value_loss_grad: parameter gradient of the learning step ({system}, B=2^{int(np.log2(B))})", ms=t * 1e3, samples_per_s=B / t,
                achieved=ach, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s", frac=ach / MFMA_F32_PEAK_TFLOPS, bound="mfma", flop_per_sample=flops,
                implementation=("k_train_chains + k_train_outer + k_train_reduce (two kernels, 5 KB of scratch per sample each way)" if pair else
                                "k_train_coop + k_train_coop_reduce (one cooperative kernel, operands exchanged and transposed in LDS, no scratch in HBM)"),
                scratch_bytes_per_sample=(2 * 4.0 * 40960 / 32 if pair else 0.0), algorithmic_input_bytes_per_sample=4.0 * (n + 2),
                hbm_bytes_per_sample_by_counters=traffic, arithmetic=ARITHMETIC,
                note="achieved = ALGORITHMIC float32 flops per second against the f32 MFMA peak; in the default arithmetic (f32) every product runs on "
                     "the f32 MFMA; with --arithmetic f16x2 the two-kernel form runs its eight wide chain products as f16x2 split-operand chains on the "
                     "f16 MFMA (3 piece products each) and frac is then a float32-equivalent rate, not the occupancy of one pipe; "
                     "hbm_bytes_per_sample_by_counters is static (profiles/r03_train_coop_summary.json: 2 x FETCH_SIZE + WRITE_SIZE at B = 2^20)")


def optimiser_step(world, dist, system="cartpole", total=256):
    """params_update (reference controller/vhjb.py:255-288) on `total` samples IN TOTAL (total / G per rank; 256 = the reference's
    minibatch; 2^20 per GPU = the sharded training step of BASELINE configs[4]): updates per second including, for G > 1, the flat
    gradient all-reduce (RCCL)."""
    wl = make_workload(system, "euler", "relu", max(4096, total // world), 11)
    ctl = wl["ctl"]
    per_rank = max(1, total // world)
    gen = torch.Generator(device="cuda").manual_seed(5)
    xs = wl["x0"][:per_rank].contiguous()
    dones = (torch.rand(per_rank, generator=gen, device="cuda") < 0.1).float()
    costs = torch.rand(per_rank, generator=gen, device="cuda") * 5
    update = ctl.params_update_graphed if ctl.graph_updates else ctl.params_update
    for _ in range(10):
        update(xs, dones, costs, 1e-5)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    R = 100 if per_rank <= 65536 else 20
    for _ in range(R):
        update(xs, dones, costs, 1e-5)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    fit = None
    if ctl._fit_graph_usable() and per_rank <= 4096:
        # the same update as train() issues it (vhjb.py:314-324): minibatch selection, regularisation schedule and loss sums on the device,
        # one graph replay per update, one read-back per epoch
        rb = ctl.replay_buffer
        nbt = min(rb.capacity // per_rank, 400)
        while rb.size < nbt * per_rank:
            k = min(wl["x0"].shape[0], nbt * per_rank - rb.size)
            rb.extend(wl["x0"][:k], torch.rand(k, generator=gen, device="cuda") * 5, (torch.rand(k, generator=gen, device="cuda") < 0.1).float())
        nbt = rb.size // per_rank
        ctl._fit_epoch_graphed(per_rank, nbt)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t1 = time.perf_counter()
            ctl._fit_epoch_graphed(per_rank, nbt)               # ends with the read-back of the loss sums: synchronous
            ts.append((time.perf_counter() - t1) / nbt)
        fit = dict(updates_per_epoch=nbt, ms_per_update=float(np.median(ts)) * 1e3, updates_per_s=1.0 / float(np.median(ts)),
                   what="VHJBController.train's fit phase: one hipGraph replay per update (gather -> parameter gradient -> reduce + mix + Adam epilogue), per-epoch "
                        "host work (permutation, schedule table, loss read-back) included")
    return dict(name=f"params_update ({system}, {total} samples in total)", ranks=world, samples_per_rank=per_rank, updates_per_s=R / dt,
                samples_per_s=R * per_rank * world / dt, ms_per_update=dt / R * 1e3, fit_phase=fit,
                gradient=("fused MFMA kernels (hjbx_value_loss_grad_f32)" if ctl.fused_param_grad else "PyTorch autograd"),
                mode=("hipGraph replay" if ctl.graph_updates else "eager launches" + (" + one flat all-reduce" if world > 1 else "")))


def strong_scaling_line(world, rank, dist, barrier, system="quad2d", total=1 << 18, K=40, W=10, reps=3):
    """BASELINE configs[3] as a STRONG-scaling measurement: `total` planar quadrotors in all, rank r rolls out the contiguous shard
    [r total / G, (r + 1) total / G) (2^18 -> 8 x 2^15 on a full node); no data-path collective; time = max over ranks per repetition."""
    lo, hi = rank * total // world, (rank + 1) * total // world
    wl = make_workload(system, "euler", "relu", hi - lo, 1000 + rank)
    walls, _, live = time_fused(wl, K, W, reps, 0, barrier, prewarm=0.0)
    wt = torch.tensor(walls, device="cuda", dtype=torch.float64)
    lt = torch.tensor([live], device="cuda", dtype=torch.int64)
    if dist is not None:
        dist.all_reduce(wt, op=dist.ReduceOp.MAX)
        dist.all_reduce(lt)
    el = float(np.median(wt.cpu().numpy()))
    del wl
    torch.cuda.empty_cache()
    return dict(name=f"strong scaling: fused vhjb rollout, {system}, 2^{int(np.log2(total))} environments IN TOTAL over {world} rank(s) (BASELINE configs[3])",
                value=int(lt.item()) / el, unit="env-steps/s", ms_per_step=el / K * 1e3, ranks=world, shard=hi - lo, scaling="strong", arithmetic=ARITHMETIC)


def parity_evidence(arith, system):
    """What the GPU parity tests measured for this arithmetic (tests/test_gpu_f32_parity.py writes the report; the committed copy is
    profiles/r03_f32_parity_report.json): per-element errors of one teacher-forced step at full batch against the f64 oracle, relative to each
    element's own term scale, for the kernel and -- the yardstick the tests assert against -- for the oracle's float build on the CPU.
    Static data, not re-measured by the bench."""
    path = os.path.join(ROOT, "profiles", "r03_f32_parity_report.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        rep = json.load(f)
    out = dict(source="profiles/r03_f32_parity_report.json (tests/test_gpu_f32_parity.py, full batch, f64 oracle)",
               bound="kernel max and p99.9 error <= 2 x the same figures of the oracle's float32 build (CPU), per quantity, errors relative to the element's term scale")
    for a in dict.fromkeys((arith, "f32")):
        rows = {"/".join(k.split("/")[3:]): v for k, v in rep.items() if k.startswith(f"teacher_forced/{a}/{system}/")}
        if rows:
            out[a] = {w: dict(kernel_max={q: r[q]["kernel"]["max"] for q in ("x_next", "u", "cost", "residual")},
                              cpu_f32_max={q: r[q]["cpu_f32"]["max"] for q in ("x_next", "u", "cost", "residual")},
                              at_relu_kink_envs=r["at_kink_envs"], at_relu_kink_matching_neither_side=r["at_kink_matching_no_side"])
                      for w, r in rows.items()}
        ds = rep.get(f"done_step/{a}/{system}/euler")
        if ds and a in out:
            out[a]["done_step_30_steps"] = {k: ds[k] for k in ("mismatches_in_safe", "cpu_f32_mismatches_in_safe", "mismatches_in_band", "cpu_f32_mismatches_in_band",
                                                               "filtered_fraction") if k in ds}
    return out


# ----------------------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))              # the parent never touches the GPU

    # stdout carries ONE JSON line: whatever the libraries print there meanwhile (gloo announces its connections on stdout) goes to stderr
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.dry_run:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        one = torch.ones(1)
        dist.all_reduce(one)                                     # the same `ranks_seen` probe the measured run does on the device
        total = args.batch or HEADLINE_BATCH[args.system]
        shard = [(r * total // world, (r + 1) * total // world) for r in range(world)] if args.scaling == "strong" else None
        if rank == 0:
            sys.stdout.flush()
            os.write(stdout_fd, (json.dumps(dict(dry_run=True, n_gpus=world, rank_sum=int(t.item()), ranks_seen=int(one.item()), scaling=args.scaling, steps=args.steps, warmup=args.warmup,
                                                 global_batch=total if args.scaling == "strong" else total * world, shards=shard)) + "\n").encode())
        dist.barrier()
        dist.destroy_process_group()
        return
    ndev = torch.cuda.device_count()             # (counting devices does not initialise the GPU)
    if ndev == 0:
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU implementation")
    if world > ndev and args.backend == "nccl":
        raise SystemExit(f"bench.py: --gpus {world} with RCCL needs {world} visible GPUs, found {ndev} "
                         "(use --backend gloo only to rehearse the multi-rank path on fewer GPUs)")
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)

    # proof that the collective backend really spans N ranks: an all-reduce of ones on the DEVICE (RCCL under the nccl backend) must
    # return N on every rank; it goes into the JSON line as `ranks_seen`
    ranks_seen = 1
    if world > 1:
        one = torch.ones(1, device="cuda")
        dist.all_reduce(one)
        ranks_seen = int(round(float(one.item())))
        if ranks_seen != world:
            raise SystemExit(f"bench.py: the {args.backend} all-reduce saw {ranks_seen} ranks, expected {world}")

    import q_learning_with_hjb_amd as pkg
    if rank == 0:
        pkg.build_library()                      # no-op when csrc/libhjbx.so is current (it ships prebuilt)
    if world > 1:
        dist.barrier()
    from q_learning_with_hjb_amd import _abi, _ops
    global ARITHMETIC
    ARITHMETIC = args.arithmetic
    _abi.set_option(_abi.OPT_MLP_ARITHMETIC, {"f32": 0, "bf16x3": 1, "f16x2": 2}[args.arithmetic])

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    K, W, reps = args.steps, args.warmup, max(1, args.reps)
    total = args.batch or HEADLINE_BATCH[args.system]
    if args.scaling == "strong":
        lo, hi = rank * total // world, (rank + 1) * total // world     # contiguous shards of one global batch
        B = hi - lo
        global_batch = total
    else:
        B, global_batch = total, total * world
    wl = make_workload(args.system, args.integrator, args.activation, B, rank, torch_mlp=args.torch_mlp)
    n, m, ctl, dyn = wl["n"], wl["m"], wl["ctl"], wl["dyn"]
    fused = ctl.fused_value_grad and not args.stepwise
    kernel_ms = {}

    if fused:
        walls, launches, live = time_fused(wl, K, W, reps, args.chunk, barrier, prewarm=args.prewarm)
        roofline = mfma_roofline(wl, launches, B)
        kernel_ms["k_vhjb_rollout_mfma_ms_per_launch"] = roofline["avg_launch_ms"]
    else:
        walls, live, roofline, kernel_ms = stepwise_loop(args, wl, K, W, reps, barrier)

    walls_t = torch.tensor(walls, device="cuda", dtype=torch.float64)
    live_t = torch.tensor([live], device="cuda", dtype=torch.int64)
    per_rank_ms = [float(np.median(walls)) / K * 1e3]
    if dist is not None:
        mine = torch.tensor([per_rank_ms[0]], device="cuda", dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)                          # every rank's own median (before the max over ranks below)
        per_rank_ms = [float(t.item()) for t in allr]
        dist.all_reduce(walls_t, op=dist.ReduceOp.MAX)      # per repetition: the slowest rank
        dist.all_reduce(live_t)
    walls = walls_t.cpu().numpy()
    live = int(live_t.item())
    elapsed = float(np.median(walls))
    value = live / elapsed

    out = dict(metric="env-steps/sec (batched HJB rollouts), cartpole batch=2^20" if args.system == "cartpole" else
               f"env-steps/sec (batched HJB rollouts), {args.system}", value=value, unit="env-steps/s", n_gpus=world,
               steps=K, warmup=W, ms_per_step=elapsed / K * 1e3, higher_is_better=True, scaling=args.scaling, vs_baseline=None,
               dtype={"f32": "f32", "bf16x3": "f32 (value-network products as exact 3-way bf16 splits on the bf16 MFMA, f32 accumulation)",
                      "f16x2": "f32 (value-network products as scaled 2-way f16 splits, 22 bits, on the f16 MFMA, f32 accumulation)"}[args.arithmetic],
               data="synthetic", reps=reps, ms_per_step_min=float(walls.min()) / K * 1e3, ms_per_step_max=float(walls.max()) / K * 1e3,
               ranks_seen=ranks_seen, per_rank_ms_per_step=per_rank_ms,
               config=dict(workload=wl["label"], batch_per_gpu=B, global_batch=global_batch,
                           state_dim=n, control_dim=m, integrator=args.integrator, mlp=f"{n}-128-128-64 {args.activation}, no bias",
                           value_grad=("persistent fused rollout kernel (MFMA value net + step)" if fused else
                                       "fused HIP MFMA kernel + step kernel" if ctl.fused_value_grad else "PyTorch-ROCm matmuls + step kernel"),
                           live_fraction=live / (global_batch * K), parallelism=f"env-shard x{world}, no data-path collective",
                           backend=(args.backend if world > 1 else None)),
               roofline=roofline, kernels=kernel_ms)

    if not args.no_secondary:
        sec = []
        if rank == 0 and world == 1 and fused:
            # the headline workload in the other (opt-in) arithmetics of the value network
            for other in ("f32", "bf16x3", "f16x2"):
                if other == args.arithmetic:
                    continue
                ARITHMETIC = other
                _abi.set_option(_abi.OPT_MLP_ARITHMETIC, {"f32": 0, "bf16x3": 1, "f16x2": 2}[other])
                ws, ls, lv = time_fused(wl, 100, 20, 3, 0, lambda: torch.cuda.synchronize(), prewarm=0.0)
                rf = mfma_roofline(wl, ls, B)
                sec.append(dict(name=f"the headline workload with --arithmetic {other}", value=lv / float(np.median(ws)), unit="env-steps/s",
                                ms_per_step=float(np.median(ws)) / 100 * 1e3, achieved=rf["achieved"], peak=rf["peak"], frac=rf["frac"], bound="mfma",
                                kernel=rf["kernel"], f32_equivalent=rf.get("f32_equivalent", rf["achieved"])))
            ARITHMETIC = args.arithmetic
            _abi.set_option(_abi.OPT_MLP_ARITHMETIC, {"f32": 0, "bf16x3": 1, "f16x2": 2}[args.arithmetic])
            for system, integ in (("acrobot", "euler"), ("quad2d", "euler"), ("nearhover", "euler"), ("nearhover", "rk4"), ("cartpole", "rk4")):
                if system == args.system and integ == args.integrator:
                    continue
                w2 = make_workload(system, integ, "relu", HEADLINE_BATCH[system], 1)
                ws, ls, lv = time_fused(w2, 40, 10, 3, 0, lambda: torch.cuda.synchronize(), prewarm=0.0)
                rf = mfma_roofline(w2, ls, w2["B"])
                sec.append(dict(name=f"fused vhjb rollout: {system} {integ} B=2^{int(np.log2(w2['B']))}", value=lv / float(np.median(ws)), unit="env-steps/s",
                                ms_per_step=float(np.median(ws)) / 40 * 1e3, achieved=rf["achieved"], peak=rf["peak"], frac=rf["frac"], bound="mfma",
                                live_fraction=lv / (w2["B"] * 40)))
                del w2
                torch.cuda.empty_cache()
            for act in ("tanh", "sin"):      # the notebooks' value networks (cartpole_balancing.ipynb cell 6; double_integrator_optimal_time.ipynb cell 5)
                w2 = make_workload("cartpole", "euler", act, 1 << 20, 1)
                ws, ls, lv = time_fused(w2, 40, 10, 3, 0, lambda: torch.cuda.synchronize(), prewarm=0.0)
                rf = mfma_roofline(w2, ls, w2["B"])
                sec.append(dict(name=f"fused vhjb rollout: cartpole euler B=2^20, {act} value network", value=lv / float(np.median(ws)), unit="env-steps/s",
                                ms_per_step=float(np.median(ws)) / 40 * 1e3, achieved=rf["achieved"], peak=rf["peak"], frac=rf["frac"], bound="mfma",
                                live_fraction=lv / (w2["B"] * 40)))
                del w2
                torch.cuda.empty_cache()
            sec += hbm_entry_points("cartpole")
            torch.cuda.empty_cache()
            sec.append(param_gradient_kernels("nearhover", 1 << 20))
            torch.cuda.empty_cache()
        def guarded(name, fn):
            """A secondary measurement must not cost the run its JSON line: an exception becomes an entry of its own.  (Every rank takes part
            in these three; an error raised on every rank alike -- a shape, a missing kernel -- is skipped by all of them together.)"""
            try:
                return fn()
            except Exception as exc:  # noqa: BLE001
                return dict(name=name, error=f"{type(exc).__name__}: {exc}")

        # configs[3]: 2^18 planar quadrotors IN TOTAL, sharded over the ranks
        sec_strong = guarded("strong scaling: fused vhjb rollout, quad2d", lambda: strong_scaling_line(world, rank, dist, barrier))
        # every rank takes part (all-reduce inside for G > 1)
        sec_opt = guarded("params_update (cartpole, 256 samples in total)", lambda: optimiser_step(world, dist))
        # configs[4]: the sharded full-batch learning step
        sec_big = guarded("params_update (nearhover, 2^20 samples per GPU)", lambda: optimiser_step(world, dist, "nearhover", (1 << 20) * world))
        if rank == 0:
            sec += [sec_strong, sec_opt, sec_big]
            out["secondary"] = sec

    if rank == 0:
        out["config"]["value_network_arithmetic"] = {
            "f32": "f32 MFMA (bitwise an fmaf chain)",
            "bf16x3": "float32 operands split exactly into 3 bf16 pieces, 6 piece products on the bf16 MFMA, f32 accumulation",
            "f16x2": "float32 operands scaled per environment by a power of two and rounded to 2 f16 pieces (22 significant bits), 3 piece products "
                     "on the f16 MFMA, f32 accumulation; inputs, outputs, layer 1, dynamics and costs are float32"}[args.arithmetic]
        out["parity_evidence"] = parity_evidence(args.arithmetic, args.system)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:     # the host-core baseline is reported at N = 1 only
        out["cpu_baseline"] = cpu_baseline(dyn, ctl, wl["x0"], args.cpu_sample_envs)
    if rank == 0:
        sys.stdout.flush()
        os.write(stdout_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def stepwise_loop(args, wl, K, W, reps, barrier):
    """Diagnostic modes (--stepwise / --torch-mlp): one value-gradient launch + one step launch per environment step."""
    from q_learning_with_hjb_amd import _ops
    ctl, dyn, n, B = wl["ctl"], wl["dyn"], wl["n"], wl["B"]
    sysh, task = dyn.system, ctl._task
    step_bytes_per_env = 4.0 * (3 * n + 2)
    RING = 64                                    # time-major log ring: (RING, B, n) states + costs + done flags
    traj = torch.empty((RING, B, n), device="cuda")
    cost = torch.empty((RING, B), device="cuda")
    done = torch.empty((RING, B), device="cuda")
    done_step = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    traj[0].copy_(wl["x0"])

    def step(t):
        s_, d_ = t % RING, (t + 1) % RING
        g = ctl.get_v_gradient(traj[s_])
        _ops.vhjb_step(sysh, task, t, T_INF, traj[s_], g, traj[d_], cost[s_], done[s_], done_step, integrator=dyn.integrator)

    for t in range(W):
        step(t)
    walls = []
    t_cur = W
    for _ in range(reps):                        # (consecutive K-step blocks of one trajectory: no state reset in this diagnostic mode)
        barrier()
        t0 = time.perf_counter()
        for t in range(t_cur, t_cur + K):
            step(t)
        barrier()
        walls.append(time.perf_counter() - t0)
        t_cur += K
    ds = done_step.long()
    live = int(torch.where(ds < 0, torch.full_like(ds, K), (ds - (t_cur - K)).clamp(min=0, max=K)).sum().item())

    def timed_run(fn, reps_=100, warm=20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for r in range(warm):
            fn(r)
        e0.record()
        for r in range(reps_):
            fn(warm + r)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps_

    g_hold = [ctl.get_v_gradient(traj[t_cur % RING])]
    vg_ms = timed_run(lambda r: g_hold.__setitem__(0, ctl.get_v_gradient(traj[(t_cur + r) % RING])))
    ds_scratch = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    st_ms = timed_run(lambda r: _ops.vhjb_step(sysh, task, t_cur + r, T_INF, traj[(t_cur + r) % RING], g_hold[0],
                                                traj[(t_cur + r + 1) % RING], cost[(t_cur + r) % RING], done[(t_cur + r) % RING], ds_scratch))
    kernel_ms = dict(value_grad=vg_ms, k_vhjb_step=st_ms, k_vhjb_step_GBs=step_bytes_per_env * B / (st_ms * 1e-3) / 1e9)
    if ctl.fused_value_grad:
        ach = wl["flops_per_env"] * B / (vg_ms * 1e-3) / 1e12
        roofline = dict(bound="mfma", kernel="k_value_grad_mfma (hjbx_value_grad_f32)", achieved=ach, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=ach / MFMA_F32_PEAK_TFLOPS, traffic=None, avg_launch_ms=vg_ms)
    else:
        ach = step_bytes_per_env * B / (st_ms * 1e-3) / 1e9
        roofline = dict(bound="hbm", kernel="k_vhjb_step (hjbx_vhjb_step_f32)", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                        traffic=None, avg_launch_ms=st_ms)
    return walls, live, roofline, kernel_ms


def cpu_baseline(dyn, ctl, x0, sample_envs):
    """The oracle's restatement of rollout_trajectory (env by env, value gradient per step) on the host
    cores, OpenMP over environments; f64 state like the reference's CPU rollout (`value`), and float32 (`value_f32`)."""
    from oracle import oracle as O
    vf = ctl.value_function_approximator
    Wts = [w.detach().cpu().numpy().astype(np.float64) for w in vf.weights]
    mlp = O.make_mlp(vf.features, vf._np["mean"], vf._np["std"], vf._np["xf"], vf.epsilon_scalar, activation=vf.activation)
    s = O.System.from_dynamics(dyn)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = O.threads(min(avail, 16))            # the GPU box's CPU share per GPU is 16 cores
    T = 50
    # calibrate on a small sample, then size the run for ~10 s
    xs = x0[:256].cpu().numpy().astype(np.float64)
    t0 = time.perf_counter(); r = O.vhjb_rollout(s, ctl._task, mlp, *Wts, xs, T, log=False); dt = time.perf_counter() - t0
    rate = max(r["live_steps"], 1) / dt
    nenv = sample_envs or int(min(x0.shape[0], max(512, rate * 10.0 / T)))
    xs = x0[:nenv].cpu().numpy().astype(np.float64)
    t0 = time.perf_counter(); r = O.vhjb_rollout(s, ctl._task, mlp, *Wts, xs, T, log=False); dt = time.perf_counter() - t0
    multi = r["live_steps"] / dt
    t0 = time.perf_counter(); r32 = O.vhjb_rollout(s, ctl._task, mlp, *Wts, xs, T, log=False, dtype=np.float32); dt32 = time.perf_counter() - t0
    O.threads(1)                                 # the reference's own execution model: one environment at a time, one thread
    n1 = max(64, nenv // (4 * cores))
    t0 = time.perf_counter(); r1 = O.vhjb_rollout(s, ctl._task, mlp, *Wts, xs[:n1], T, log=False); dt1 = time.perf_counter() - t0
    return dict(value=multi, unit="env-steps/s", cores=cores, kind="port",
                sample=f"{nenv} envs x {T} steps of the same workload (f64, OpenMP over envs), {dt:.1f} s",
                value_f32=r32["live_steps"] / dt32, sample_f32=f"same sample in float32, {dt32:.1f} s",
                single_thread_value=r1["live_steps"] / dt1, single_thread_sample=f"{n1} envs x {T} steps, {dt1:.1f} s")


if __name__ == "__main__":
    main()
