#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched VHJB closed loop on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1]): cartpole balancing, VHJB controller with the
4->128->128->64 value network, B = 2^20 float32 environments PER GPU (weak scaling; environments are
independent, no data-path collective).  One "step" = one closed-loop environment step for the whole
batch, i.e. one iteration of rollout_trajectory's loop (reference controller/vhjb.py:175-186) for B
environments: value gradient of the current states -> HJB-optimal control -> running cost -> bounds /
termination -> forward-Euler step -> log (x, cost, done).  States stay resident in HBM.

The value network carries synthetic "trained" weights (the LQR value function embedded exactly, plus
5 % dense lecun-normal noise so no weight is zero), so environments stay inside the observation box like
the reference's trained policy (average trajectory length 200/200, examples/cartpole_balancing.ipynb
cell 10); `value` counts LIVE environment steps only.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant hand-written kernel, timed live with
HIP events on the launch stream; `cpu_baseline` is the CPU oracle (oracle/, the port of the reference's
batch-1 loop) on this box's host cores over a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3  # dense f32-input MFMA peak (same guide)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1 << 20, help="environments per GPU")
    ap.add_argument("--system", default="cartpole", choices=["cartpole", "acrobot", "quad2d", "nearhover", "linear"],
                    help="default cartpole = BASELINE configs[1]; quad2d / nearhover = the VHJB loops of configs[3] / configs[4]")
    ap.add_argument("--integrator", default="euler", choices=["euler", "rk4"], help="euler = the reference's integrator (parity mode)")
    ap.add_argument("--activation", default="relu", choices=["relu", "tanh"], help="relu = controller/vhjb.py (the BASELINE workload); tanh = the cartpole notebook's network")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the multi-rank path on one GPU)")
    ap.add_argument("--torch-mlp", action="store_true", help="value gradient through PyTorch matmuls instead of the fused kernel")
    ap.add_argument("--stepwise", action="store_true", help="two launches per step (value_grad + vhjb_step) instead of the persistent rollout kernel")
    ap.add_argument("--cpu-sample-envs", type=int, default=0, help="0 = auto (about 10-20 s of CPU work)")
    return ap.parse_args()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU implementation")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import q_learning_with_hjb_amd as pkg
    if local_rank == 0:
        pkg.build_library()                      # no-op when csrc/libhjbx.so is current (it ships prebuilt)
    if world > 1:
        dist.barrier()
    from q_learning_with_hjb_amd import _abi, _ops
    from q_learning_with_hjb_amd.configs import defaults as D
    from q_learning_with_hjb_amd.controller.vhjb import VHJBController
    from q_learning_with_hjb_amd.dynamics.acrobot import Acrobot
    from q_learning_with_hjb_amd.dynamics.cartpole import Cartpole
    from q_learning_with_hjb_amd.dynamics.linear import LinearDynamics
    from q_learning_with_hjb_amd.dynamics.quadrotors import NearHoverQuadcopter, Quadrotors2D

    B, K, W = args.batch, args.steps, args.warmup
    dyn, ccfg, label = {
        "cartpole": lambda: (Cartpole(D.cartpole_dynamics_config()), D.cartpole_vhjb_config(), "cartpole balancing + vhjb controller (BASELINE configs[1])"),
        "acrobot": lambda: (Acrobot(D.acrobot_dynamics_config(x0_mean=[np.pi, 0, 0, 0], x0_std=[0.5, 0.5, 1, 1])), D.acrobot_vhjb_config(),
                            "acrobot at the upright + vhjb controller (BASELINE configs[2], closed-loop part)"),
        "quad2d": lambda: (Quadrotors2D(D.quadrotors2d_dynamics_config()), D.quadrotors2d_vhjb_config(), "Quadrotors2D hovering + vhjb controller (BASELINE configs[3])"),
        "nearhover": lambda: (NearHoverQuadcopter(D.near_hover_dynamics_config()), D.near_hover_vhjb_config(), "10-D near-hover quadcopter + vhjb controller (BASELINE configs[4])"),
        "linear": lambda: (LinearDynamics(D.linear_dynamics_config()), D.linear_vhjb_config(), "double integrator + vhjb controller"),
    }[args.system]()
    if args.integrator == "rk4":
        dyn.integrator = _abi.RK4
    ctl = VHJBController(dyn, ccfg, fused_value_grad=not args.torch_mlp, activation=args.activation)
    vf = ctl.value_function_approximator
    wgen = torch.Generator(device="cuda"); wgen.manual_seed(1234)
    vf.load_quadratic(ctl.P, noise=0.05, generator=wgen)
    gen = torch.Generator(device="cuda"); gen.manual_seed(rank)
    x0 = dyn.get_initial_state(B, generator=gen)
    if args.system in ("quad2d", "nearhover"):
        x0 = (x0 * 0.5).contiguous()             # start inside the observation box (the stock x0 box is as wide as it)

    n, m = dyn.get_dimension()
    sysh, task = dyn.system, ctl._task
    T_max = 1 << 30                              # no forced termination inside the timed region
    flops_per_env = 4.0 * (n * 128 + 128 * 128 + 128 * 64)          # value net fwd + input-grad MACs x 2
    step_bytes_per_env = 4.0 * (3 * n + 2)                          # step kernel: read x, gradV; write x', cost, done
    done_step = torch.full((B,), -1, dtype=torch.int32, device="cuda")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fused = ctl.fused_value_grad and not args.stepwise
    kernel_ms = {}
    if fused:
        # ---- the whole closed loop in persistent launches of <= CHUNK steps (hjbx_vhjb_rollout_f32) ----------------
        # launches of 100 steps: the grid is one persistent workgroup per CU with equal shares, so a CU that is briefly unavailable when a
        # launch starts makes that launch wait for a second round (seen on shared hosts: ~2 of 60 runs came out 1.9x slow with one
        # 200-step launch); shorter launches bound what such an event can cost, for one more launch per 200 steps (-0.3 %)
        CHUNK = 100
        desc = vf.descriptor()
        x_cur = x0
        t = 0

        def run(nsteps, events=None):
            nonlocal x_cur, t
            left, out = nsteps, None
            while left > 0:
                k = min(CHUNK, left)
                if events is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                out = _ops.vhjb_rollout(sysh, task, desc, x_cur, k, T_max, done_step, t_first=t, integrator=dyn.integrator, log_traj=True,
                                        want_x_out=True)
                if events is not None:
                    e1.record()
                    events.append((e0, e1, k))
                x_cur = out["x_out"]
                t += k
                left -= k
            return out

        # device pre-warm (not the workload: scratch state, results discarded): ~0.3 s of the same kernel so the clocks have ramped
        # before the W warm-up steps start (the first launch of a fresh process runs ~15 % below the settled rate)
        scratch_ds = done_step.clone()
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < 0.3:
            _ops.vhjb_rollout(sysh, task, desc, x0, 25, T_max, scratch_ds, integrator=dyn.integrator, log_traj=False)
            torch.cuda.synchronize()
        del scratch_ds
        run(W)
        barrier()
        evs = []
        t0 = time.perf_counter()
        run(K, evs)
        barrier()
        elapsed = time.perf_counter() - t0
        launch_ms = sum(a.elapsed_time(b) for a, b, _ in evs) / len(evs)
        steps_per_launch = sum(k for _, _, k in evs) / len(evs)
        kernel_ms["k_vhjb_rollout_mfma"] = launch_ms
        roofline = dict(bound="mfma", kernel="k_vhjb_rollout_mfma (hjbx_vhjb_rollout_f32)",
                        achieved=flops_per_env * B * steps_per_launch / (launch_ms * 1e-3) / 1e12, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
                        traffic=None, avg_launch_ms=launch_ms, steps_per_launch=steps_per_launch)
    else:
        # ---- one value-gradient launch + one step launch per environment step ---------------------------------------
        RING = 64                                    # time-major log ring: (RING, B, n) states + costs + done flags
        traj = torch.empty((RING, B, n), device="cuda")
        cost = torch.empty((RING, B), device="cuda")
        done = torch.empty((RING, B), device="cuda")
        traj[0].copy_(x0)

        def step(t):
            s_, d_ = t % RING, (t + 1) % RING
            g = ctl.get_v_gradient(traj[s_])
            _ops.vhjb_step(sysh, task, t, T_max, traj[s_], g, traj[d_], cost[s_], done[s_], done_step, integrator=dyn.integrator)

        for t in range(W):
            step(t)
        barrier()
        t0 = time.perf_counter()
        for t in range(W, W + K):
            step(t)
        barrier()
        elapsed = time.perf_counter() - t0

        # per-kernel launch durations: one event pair brackets a run of back-to-back launches of ONE kernel (an event
        # per launch would put the event's own queue packet into every measurement)
        def timed_run(fn, reps=100, warm=20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for r in range(warm):
                fn(r)
            e0.record()
            for r in range(reps):
                fn(warm + r)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        t_base = W + K
        g_hold = [ctl.get_v_gradient(traj[t_base % RING])]
        vg_ms = timed_run(lambda r: g_hold.__setitem__(0, ctl.get_v_gradient(traj[(t_base + r) % RING])))
        ds_scratch = torch.full((B,), -1, dtype=torch.int32, device="cuda")   # keeps the real done_step untouched
        st_ms = timed_run(lambda r: _ops.vhjb_step(sysh, task, t_base + r, T_max, traj[(t_base + r) % RING], g_hold[0],
                                                    traj[(t_base + r + 1) % RING], cost[(t_base + r) % RING], done[(t_base + r) % RING],
                                                    ds_scratch))
        kernel_ms = dict(value_grad=vg_ms, k_vhjb_step=st_ms, k_vhjb_step_GBs=step_bytes_per_env * B / (st_ms * 1e-3) / 1e9)
        if ctl.fused_value_grad:
            roofline = dict(bound="mfma", kernel="k_value_grad_mfma (hjbx_value_grad_f32)", achieved=flops_per_env * B / (vg_ms * 1e-3) / 1e12,
                            peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s", traffic=None, avg_launch_ms=vg_ms)
        else:
            roofline = dict(bound="hbm", kernel="k_vhjb_step (hjbx_vhjb_step_f32)", achieved=step_bytes_per_env * B / (st_ms * 1e-3) / 1e9,
                            peak=HBM_PEAK_GBS, unit="GB/s", traffic=None, avg_launch_ms=st_ms)

    if dist is not None:
        tt = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # live environment steps inside the timed window [W, W+K)
    ds = done_step.long()
    live = torch.where(ds < 0, torch.full_like(ds, K), (ds - W).clamp(min=0, max=K)).sum()
    if dist is not None:
        dist.all_reduce(live)
    live = int(live.item())
    value = live / elapsed

    roofline["frac"] = roofline["achieved"] / roofline["peak"]
    # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE x2 + WRITE_SIZE, gfx950 correction);
    # the persistent kernel's traffic is recorded per environment step and scaled to this run's steps per launch
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile):
        with open(tfile) as f:
            rec = json.load(f).get(roofline["kernel"].split(" ")[0])
        if isinstance(rec, dict):
            roofline["traffic"] = rec["bytes_per_env_step"] * B * roofline.get("steps_per_launch", 1)
        elif rec is not None:
            roofline["traffic"] = rec * (B / float(1 << 20))
    other = kernel_ms

    out = dict(metric="env-steps/sec (batched HJB rollouts), cartpole batch=2^20" if args.system == "cartpole" else
               f"env-steps/sec (batched HJB rollouts), {args.system}", value=value, unit="env-steps/s", n_gpus=world,
               steps=K, warmup=W, ms_per_step=elapsed / K * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32",
               data="synthetic",
               config=dict(workload=label, batch_per_gpu=B, global_batch=B * world,
                           state_dim=n, control_dim=m, integrator=args.integrator, mlp=f"{n}-128-128-64 {args.activation}, no bias",
                           value_grad=("persistent fused rollout kernel (MFMA value net + step)" if fused else
                                       "fused HIP MFMA kernel + step kernel" if ctl.fused_value_grad else "PyTorch-ROCm matmuls + step kernel"),
                           live_fraction=live / (B * world * K), parallelism=f"env-shard x{world}, no data-path collective"),
               roofline=roofline, kernels=other)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:     # the host-core baseline is reported at N = 1 only
        out["cpu_baseline"] = cpu_baseline(dyn, ctl, x0, args.cpu_sample_envs)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(dyn, ctl, x0, sample_envs):
    """The oracle's restatement of rollout_trajectory (env by env, value gradient per step) on the host
    cores, OpenMP over environments; f64 state like the reference's CPU rollout."""
    from oracle import oracle as O
    vf = ctl.value_function_approximator
    Wts = [w.detach().cpu().numpy().astype(np.float64) for w in vf.weights]
    mlp = O.make_mlp(vf.features, vf._np["mean"], vf._np["std"], vf._np["xf"], vf.epsilon_scalar, activation=vf.activation)
    s = O.System.from_dynamics(dyn)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = O.threads(min(avail, 16))            # the GPU box's CPU share per GPU is 16 cores
    T = 50
    # calibrate on a small sample, then size the run for ~12 s
    xs = x0[:256].cpu().numpy().astype(np.float64)
    t0 = time.perf_counter(); r = O.vhjb_rollout(s, ctl._task, mlp, *Wts, xs, T, log=False); dt = time.perf_counter() - t0
    rate = max(r["live_steps"], 1) / dt
    nenv = sample_envs or int(min(x0.shape[0], max(512, rate * 12.0 / T)))
    xs = x0[:nenv].cpu().numpy().astype(np.float64)
    t0 = time.perf_counter(); r = O.vhjb_rollout(s, ctl._task, mlp, *Wts, xs, T, log=False); dt = time.perf_counter() - t0
    multi = r["live_steps"] / dt
    O.threads(1)                                 # the reference's own execution model: one environment at a time, one thread
    n1 = max(64, nenv // (4 * cores))
    t0 = time.perf_counter(); r1 = O.vhjb_rollout(s, ctl._task, mlp, *Wts, xs[:n1], T, log=False); dt1 = time.perf_counter() - t0
    return dict(value=multi, unit="env-steps/s", cores=cores, kind="port",
                sample=f"{nenv} envs x {T} steps of the same workload (f64, OpenMP over envs), {dt:.1f} s",
                single_thread_value=r1["live_steps"] / dt1, single_thread_sample=f"{n1} envs x {T} steps, {dt1:.1f} s")


if __name__ == "__main__":
    main()
