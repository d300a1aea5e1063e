#!/usr/bin/env python3
"""Per-kernel timing of every HBM-bound entry point at the BASELINE batch sizes, as achieved GB/s of the
ALGORITHMIC bytes against the 8 TB/s HBM peak.  Run on the GPU box:

    python tools/kernel_bench.py [--systems cartpole,quad2d,...] [--batch 1048576] [--json out.json] [--pool-mb 640]

Every launch works on a different buffer set of a pool larger than 2 x the 256 MB Infinity Cache, cycled round-robin, so reads come
from HBM (round 1 relaunched on the same 16-40 MB and measured cache bandwidth).  Timing: one HIP-event pair around a run of
back-to-back launches (after warm-up).  Under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` the same launches give counter bytes
(tools/summarize_pmc.py)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import make_dynamics, make_vhjb_config  # noqa: E402
from q_learning_with_hjb_amd import _abi, _ops  # noqa: E402


def timed(fn, sets, reps=60, warm=12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for r in range(warm):
        fn(sets[r % len(sets)])
    e0.record()
    for r in range(reps):
        fn(sets[(warm + r) % len(sets)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def controller_for(name, d):
    if name == "linear":
        from q_learning_with_hjb_amd.controller.lqr import LQR
        return LQR(d, np.eye(2), np.eye(1))
    if name == "cartpole":
        from q_learning_with_hjb_amd.controller.cartpole_energy_shaping import CartpoleEnergyShapingController
        return CartpoleEnergyShapingController(d)
    if name == "acrobot":
        from q_learning_with_hjb_amd.controller.acrobot_energy_shaping import AcrobotEnergyShapingController
        return AcrobotEnergyShapingController(d)
    if name == "quad2d":
        from q_learning_with_hjb_amd.controller.quadrotors_model_based_controller import Quadrotors2DHoveringController
        return Quadrotors2DHoveringController(d, np.zeros(6), np.eye(6), np.eye(2))
    from q_learning_with_hjb_amd.controller.quadrotors_model_based_controller import NearHoverQuadcopterHoveringController
    return NearHoverQuadcopterHoveringController(d, np.zeros(10), np.eye(10), np.eye(3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--systems", default="linear,cartpole,acrobot,quad2d,nearhover")
    ap.add_argument("--batch", type=int, default=1 << 20)
    ap.add_argument("--T", type=int, default=200)
    ap.add_argument("--pool-mb", type=int, default=640)
    ap.add_argument("--no-rollouts", action="store_true", help="skip the fused closed-form rollouts (counter runs)")
    ap.add_argument("--rows", type=int, default=0, help="HJBX_OPT_STREAM_ROWS: rows per thread of the streaming kernels (0 = library default)")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    B = args.batch
    rows = []
    _abi.set_option(_abi.OPT_STREAM_ROWS, args.rows)
    # reference point: a plain device copy of the same size class, timed the same way (rotating buffers)
    nset = max(2, int(np.ceil(args.pool_mb * (1 << 20) / (8.0 * B * 4))))
    cps = [dict(a=torch.randn((B, 4), device="cuda"), b=torch.empty((B, 4), device="cuda")) for _ in range(nset)]
    secs = timed(lambda s: s["b"].copy_(s["a"]), cps)
    rows.append(dict(system="-", kernel="torch copy 16 MB -> 16 MB", us=secs * 1e6, bytes_per_env=32, GBs=32 * B / secs / 1e9, frac_of_8TBs=32 * B / secs / 1e9 / 8000,
                     env_steps_per_s=B / secs, rotating_sets=nset))
    del cps
    torch.cuda.empty_cache()
    for name in args.systems.split(","):
        d = make_dynamics(name)
        cfg = make_vhjb_config(name)
        n, m = d.get_dimension()
        task = _abi.make_task(n, m, cfg.Q, cfg.R, np.eye(n) * 2, cfg.xf, cfg.uf, cfg.obs_min, cfg.obs_max, cfg.epsilon)
        gen = torch.Generator(device="cuda").manual_seed(0)
        xf = torch.as_tensor(np.asarray(cfg.xf, np.float32), device="cuda")
        nsets = max(2, int(np.ceil(args.pool_mb * (1 << 20) / (4.0 * B * (4 * n + m + 3)))))

        def mk():
            x = (xf + (torch.rand((B, n), generator=gen, device="cuda") - 0.5) * 0.5).contiguous()
            return dict(x=x, u=torch.zeros((B, m), device="cuda"), g=torch.randn((B, n), generator=gen, device="cuda"),
                        done=torch.zeros((B,), device="cuda"), xn=torch.empty_like(x), c=torch.empty(B, device="cuda"), dn=torch.empty(B, device="cuda"),
                        ds=torch.full((B,), -1, dtype=torch.int32, device="cuda"), uo=torch.empty((B, m), device="cuda"))
        sets = [mk() for _ in range(nsets)]
        ctrl = controller_for(name, d)
        desc = ctrl._descriptor()

        def add(kernel, secs, bytes_per_env, steps=1):
            gbs = bytes_per_env * B * steps / secs / 1e9
            rows.append(dict(system=name, kernel=kernel, us=secs * 1e6, bytes_per_env=bytes_per_env, GBs=gbs, frac_of_8TBs=gbs / 8000.0,
                             env_steps_per_s=B * steps / secs, rotating_sets=nsets))

        add("simulate (euler)", timed(lambda s: _ops.simulate(d.system, s["x"], s["u"], _abi.EULER, out=s["xn"]), sets), 4 * (2 * n + m))
        add("simulate (rk4)", timed(lambda s: _ops.simulate(d.system, s["x"], s["u"], _abi.RK4, out=s["xn"]), sets), 4 * (2 * n + m))
        add("vhjb_step (euler)", timed(lambda s: _ops.vhjb_step(d.system, task, 0, 1 << 30, s["x"], s["g"], s["xn"], s["c"], s["dn"], s["ds"]), sets), 4 * (3 * n + 3))
        add("vhjb_step (rk4)", timed(lambda s: _ops.vhjb_step(d.system, task, 0, 1 << 30, s["x"], s["g"], s["xn"], s["c"], s["dn"], s["ds"], integrator=_abi.RK4), sets),
            4 * (3 * n + 3))
        add("hjb_residual fwd+bwd+sums", timed(lambda s: _ops.hjb_residual(d.system, task, s["x"], s["g"], s["done"], want_loss=False), sets), 4 * (3 * n + 1))
        add("controller", timed(lambda s: _ops.controller(d.system, desc, s["x"]), sets), 4 * (n + m))
        if args.no_rollouts:
            continue
        del sets
        torch.cuda.empty_cache()
        T = args.T
        Bl = min(B, 1 << 18) if n >= 6 else B          # keep the (T+1, B, n) log under ~5 GB
        x0 = (xf + (torch.rand((Bl, n), generator=gen, device="cuda") - 0.5) * 0.5).contiguous()
        one = [None]
        s = timed(lambda _: _ops.rollout_feedback(d.system, desc, x0, T, task=task, terminate=False, log_traj=True, log_u=False, log_cost=True), one,
                  reps=5, warm=2)
        rows.append(dict(system=name, kernel=f"rollout_feedback T={T} logging x,cost (B={Bl})", us=s * 1e6, bytes_per_env=4 * (n + 1) + 4 * n / T,
                         GBs=(4 * (n + 1) * (T + 1) + 4 * n) * Bl / s / 1e9, frac_of_8TBs=(4 * (n + 1) * (T + 1) + 4 * n) * Bl / s / 1e9 / 8000,
                         env_steps_per_s=Bl * T / s, rotating_sets=1))
        s = timed(lambda _: _ops.rollout_feedback(d.system, desc, x0, T, task=None, log_traj=False), one, reps=5, warm=2)
        rows.append(dict(system=name, kernel=f"rollout_feedback T={T} no logs (B={Bl})", us=s * 1e6, bytes_per_env=8 * n / T, GBs=8 * n * Bl / s / 1e9,
                         frac_of_8TBs=8 * n * Bl / s / 1e9 / 8000, env_steps_per_s=Bl * T / s, rotating_sets=1))
    print(f"{'system':10s} {'kernel':52s} {'us':>10s} {'B/env':>7s} {'GB/s':>9s} {'%HBM':>6s} {'env-steps/s':>12s}")
    for r in rows:
        print(f"{r['system']:10s} {r['kernel']:52s} {r['us']:10.1f} {r['bytes_per_env']:7.1f} {r['GBs']:9.1f} {100*r['frac_of_8TBs']:6.1f} {r['env_steps_per_s']:12.3e}")
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
