"""f32 parity of the kernel the headline number comes from (`k_vhjb_rollout_mfma`, hjbx_vhjb_rollout_f32) at the north-star
tolerance, on BASELINE configs[1] / [2] / [3] / [4] at their full batch sizes, in every value-network arithmetic (f32 MFMA = the library
default, and the opt-in bf16x3 / f16x2 split modes against the SAME bounds), forward Euler (the reference's integrator) and -- for the
near-hover quadcopter of configs[4] -- the fused residual + RK4 kernel as well.

Everything is compared with the f64 oracle fed the same float32-rounded inputs.  The bound is SELF-CALIBRATING (round 3): the oracle is
also compiled for float (liborc.so, `dtype=np.float32`: the same statements in float32 on the CPU), and the kernel's error distribution
must stay within 2x the CPU-float32 one -- max and p99.9 of err / (the element's own term scale).  The analytic per-element bound of
round 2 (1e-5 |want| + 1e-5 x sum of the |terms| of that element) is still evaluated, printed, and asserted for the Euler cases.

  (i)   teacher-forced single step (n_steps = 1), per ELEMENT: x', u, cost and the HJB residual.  dV/dx of a ReLU network is
        DISCONTINUOUS where a pre-activation changes sign, so any float32 evaluation (the reference's own JAX float32 network included)
        may take a unit within rounding of its kink on the other side than float64 does.  Environments with such a unit are NOT skipped:
        they are compared with the float64 network evaluated with the near-kink unit(s) FORCED to either side, and must match one of
        those evaluations within the same bound;
  (ii)  integer outputs: `done_step` BIT-EQUAL to the f64 oracle's for every environment whose f64 error coordinates keep a margin
        > DELTA from the observation box while it is alive; a mismatch outside that band is accepted only if the f64 loop, restarted at
        the step where the trajectories separate with the near-kink unit(s) of that state forced to the other side, reproduces the
        kernel's `done_step`;
  (iii) T = 200 closed loop under the LQR-embedded value network: median AND p99 of |x_f32 - x_f64| at EVERY step within 2x the same
        statistic of the CPU float32 loop; the median also within the analytic 1e-5 bound.

The numbers are printed (pytest -s shows them) and written to $HJBX_REPORT_DIR/f32_parity_report.json (default gpurun_out/).

controller/vhjb.py of the reference needs JAX: the oracle side of these tests is the restatement (PARITY UNPINNED).
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ANGLE_IDX, ARITHMETICS, ROOT, make_dynamics, make_vhjb_config, wrapped_diff
from netref import NetRef
from oracle import oracle as O
from parity_util import F32_ULP, FACTOR, abs_err, assert_within_cpu_yardstick, step_term_scales
from q_learning_with_hjb_amd import _abi, _ops
from q_learning_with_hjb_amd.controller.vhjb import VHJBController

pytestmark = pytest.mark.gpu


RTOL = 1e-5                      # BASELINE.json north_star: "trajectories matching the CPU reference to rtol 1e-5"
DELTA = 1e-3                     # (ii): margin to the observation box, in error-coordinate units
KINK = 1e-5                      # a unit is "at its kink" when |pre-activation| < KINK x (sum of |terms| of that unit);
                                 #      float32 accumulation of a 128-term pre-activation is good to ~1e-6 of that sum (worst case 128 x 6e-8 = 8e-6)
FULL = {"cartpole": 1 << 20, "acrobot": 1 << 20, "quad2d": 1 << 18, "nearhover": 1 << 20}   # configs[1], [2], [3], [4]
INTEG = {"euler": _abi.EULER, "rk4": _abi.RK4}
CASES = [("cartpole", "euler"), ("acrobot", "euler"), ("quad2d", "euler"), ("nearhover", "euler"), ("nearhover", "rk4")]   # configs[4] names the RK4 kernel
_report = {}


def _save_report():
    d = os.environ.get("HJBX_REPORT_DIR", os.path.join(ROOT, "gpurun_out"))
    try:
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "f32_parity_report.json")
        old = {}
        if os.path.exists(path):
            with open(path) as f:
                old = json.load(f)
        old.update(_report)
        with open(path, "w") as f:
            json.dump(old, f, indent=1, sort_keys=True)
    except OSError:
        pass


class arithmetic:
    """context manager: run the fused kernels in one value-network arithmetic (HJBX_OPT_MLP_ARITHMETIC)"""

    def __init__(self, name):
        self.mode = ARITHMETICS[name]

    def __enter__(self):
        self.prev = _abi.set_option(_abi.OPT_MLP_ARITHMETIC, self.mode)

    def __exit__(self, *exc):
        _abi.set_option(_abi.OPT_MLP_ARITHMETIC, self.prev)


def setup(name, weights, integ="euler"):
    d = make_dynamics(name)
    d.integrator = INTEG[integ]
    ctl = VHJBController(d, make_vhjb_config(name), dtype=torch.float32)
    vf = ctl.value_function_approximator
    if weights == "lqr":       # the bench's network: LQR value function embedded exactly + 5 % dense lecun-normal noise
        vf.load_quadratic(ctl.P, noise=0.05, generator=torch.Generator(device="cuda").manual_seed(1234))
    W = [w.detach().cpu().numpy().astype(np.float64) for w in vf.weights]     # the oracle sees the SAME float32 weights
    mlp = O.make_mlp(vf.features, vf._np["mean"], vf._np["std"], vf._np["xf"], vf.epsilon_scalar)
    return d, ctl, vf, mlp, W


def start_states(d, ctl, B, seed, frac, vel_frac=None):
    """x0 = xf + U(-1,1) * frac * box (box = the observation box, rates capped at 3): frac < 1 starts inside it.  `vel_frac` scales
    the second half of the state (the rates, in all five systems) separately."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    n = d.state_dim
    box = np.asarray(ctl.obs_max, np.float64).clip(max=3.0) * frac
    if vel_frac is not None:
        box[n // 2:] *= vel_frac / frac
    box = torch.as_tensor(box, dtype=torch.float32, device="cuda")
    xf = torch.as_tensor(np.asarray(ctl.xf, np.float64), dtype=torch.float32, device="cuda")
    u = torch.rand((B, n), generator=g, device="cuda") * 2 - 1
    return _ops.wrap(d.system, (xf + u * box).contiguous())


@pytest.mark.parametrize("weights", ["lqr", "random"])
@pytest.mark.parametrize("name,integ", CASES)
def test_teacher_forced_single_step_per_element(name, integ, weights):
    """(i) one closed-loop step of the fused MFMA kernel from the same float32 states, full batch, per element, every arithmetic."""
    d, ctl, vf, mlp, W = setup(name, weights, integ)
    B = FULL[name]
    n, m = d.get_dimension()
    ai = ANGLE_IDX[name]
    x = start_states(d, ctl, B, 11, 0.97)
    s = O.System.from_dynamics(d)
    xr = x.cpu().numpy().astype(np.float64)
    ds0 = np.full(B, -1, np.int32)
    # float64 oracle, CPU float32 oracle (same statements, float), and the float64 network with forced masks at the kinks
    _, g = O.value_grad(s, mlp, *W, xr)
    oxn, ou, oc, od, ods, ors = O.vhjb_step(s, ctl._task, 0, 1 << 30, xr, g, ds0, integrator=INTEG[integ])
    _, g32 = O.value_grad(s, mlp, *W, xr, dtype=np.float32)
    cxn, cu, cc, cd, cds, crs = O.vhjb_step(s, ctl._task, 0, 1 << 30, xr, g32, ds0, integrator=INTEG[integ], dtype=np.float32)
    net = NetRef.of(ctl, W, s)
    fw = net.forward(xr)
    assert np.abs(net.grad(fw) - g).max() <= 1e-9 * max(1.0, np.abs(g).max())          # the NumPy restatement == the C oracle
    _, gabs, _ = net.term_scales(fw)
    S = step_term_scales(name, d, ctl, s, xr, g, gabs, ou, oc, integ)
    live = ods < 0
    assert live.mean() > 0.99 and np.array_equal(cds, ods)
    c1, c2 = net.kink_candidates(fw, KINK)
    at_kink = live & (c1.any(1) | c2.any(1))
    clean = live & ~at_kink
    rows = np.nonzero(at_kink)[0]
    combos, overflow = net.forced_grads(fw, rows, KINK)
    forced = [O.vhjb_step(s, ctl._task, 0, 1 << 30, xr[rows], gc, ds0[rows], integrator=INTEG[integ]) for gc in combos]
    want = dict(x_next=oxn, u=ou, cost=oc, residual=ors)
    cpu32 = dict(x_next=cxn, u=cu, cost=cc, residual=crs)
    fidx = dict(x_next=0, u=1, cost=2, residual=5)
    del fw, c1, c2
    assert at_kink.mean() < 0.02

    for arith in ARITHMETICS:
        ds = torch.full((B,), -1, dtype=torch.int32, device="cuda")
        with arithmetic(arith):
            out = _ops.vhjb_rollout(d.system, ctl._task, vf.descriptor(), x, 1, 1 << 30, ds, integrator=INTEG[integ], log_traj=True, log_u=True,
                                    log_residual=True)
            torch.cuda.synchronize()
        got = dict(x_next=out["traj"][1].cpu().numpy().astype(np.float64), u=out["u"][0].cpu().numpy().astype(np.float64),
                   cost=out["cost"][0].cpu().numpy().astype(np.float64), residual=out["residual"][0].cpu().numpy().astype(np.float64))
        assert np.array_equal(ds.cpu().numpy(), ods), f"[{arith}] teacher-forced step: done_step differs"       # all start inside the box (frac < 1)
        assert np.array_equal(out["done"][0].cpu().numpy().astype(np.float64), od)
        print(f"\n[f32 parity (i), {arith}] {name} {integ} {weights} B={B}: {at_kink.mean():.3%} of the environments have a unit within {KINK:g} of its ReLU kink "
              f"(compared against the float64 network with that unit on either side); the rest against the CPU float32 yardstick:")
        rep = dict(B=B, kink_threshold=KINK, at_kink_fraction=float(at_kink.mean()), at_kink_more_than_3_units=overflow)
        thresholds = {}
        for key in ("x_next", "u", "cost", "residual"):
            wrapped = ai if key == "x_next" else ()
            sg, sc = assert_within_cpu_yardstick(f"{key}", got[key], cpu32[key], want[key], S[key], angle_idx=wrapped, keep=clean, report=rep)
            thresholds[key] = FACTOR * max(sc["max"], F32_ULP)
            # the analytic bound of round 2: |err| <= 1e-5 |want| + 1e-5 x (the element's term scale)
            err = abs_err(got[key], want[key], wrapped)
            ratio = err / (RTOL * np.abs(want[key]) + RTOL * S[key])
            rep[key]["analytic_max_ratio"] = float(ratio[clean].max())
            rep[key]["max_err"] = float(err[clean].max())
            print(f"        analytic bound 1e-5 (|want| + term scale): max err {err[clean].max():.2e}, max err / bound {ratio[clean].max():.3f}")
            if integ == "euler":
                assert ratio[clean].max() <= 1.0, f"[{arith}] {key}: max err / analytic bound {ratio[clean].max():.2f}"
        # environments at a kink: every quantity within the (calibrated) bound of ONE forced float64 evaluation
        if len(rows):
            worst = np.full(len(rows), np.inf)
            for f in forced:
                q = np.zeros(len(rows))
                for key in ("x_next", "u", "cost", "residual"):
                    e = abs_err(got[key][rows], f[fidx[key]], ai if key == "x_next" else ()) / S[key][rows] / thresholds[key]
                    q = np.maximum(q, e.reshape(len(rows), -1).max(1))
                worst = np.minimum(worst, q)
            n_bad = int((worst > 1.0).sum())
            plain = np.zeros(len(rows))
            for key in ("x_next", "u", "cost", "residual"):
                e = abs_err(got[key][rows], want[key][rows], ai if key == "x_next" else ()) / S[key][rows] / thresholds[key]
                plain = np.maximum(plain, e.reshape(len(rows), -1).max(1))
            rep.update(at_kink_envs=int(len(rows)), at_kink_on_the_other_side=int((plain > 1.0).sum()), at_kink_matching_no_side=n_bad,
                       at_kink_worst_ratio=float(worst.max()))
            print(f"    at a kink: {len(rows)} environments, {int((plain > 1.0).sum())} of them took a unit on the other side than float64 did, "
                  f"{n_bad} match no side (worst err / calibrated bound {worst.max():.2f})")
            assert n_bad == 0, f"[{arith}] {n_bad} at-kink environments match the float64 network for no side of their near-kink units"
        _report[f"teacher_forced/{arith}/{name}/{integ}/{weights}"] = rep
        _save_report()


def _margins(s, task_cfg, traj, ai):
    """min over coordinates of the distance of wrap(x - xf) to the observation box faces, per (t, env); negative = outside."""
    T1, B, n = traj.shape
    xf = np.asarray(task_cfg.xf, np.float64)
    e = O.wrap(s, (traj.reshape(-1, n) - xf[None, :])).reshape(T1, B, n)
    omin, omax = np.asarray(task_cfg.obs_min, np.float64), np.asarray(task_cfg.obs_max, np.float64)
    return np.minimum(omax[None, None, :] - e, e - omin[None, None, :]).min(-1)


def _kink_side_explains(net, s, ctl, mlp, W, ref_traj, got_traj, b, got_ds, T, integ, ai, scale):
    """A `done_step` mismatch of environment b outside the margin band: find the first step at which the kernel's trajectory leaves the float64
    one (by more than 1e-5 of the coordinate ranges), take the float64 state BEFORE it, force its near-kink unit(s) to every side, and
    continue the float64 loop from each: True if one of those loops ends at the kernel's `done_step`."""
    e_t = (np.abs(wrapped_diff(got_traj, ref_traj, ai)) / scale[None, :]).max(-1)
    jump = np.nonzero(e_t > 1e-5)[0]
    if not len(jump) or jump[0] == 0:
        return False, -1, np.inf
    t0 = int(jump[0]) - 1
    x0 = ref_traj[t0][None, :]
    fw = net.forward(x0)
    margin = float(net.kink_margin(fw)[0])
    combos, _ = net.forced_grads(fw, np.array([0]), 10 * KINK)
    for gc in combos:
        xn, _, _, _, ds1, _ = O.vhjb_step(s, ctl._task, t0, T, x0, gc, np.full(1, -1, np.int32), integrator=INTEG[integ])
        if ds1[0] >= 0:
            end = int(ds1[0])
        else:
            r = O.vhjb_rollout(s, ctl._task, mlp, *W, xn, T - (t0 + 1), integrator=INTEG[integ])
            end = int(r["done_step"][0]) + t0 + 1
        if end == got_ds:
            return True, t0, margin
    return False, t0, margin


@pytest.mark.parametrize("name,integ", CASES)
def test_done_step_bit_equal_outside_margin(name, integ):
    """(ii) `done_step` of the fused rollout == the f64 oracle's, bit for bit, for every environment that never comes within DELTA of
    a face of the observation box while alive (an environment inside that band can legitimately cross one step apart in float32)."""
    d, ctl, vf, mlp, W = setup(name, "lqr", integ)
    B, T = FULL[name], 30
    ai = ANGLE_IDX[name]
    x0 = start_states(d, ctl, B, 12, 1.04, vel_frac=0.3)    # some start outside the box, more leave during the 30 steps
    s = O.System.from_dynamics(d)
    x0r = x0.cpu().numpy().astype(np.float64)
    ref = O.vhjb_rollout(s, ctl._task, mlp, *W, x0r, T, integrator=INTEG[integ])
    c32 = O.vhjb_rollout(s, ctl._task, mlp, *W, x0r, T, integrator=INTEG[integ], dtype=np.float32, log=False)
    rs = ref["done_step"]
    mg = _margins(s, ctl, ref["traj"], ai)                                 # (T+1, B)
    alive = np.arange(T + 1)[:, None] <= rs[None, :]                       # steps at which the env's box test is evaluated
    near = ((np.abs(mg) <= DELTA) & alive).any(0)
    safe = ~near
    n_term = int((rs < T).sum())
    cpu_bad = int((safe & (c32["done_step"] != rs)).sum())
    assert 0.02 < n_term / B < 0.98, "the test needs both terminating and surviving environments"
    assert near.mean() < 0.05, "the margin band should filter out only a small fraction"
    net = NetRef.of(ctl, W, s)
    scale = np.maximum(np.abs(ref["traj"]).reshape(-1, d.state_dim).max(0), 1.0)
    for arith in ARITHMETICS:
        with arithmetic(arith):
            out = ctl.rollout_batch(x0, max_steps=T)
            torch.cuda.synchronize()
        ds = out["done_step"].cpu().numpy()
        bad = np.nonzero(safe & (ds != rs))[0]
        explained = 0
        if len(bad):
            trb = out["traj"][:, torch.as_tensor(bad, device="cuda"), :].cpu().numpy().astype(np.float64)
        for j, b in enumerate(bad):
            last = int(min(ds[b], rs[b]))
            ok, t0, mk = _kink_side_explains(net, s, ctl, mlp, W, ref["traj"][: last + 2, b, :], trb[: last + 2, j, :], b, int(ds[b]), T, integ, ai, scale)
            explained += bool(ok)
            print(f"    [{arith}] done_step mismatch outside the band: env {int(b)} got {int(ds[b])} want {int(rs[b])}; the trajectories separate after step {t0} "
                  f"(kink margin of that state {mk:.2e}); the float64 loop with that unit on the other side {'reproduces' if ok else 'does NOT reproduce'} the kernel's result")
        rep = dict(B=B, T=T, delta=DELTA, filtered_fraction=float(near.mean()), terminated_before_T=n_term / B,
                   mismatches_in_safe=int(len(bad)), mismatches_in_safe_reproduced_by_the_other_side_of_a_kink=int(explained),
                   mismatches_in_band=int((ds[near] != rs[near]).sum()), cpu_f32_mismatches_in_safe=cpu_bad,
                   cpu_f32_mismatches_in_band=int((c32["done_step"][near] != rs[near]).sum()))
        # reported, not asserted: the float32 drift of the trajectories over these 30 steps
        tr = out["traj"].cpu().numpy().astype(np.float64)
        err = np.abs(wrapped_diff(tr, ref["traj"], ai))[:, safe].max(-1)      # (T+1, safe)
        err = np.where(alive[:, safe], err, 0.0).max(0)
        rep.update(drift_median=float(np.median(err)), drift_p999=float(np.quantile(err, 0.999)), drift_max=float(err.max()))
        del tr, err
        _report[f"done_step/{arith}/{name}/{integ}"] = rep
        _save_report()
        print(f"\n[f32 parity (ii), {arith}] {name} {integ} B={B} T={T}: {near.mean():.3%} of the environments within {DELTA:g} of a box face (filtered), "
              f"{n_term / B:.1%} terminate before T; mismatches: {len(bad)} outside the band ({explained} reproduced by the other side of a kink; the CPU "
              f"float32 loop has {cpu_bad}), {rep['mismatches_in_band']} inside (CPU float32: {rep['cpu_f32_mismatches_in_band']}); drift median "
              f"{rep['drift_median']:.2e} p99.9 {rep['drift_p999']:.2e} max {rep['drift_max']:.2e}")
        assert explained == len(bad), f"[{arith}] {len(bad) - explained} done_step mismatches outside the margin band that no side of a ReLU kink reproduces"
        assert len(bad) <= max(4, 2 * cpu_bad), "kink events are a few per 2^20 environments and step: more mismatches than that is something else"


@pytest.mark.parametrize("name,integ", [("cartpole", "euler"), ("quad2d", "euler"), ("nearhover", "euler"), ("nearhover", "rk4")])
def test_closed_loop_error_curve_T200(name, integ):
    """(iii) 200 closed-loop steps (the reference's maximum_step) under the LQR-embedded value network, B = 2^16: error curve of the
    fused float32 rollout against the f64 oracle from the same float32 start states, next to the same curve of the CPU float32 loop."""
    d, ctl, vf, mlp, W = setup(name, "lqr", integ)
    B, T = 1 << 16, 200
    ai = ANGLE_IDX[name]
    x0 = start_states(d, ctl, B, 13, 0.5)
    s = O.System.from_dynamics(d)
    x0r = x0.cpu().numpy().astype(np.float64)
    ref = O.vhjb_rollout(s, ctl._task, mlp, *W, x0r, T, integrator=INTEG[integ])
    c32 = O.vhjb_rollout(s, ctl._task, mlp, *W, x0r, T, integrator=INTEG[integ], dtype=np.float32)
    rs = ref["done_step"]
    rng_k = np.abs(ref["traj"]).reshape(-1, d.state_dim).max(0)             # per-coordinate range over the whole run
    bound = RTOL * np.abs(ref["traj"]) + RTOL * np.maximum(rng_k, 1.0)[None, None, :]
    same_c = c32["done_step"] == rs
    ec = (np.abs(wrapped_diff(c32["traj"].astype(np.float64), ref["traj"], ai)) / bound).max(-1)        # (T+1, B): worst coordinate, in units of the bound
    curve_t = [1, 2, 5, 10, 20, 50, 100, 150, 200]
    for arith in ARITHMETICS:
        with arithmetic(arith):
            out = ctl.rollout_batch(x0, max_steps=T)
            torch.cuda.synchronize()
        ds = out["done_step"].cpu().numpy()
        same = ds == rs
        tr = out["traj"].cpu().numpy().astype(np.float64)
        err = np.abs(wrapped_diff(tr, ref["traj"], ai))                      # (T+1, B, n)
        eg = (err / bound).max(-1)
        es = err.max(-1)
        both = same & same_c
        med_g, med_c = np.median(eg[:, both], axis=1), np.median(ec[:, both], axis=1)
        p99_g, p99_c = np.quantile(eg[:, both], 0.99, axis=1), np.quantile(ec[:, both], 0.99, axis=1)
        curve = {str(t): dict(median=float(np.median(es[t][both])), p99=float(np.quantile(es[t][both], 0.99)), p999=float(np.quantile(es[t][both], 0.999)),
                              max=float(es[t][both].max()), median_over_bound=float(med_g[t]), p99_over_bound=float(p99_g[t]),
                              cpu_f32_median_over_bound=float(med_c[t]), cpu_f32_p99_over_bound=float(p99_c[t]),
                              within_bound=float((eg[t][both] <= 1).mean()), cpu_f32_within_bound=float((ec[t][both] <= 1).mean()))
                 for t in curve_t}
        tot_g = (out["cost"].double() * (torch.arange(T + 1, device="cuda")[:, None] <= out["done_step"][None, :])).sum(0).cpu().numpy()
        tot_r = (ref["cost"] * (np.arange(T + 1)[:, None] <= rs[None, :])).sum(0)
        tot_c = (c32["cost"].astype(np.float64) * (np.arange(T + 1)[:, None] <= c32["done_step"][None, :])).sum(0)
        rel_tot = np.abs(tot_g - tot_r)[both] / np.maximum(np.abs(tot_r[both]), 1e-12)
        rel_tot_c = np.abs(tot_c - tot_r)[both] / np.maximum(np.abs(tot_r[both]), 1e-12)
        rep = dict(B=B, T=T, done_step_agree=float(same.mean()), cpu_f32_done_step_agree=float(same_c.mean()), survive_to_T=float((rs == T).mean()), curve=curve,
                   trajectory_cost_rel_err_median=float(np.median(rel_tot)), trajectory_cost_rel_err_p999=float(np.quantile(rel_tot, 0.999)),
                   trajectory_cost_rel_err_max=float(rel_tot.max()), cpu_f32_trajectory_cost_rel_err_median=float(np.median(rel_tot_c)),
                   cpu_f32_trajectory_cost_rel_err_p999=float(np.quantile(rel_tot_c, 0.999)),
                   worst_step_median_kernel_over_cpu=float((med_g[1:] / np.maximum(med_c[1:], 1e-3)).max()),
                   worst_step_p99_kernel_over_cpu=float((p99_g[1:] / np.maximum(p99_c[1:], 1e-3)).max()))
        _report[f"closed_loop_T200/{arith}/{name}/{integ}"] = rep
        _save_report()
        print(f"\n[f32 parity (iii), {arith}] {name} {integ} B={B} T={T}: done_step agreement {same.mean():.5f} (CPU float32 {same_c.mean():.5f}); "
              "|x_f32 - x_f64| / (1e-5|x| + 1e-5 range), median / p99 of the kernel [of the CPU float32 loop]: " +
              ", ".join(f"t={t}: {c['median_over_bound']:.2f} / {c['p99_over_bound']:.2f} [{c['cpu_f32_median_over_bound']:.2f} / {c['cpu_f32_p99_over_bound']:.2f}]"
                        for t, c in curve.items()) +
              f"; trajectory cost rel err median {np.median(rel_tot):.1e} [{np.median(rel_tot_c):.1e}], p99.9 {np.quantile(rel_tot, 0.999):.1e} "
              f"[{np.quantile(rel_tot_c, 0.999):.1e}]")
        assert same.mean() > 0.999 and same.mean() >= same_c.mean() - 1e-3
        # every step: median and p99 within FACTOR x the CPU float32 loop's (floor: 1e-3 of the bound = 1e-8 relative)
        assert (med_g[1:] <= FACTOR * np.maximum(med_c[1:], 1e-3)).all(), f"[{arith}] median error exceeds {FACTOR} x the CPU float32 loop's at step {int((med_g[1:] / np.maximum(med_c[1:], 1e-3)).argmax()) + 1}"
        assert (p99_g[1:] <= FACTOR * np.maximum(p99_c[1:], 1e-3)).all(), f"[{arith}] p99 error exceeds {FACTOR} x the CPU float32 loop's at step {int((p99_g[1:] / np.maximum(p99_c[1:], 1e-3)).argmax()) + 1}"
        # the typical environment tracks the float64 loop within the analytic bound at every step
        assert med_g.max() <= 1.0, f"[{arith}] median closed-loop error exceeds the 1e-5 bound at step {int(med_g.argmax())} ({med_g.max():.2f}x)"
        assert es.max() < 0.5, "closed-loop error is not bounded"
        assert np.median(rel_tot) <= max(FACTOR * np.median(rel_tot_c), 1e-6) and rel_tot.max() < 5e-2
        del tr, err, eg, es
