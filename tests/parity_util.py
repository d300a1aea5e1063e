"""Shared comparison helpers of the GPU parity tests (TEST INFRASTRUCTURE).

Two kinds of bound, both PER ELEMENT (nothing batch-wide, no hidden factors):

  * `check(got, want, tol, scale)`: |got - want| <= tol |want| + tol scale, where `scale` is an array broadcastable to the data holding the sum of
    the magnitudes of the element's own terms (the forward-error bound of a float sum is a few ulps of THAT, also when the value cancels);
  * `assert_within_cpu_yardstick(...)`: a float32 kernel is judged against the CPU oracle compiled for float (the same statements in
    float32, gcc, -ffp-contract=off) -- the NEUTRAL yardstick: both are float32 evaluations of the same formulas from the same inputs, so
    the errors against the float64 oracle must have the same distribution.  Asserted: max and p99.9 of err / scale of the kernel
    <= FACTOR x the same statistic of the CPU float32 evaluation (FACTOR = 2), with a floor of one float32 ulp of the scale.
"""
import numpy as np
import torch

from conftest import ANGLE_IDX, wrapped_diff

F32_ULP = 2.0 ** -24
FACTOR = 2.0


def to_np(a):
    return a.detach().cpu().numpy().astype(np.float64) if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)


def abs_err(got, want, angle_idx=()):
    got, want = to_np(got), to_np(want)
    assert got.shape == want.shape, (got.shape, want.shape)
    return np.abs(wrapped_diff(got, want, angle_idx)) if len(angle_idx) else np.abs(got - want)


def check(got, want, tol, scale=None, angle_idx=(), max_bad_frac=0.0):
    """Per element: |got - want| <= tol |want| + tol scale (scale: per-element term magnitudes, broadcastable; default 1)."""
    d = abs_err(got, want, angle_idx)
    want = to_np(want)
    bound = tol * np.abs(want) + tol * (1.0 if scale is None else np.broadcast_to(np.asarray(scale, np.float64), want.shape))
    bad = d > bound
    ratio = np.divide(d, bound, out=np.where(d > 0, np.inf, 0.0), where=bound > 0)
    assert bad.mean() <= max_bad_frac, (f"max err / bound {ratio.max():.3f} (rtol = atol / scale = {tol:g}), {bad.mean():.2%} bad, at {np.argwhere(bad)[:4].tolist()}")
    return float(ratio.max()) if ratio.size else 0.0


def ratio_stats(err, scale):
    scale = np.broadcast_to(np.asarray(scale, np.float64), err.shape)
    q = np.divide(err, scale, out=np.where(err > 0, np.inf, 0.0), where=scale > 0)
    if q.size == 0:
        return dict(max=0.0, p999=0.0)
    return dict(max=float(q.max()), p999=float(np.quantile(q, 0.999)))


def assert_within_cpu_yardstick(label, got, cpu32, want, scale, angle_idx=(), keep=None, factor=FACTOR, report=None):
    """got: float32 kernel result; cpu32: the oracle's float32 evaluation of the same statements; want: the float64 oracle.
    keep: boolean mask over the leading axis (environments compared); scale: per-element term magnitudes."""
    eg, ec = abs_err(got, want, angle_idx), abs_err(cpu32, want, angle_idx)
    scale = np.broadcast_to(np.asarray(scale, np.float64), eg.shape)
    if keep is not None:
        eg, ec, scale = eg[keep], ec[keep], scale[keep]
    sg, sc = ratio_stats(eg, scale), ratio_stats(ec, scale)
    line = (f"{label}: err / term scale  kernel max {sg['max']:.2e} p99.9 {sg['p999']:.2e} | CPU float32 max {sc['max']:.2e} p99.9 {sc['p999']:.2e} "
            f"| kernel / CPU: max {sg['max'] / max(sc['max'], F32_ULP):.2f} p99.9 {sg['p999'] / max(sc['p999'], F32_ULP):.2f}")
    print("    " + line)
    if report is not None:
        report[label] = dict(kernel=sg, cpu_f32=sc)
    assert sg["max"] <= factor * max(sc["max"], F32_ULP), line
    assert sg["p999"] <= factor * max(sc["p999"], F32_ULP), line
    return sg, sc


def step_term_scales(name, d, ctl, s, xr, g, gabs, ou, oc, integ="euler"):
    """`ctl`: anything with R, R_inv, uf, epsilon (a VHJBController, or a namespace).  gabs >= |g|: the term scale of dV/dx (|g| itself when the
    gradient is an input).  Per element: the sum of the magnitudes of the terms that make up u, x', cost and the residual of ONE closed-loop step (float64),
    each including what the allowed error of the quantities it is computed from contributes.  bound = RTOL x these."""
    from oracle import oracle as O
    n, m = d.get_dimension()
    dt = float(d.dt)
    umax = np.maximum(np.abs(d.umin), np.abs(d.umax)).astype(np.float64)
    f1, f2 = O.affine(s, xr)
    ai = ANGLE_IDX[name]
    Rinv = np.asarray(ctl.R_inv, np.float64).reshape(m, m)
    R = np.asarray(ctl.R, np.float64).reshape(m, m)
    eps = float(ctl.epsilon)
    # u_j = clip(-1/2 sum_k Rinv_jj f2_kj g_k + uf_j), g = dV/dx: sums that cancel (acrobot: network terms ~1e5 for |u| <= 25), so the float32
    # forward error is a few ulps of the sum of the MAGNITUDES of the terms, and the result lives in [umin, umax]
    S_u = umax[None, :] + 0.5 * np.einsum("jj,bkj,bk->bj", np.abs(Rinv), np.abs(f2), gabs)
    # x'_k = wrap(x_k + dt (f1_k + sum_j f2_kj u_j)): own terms |x_k| + dt |f1_k| [+ pi for an angle: the wrap is (th + pi) mod 2 pi - pi],
    # plus what the control error contributes through dt |f2_kj|.  RK4: the stage derivatives are averaged with weights (1, 2, 2, 1) / 6; their
    # magnitudes are those of stage 1 to first order in dt, which is all a weighting needs
    S_x = np.abs(xr) + dt * np.abs(f1) + dt * np.einsum("bkj,bj->bk", np.abs(f2), S_u)
    S_x[:, ai] += np.pi
    if integ == "rk4":
        xd1 = O.dynamics_step(s, xr, ou)
        S_x += dt * np.abs(xd1)
    # cost = dt (e'Qe + du'R du): non-negative terms for the diagonal Q, R of these configs, so relative accuracy holds except for what
    # the control error contributes: d cost = 2 dt |R du| du_err; + dt (one cost unit x dt)
    du = ou - np.asarray(ctl.uf, np.float64)[None, :]
    S_c = 2 * dt * np.einsum("bj,bj->b", np.abs(du @ R.T), S_u) + dt
    # residual r = gradV . xdot / (l + eps) + 1: gradV . xdot cancels (it is ~ -l near the optimum): own terms sum_k |g_k xdot_k| / (l + eps),
    # plus the control error through |(f2' g)_j| / (l + eps) and through l itself
    xd = O.dynamics_step(s, xr, ou)
    l = oc / dt
    f2tg = np.abs(np.einsum("bkj,bk->bj", f2, g))
    vdot_abs = np.abs((g * xd).sum(1))
    dl = 2 * np.einsum("bj,bj->b", np.abs(du @ R.T), S_u)
    S_r = (1.0 + (gabs * np.abs(xd)).sum(1) / (l + eps)) + (np.einsum("bj,bj->b", f2tg, S_u) + vdot_abs * dl / (l + eps)) / (l + eps)
    return dict(x_next=S_x, u=S_u, cost=S_c, residual=S_r)


def residual_term_scales(d, task, s, xr, gr, mode):
    """Per element: sum of the |terms| of the HJB residual loss_i (B,) and of d loss_i / d gradV (B, n) for an ARBITRARY gradient input
    (SURVEY A.3 written with magnitudes; the clip is taken as inactive everywhere: an upper bound).  `task`: the ctypes hjbx_task."""
    from oracle import oracle as O
    n, m = d.get_dimension()
    R = np.array(task.R[: m * m], np.float64).reshape(m, m)
    Rinv = np.array(task.Rinv[: m * m], np.float64).reshape(m, m)
    uf = np.array(task.uf[:m], np.float64)
    eps = float(task.eps)
    f1, f2 = O.affine(s, xr)
    u = O.control_from_grad(s, task, xr, gr)
    l = O.running_cost(s, task, xr, u)
    A = np.abs(f1) + np.einsum("bkj,bj->bk", np.abs(f2), np.abs(u))                   # |terms| of xdot_k
    Tv = (np.abs(gr) * A).sum(1)                                                     # |terms| of gradV . xdot
    F = np.einsum("bkj,bk->bj", np.abs(f2), np.abs(gr))                              # |terms| of (f2' g)_j
    dudg = 0.5 * np.einsum("jq,bkq->bkj", np.abs(Rinv), np.abs(f2))                  # |du_j / dg_k|
    rdu = np.abs(u - uf[None, :]) @ (np.abs(R) + np.abs(R.T)).T
    dv = A + np.einsum("bkj,bj->bk", dudg, F)
    dl = np.einsum("bkj,bj->bk", dudg, rdu)
    den = l + eps
    if mode == 0:
        return Tv / den + 1.0, dv / den[:, None] + (Tv / den ** 2)[:, None] * dl
    return Tv + l, dv + dl
