"""GPU tests of hjbx_value_loss_grad_f32 -- the fused MFMA parameter gradient of the value-learning step (reference
controller/vhjb.py:227-253, 282-284): forward, input gradient, the two residuals and the second-order reverse sweep in closed form.

The reference's own version is jax.grad of a function of jax.grad (JAX not importable here: PARITY UNPINNED).  The checks are
against an INDEPENDENT restatement: float64 torch.autograd double back-prop of the plain loss formulas on the same float32 weights
(rtol 1e-4 on the gradient, the tolerance VERDICT r1 item 3 set), plus size-independent properties (additivity over the batch, bitwise
reproducibility, zero contribution of done / padding samples) at sizes the restatement is not run at."""
import numpy as np
import pytest
import torch

from conftest import SYSTEMS
from q_learning_with_hjb_amd import _abi, _ops
from test_gpu_vhjb import _autograd_losses, controller, states_near_target

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["coop/f32", "pair/f32", "pair/f16x2"])
def impl(request):
    """The implementations of hjbx_value_loss_grad_f32: the cooperative single kernel (float32 MFMA; the default), the round-2 pair of kernels
    in float32 (HJBX_OPT_TRAIN_KERNEL = 1) and the same pair with f16x2 split-operand chains (HJBX_OPT_MLP_ARITHMETIC = 2)."""
    kernel, arithmetic = request.param.split("/")
    pa = _abi.set_option(_abi.OPT_MLP_ARITHMETIC, {"f32": 0, "f16x2": 2}[arithmetic])
    pk = _abi.set_option(_abi.OPT_TRAIN_KERNEL, 1 if kernel == "pair" else 0)
    yield request.param
    _abi.set_option(_abi.OPT_MLP_ARITHMETIC, pa)
    _abi.set_option(_abi.OPT_TRAIN_KERNEL, pk)


def _batch(d, ctl, B, seed, frac=0.6, p_done=0.3):
    xs = states_near_target(d, ctl, B, seed, frac)
    rng = np.random.default_rng(seed + 100)
    dones = torch.as_tensor((rng.uniform(size=B) < p_done).astype(np.float32), device="cuda")
    costs = torch.as_tensor(rng.uniform(0.5, 20, B).astype(np.float32), device="cuda")
    return xs, dones, costs


def _reference_sums_f64(name, ctl32, xs, dones, costs, mode=_abi.RESIDUAL_NORMALISED, dtype=torch.float64):
    """float64 autograd double back-prop of the loss SUMS on the float32 weights -> (g_h list, g_t list, scalars).  dtype=torch.float32: the same
    graph evaluated in float32 by PyTorch (matmuls + the library's float32 residual kernels) -- the yardstick a float32 kernel is held to."""
    kw = dict(fused_param_grad=False) if dtype == torch.float32 else {}
    d64, ctl64 = controller(name, dtype, residual_mode=mode, activation=ctl32.value_function_approximator.activation, **kw)
    with torch.no_grad():
        for w64, w32 in zip(ctl64.value_function_approximator.weights, ctl32.value_function_approximator.weights):
            w64.copy_(w32.to(dtype))
    x64, dn64, c64 = xs.to(dtype), dones.to(dtype), costs.to(dtype)
    params = list(ctl64.value_function_approximator.parameters())
    if mode == _abi.RESIDUAL_NORMALISED:
        h, t = _autograd_losses(ctl64, x64, dn64, c64)          # means: multiply the normalisers back
        n_int, n_done = float((1 - dn64).sum()), float(dn64.sum())
        hs, ts = h * (n_int + ctl64.epsilon), t * (n_done + ctl64.epsilon)
    else:
        hs, hsums = ctl64._hjb_sums(x64, dn64)
        ts, _ = ctl64._termination_sums(x64, dn64, c64)
        n_int, n_done = float(hsums[1]), float(hsums[2])
    g_h = torch.autograd.grad(hs, params, retain_graph=True, allow_unused=True)
    g_t = torch.autograd.grad(ts, params, allow_unused=True)
    z = lambda g, p: torch.zeros_like(p) if g is None else g
    return [z(g, p) for g, p in zip(g_h, params)], [z(g, p) for g, p in zip(g_t, params)], (float(hs), float(ts), n_int, n_done)


def _unpack(flat, n):
    P1, P2, P3 = n * 128, 128 * 128, 128 * 64
    P = P1 + P2 + P3
    f = flat.double().cpu().numpy()
    sets = []
    for s in range(2):
        o = s * P
        sets.append([f[o:o + P1].reshape(n, 128), f[o + P1:o + P1 + P2].reshape(128, 128), f[o + P1 + P2:o + P].reshape(128, 64)])
    return sets[0], sets[1], f[2 * P:]


@pytest.mark.parametrize("B", [1, 33, 256, 1000])
@pytest.mark.parametrize("name", SYSTEMS)
def test_value_loss_grad_vs_f64_autograd(name, B, impl):
    """float32 MFMA kernels vs float64 autograd double back-prop: every gradient matrix to 1e-4 of its largest entry (per element) and
    1e-4 in the Frobenius norm; loss sums to 1e-5; counts exact.  B = 1 / 33 / 1000 exercise padding lanes and partial tiles."""
    d, ctl = controller(name)
    vf = ctl.value_function_approximator
    with torch.no_grad():
        for w in vf.weights:
            w.mul_(1.3)
    xs, dones, costs = _batch(d, ctl, B, 31)
    flat = _ops.value_loss_grad(d.system, ctl._task, vf.descriptor(), xs, costs, dones)
    torch.cuda.synchronize()
    gh, gt, sc = _unpack(flat, d.state_dim)
    rh, rt, rsc = _reference_sums_f64(name, ctl, xs, dones, costs)
    assert sc[2] == rsc[2] and sc[3] == rsc[3]
    assert abs(sc[0] - rsc[0]) <= 1e-5 * abs(rsc[0]) + 1e-6 and abs(sc[1] - rsc[1]) <= 1e-5 * abs(rsc[1]) + 1e-6, (sc, rsc)
    for label, got, want in (("hjb", gh, rh), ("termination", gt, rt)):
        for k, (a, b) in enumerate(zip(got, want)):
            b = b.cpu().numpy()
            scale = np.abs(b).max()
            if scale == 0:
                assert np.abs(a).max() == 0
                continue
            err = np.abs(a - b)
            assert err.max() <= 1e-4 * scale, f"{label} dW{k + 1}: max err {err.max():.3e} vs scale {scale:.3e} (rel {err.max() / scale:.2e})"
            assert np.linalg.norm(a - b) <= 1e-4 * np.linalg.norm(b), f"{label} dW{k + 1}: Frobenius rel {np.linalg.norm(a - b) / np.linalg.norm(b):.2e}"
    # printed, not asserted: the kernel's Frobenius error per matrix over that of PyTorch's float32 autograd (rocBLAS matmuls + the library's
    # float32 residual kernels) against the same float64 reference.  For a ReLU network the statistic is heavy-tailed -- one sample whose unit
    # sits at a kink is taken on different sides by different float32 evaluations and then dominates either error (measured 0.4 - 0.9
    # typically, outliers 0.2 - 8 in both directions), and |r| at r = 0 and the clip of u are kinks of the loss itself -- so for this row the
    # ratio is reported and the asserted bounds stay the analytic ones above (1e-4 of each matrix's scale; measured errors: 1e-6 - 4e-6).
    print(f"    {name} B={B}: kernel / PyTorch-float32 Frobenius error per matrix: " + " ".join(f"{r:.2f}" for r in _yardstick_ratios(name, ctl, xs, dones, costs, gh, gt, rh, rt)))


def _yardstick_ratios(name, ctl, xs, dones, costs, gh, gt, rh, rt, mode=_abi.RESIDUAL_NORMALISED):
    """kernel Frobenius error / PyTorch-float32-autograd Frobenius error, per gradient matrix, both against the float64 reference"""
    th, tt, _ = _reference_sums_f64(name, ctl, xs, dones, costs, mode=mode, dtype=torch.float32)
    ratios = []
    for got, want, t32 in ((gh, rh, th), (gt, rt, tt)):
        for a, b, c in zip(got, want, t32):
            b, c = b.cpu().numpy().astype(np.float64), c.cpu().numpy().astype(np.float64)
            if np.abs(b).max() == 0:
                continue
            ratios.append(np.linalg.norm(a - b) / max(np.linalg.norm(c - b), 2.0 ** -23 * np.linalg.norm(b)))
    return ratios


@pytest.mark.parametrize("name", ["cartpole", "quad2d"])
def test_value_loss_grad_raw_residual_mode(name, impl):
    """HJBX_RESIDUAL_RAW (|gradV.xdot + l|, examples/cartpole_balancing.ipynb cell 11) against autograd through the f64 HIP residual op."""
    d, ctl = controller(name, residual_mode=_abi.RESIDUAL_RAW)
    vf = ctl.value_function_approximator
    xs, dones, costs = _batch(d, ctl, 300, 5)
    flat = _ops.value_loss_grad(d.system, ctl._task, vf.descriptor(), xs, costs, dones, mode=_abi.RESIDUAL_RAW)
    gh, gt, sc = _unpack(flat, d.state_dim)
    rh, rt, rsc = _reference_sums_f64(name, ctl, xs, dones, costs, mode=_abi.RESIDUAL_RAW)
    assert abs(sc[0] - rsc[0]) <= 1e-5 * abs(rsc[0]) + 1e-6
    for a, b in zip(gh + gt, rh + rt):
        b = b.cpu().numpy()
        assert np.abs(a - b).max() <= 1e-4 * max(np.abs(b).max(), 1e-30)


def test_value_loss_grad_properties_at_scale(impl):
    """B = 2^17 + 77 (more tiles than workgroups, ragged tail): (a) bitwise reproducible, (b) additive over a split of the batch,
    (c) samples marked done contribute nothing to the hjb set and only they contribute to the termination set."""
    d, ctl = controller("cartpole")
    vf = ctl.value_function_approximator
    B = (1 << 17) + 77
    xs, dones, costs = _batch(d, ctl, B, 9, frac=0.8)
    n = d.state_dim
    P = n * 128 + 128 * 128 + 128 * 64
    call = lambda sl: _ops.value_loss_grad(d.system, ctl._task, vf.descriptor(), xs[sl].contiguous(), costs[sl].contiguous(), dones[sl].contiguous())
    full = call(slice(None))
    again = call(slice(None))
    assert torch.equal(full, again)
    k = 50011
    parts = call(slice(0, k)).double() + call(slice(k, B)).double()
    err = (full.double() - parts).abs()
    scale = torch.stack([full[:P].abs().max(), full[P:2 * P].abs().max()]).double()
    assert float(err[:P].max()) <= 2e-5 * float(scale[0]) and float(err[P:2 * P].max()) <= 2e-5 * float(scale[1])
    assert float(err[2 * P:2 * P + 2].max()) <= 1e-5 * float(full[2 * P:2 * P + 2].abs().max())
    assert torch.equal(full[2 * P + 2:], parts[2 * P + 2:].float())            # the counts are exact
    assert float(full[2 * P + 2] + full[2 * P + 3]) == B
    live = dones == 0
    only_live = _ops.value_loss_grad(d.system, ctl._task, vf.descriptor(), xs[live].contiguous(), costs[live].contiguous(), dones[live].contiguous())
    assert float(only_live[P:2 * P].abs().max()) == 0.0 and float(only_live[2 * P + 1]) == 0.0
    assert float((only_live[:P].double() - full[:P].double()).abs().max()) <= 2e-5 * float(scale[0])


def _mixed_grads(ctl, xs, dones, costs, reg):
    """The gradient the fused update applies: its Adam step rides in the mix kernel (hjbx_mix_adam_f32) and the mixed gradient is never
    materialised, so the tests take it from the same flat buffer beforehand."""
    assert ctl._native_adam
    params = list(ctl.value_function_approximator.parameters())
    mixed, _ = _ops.mix_gradients(ctl.value_loss_gradient(xs, dones, costs), sum(p.numel() for p in params), reg, ctl.epsilon)
    grads, off = [], 0
    for p in params:
        grads.append(mixed[off:off + p.numel()].view_as(p).clone())
        off += p.numel()
    return grads


def test_fused_and_autograd_updates_agree():
    """params_update through the fused kernels == params_update through PyTorch autograd (same float32 data): losses to 1e-5, the
    first Adam step's parameter change to 1e-3 of the step size except where |g| ~ Adam's eps."""
    res = {}
    for fused in (True, False):
        d, ctl = controller("cartpole", fused_param_grad=fused, graph_updates=False)
        assert ctl.fused_param_grad == fused
        xs, dones, costs = _batch(d, ctl, 256, 3)
        params = list(ctl.value_function_approximator.parameters())
        before = [p.detach().clone() for p in params]
        if fused:
            grads = _mixed_grads(ctl, xs, dones, costs, 0.37)
        tot, h, t = ctl.params_update(xs, dones, costs, 0.37)
        if not fused:
            grads = [p.grad.detach().clone() for p in params]
        res[fused] = (float(tot), float(h), float(t), [(p.detach() - b) for p, b in zip(params, before)], grads)
    a, b = res[True], res[False]
    for k in range(3):
        assert abs(a[k] - b[k]) <= 1e-5 * abs(b[k]) + 1e-7
    for ga, gb in zip(a[4], b[4]):
        assert float((ga - gb).abs().max()) <= 1e-4 * float(gb.abs().max())
    for da, db, gb in zip(a[3], b[3], b[4]):
        big = gb.abs() > 1e-5                                    # away from Adam's eps the first step is -lr sign(g)
        assert float((da - db)[big].abs().max()) <= 1e-6


@pytest.mark.parametrize("mode", [_abi.RESIDUAL_NORMALISED, _abi.RESIDUAL_RAW])
@pytest.mark.parametrize("B", [33, 256, 1000])
@pytest.mark.parametrize("name", SYSTEMS)
@pytest.mark.parametrize("activation", ["tanh", "sin"])
def test_value_loss_grad_tanh_network_vs_f64_autograd(activation, name, B, mode):
    """The tanh network of examples/cartpole_balancing.ipynb cell 6 and the sin network of examples/double_integrator_optimal_time.ipynb
    cell 5 in the fused kernel: act'' != 0 adds the second-order terms (tanh: -2 h . d . t; sin: -h . (W d_next) . t) to the reverse sweep
    (derivation at the top of csrc/hjbx_train_coop.hip).  Against float64 torch.autograd double back-prop of the plain loss formulas with
    torch.tanh / torch.sin, same bounds as the ReLU network: 1e-4 of each matrix's largest entry and 1e-4 in Frobenius norm.  Both residual
    modes (the notebooks train with the RAW residual).  sin exists for state dimensions <= 4: the 6-D and 10-D systems keep autograd."""
    d, ctl = controller(name, activation=activation, residual_mode=mode)
    vf = ctl.value_function_approximator
    if activation == "sin" and d.state_dim > 4:
        assert not ctl.fused_param_grad
        xs, dones, costs = _batch(d, ctl, 64, 41)
        with pytest.raises(NotImplementedError):
            _ops.value_loss_grad(d.system, ctl._task, vf.descriptor(), xs, costs, dones, mode=mode)
        return
    assert ctl.fused_param_grad
    with torch.no_grad():
        for w in vf.weights:
            w.mul_(1.5)                                        # part of the units towards saturation: h^2 and the second-order terms matter
    xs, dones, costs = _batch(d, ctl, B, 41)
    flat = _ops.value_loss_grad(d.system, ctl._task, vf.descriptor(), xs, costs, dones, mode=mode)
    gh, gt, sc = _unpack(flat, d.state_dim)
    rh, rt, rsc = _reference_sums_f64(name, ctl, xs, dones, costs, mode=mode)
    assert sc[2] == rsc[2] and sc[3] == rsc[3]
    assert abs(sc[0] - rsc[0]) <= 2e-5 * abs(rsc[0]) + 1e-6 and abs(sc[1] - rsc[1]) <= 2e-5 * abs(rsc[1]) + 1e-6, (sc, rsc)
    for label, got, want in (("hjb", gh, rh), ("termination", gt, rt)):
        for k, (a, b) in enumerate(zip(got, want)):
            b = b.cpu().numpy()
            scale = np.abs(b).max()
            if scale == 0:
                assert np.abs(a).max() == 0
                continue
            err = np.abs(a - b)
            assert err.max() <= 1e-4 * scale, f"{activation} {label} dW{k + 1}: max err {err.max():.3e} vs scale {scale:.3e} (rel {err.max() / scale:.2e})"
            assert np.linalg.norm(a - b) <= 1e-4 * np.linalg.norm(b), f"{activation} {label} dW{k + 1}: Frobenius rel {np.linalg.norm(a - b) / np.linalg.norm(b):.2e}"
    # (printed like in test_value_loss_grad_vs_f64_autograd: the loss itself has kinks -- |r| at r = 0, the clip of u -- so single samples can
    # dominate either float32 evaluation's error with a smooth activation too: measured 0.1 - 1.2 typically, 3.5 once)
    ratios = _yardstick_ratios(name, ctl, xs, dones, costs, gh, gt, rh, rt, mode=mode)
    print(f"    {activation} {name} B={B}: kernel / PyTorch-float32 Frobenius error per matrix: " + " ".join(f"{r:.2f}" for r in ratios))
    # the second-order terms are really there: dropping them (= treating tanh like a piecewise-linear unit) would be far outside the bound
    if B == 256 and mode == _abi.RESIDUAL_NORMALISED:
        assert np.abs(rh[1].cpu().numpy()).max() > 0


@pytest.mark.parametrize("activation", ["tanh", "sin"])
def test_tanh_updates_through_the_fused_kernels_match_autograd(activation):
    """params_update of a tanh controller (the notebook recipe that reproduces the reference's cartpole anchor; likewise sin) through the fused
    kernels == through PyTorch autograd: losses to 1e-5, gradients to 1e-4 of each matrix's largest entry."""
    res = {}
    for fused in (True, False):
        d, ctl = controller("cartpole", activation=activation, fused_param_grad=fused, graph_updates=False, residual_mode=_abi.RESIDUAL_RAW)
        assert ctl.fused_param_grad == fused
        xs, dones, costs = _batch(d, ctl, 256, 3)
        grads = _mixed_grads(ctl, xs, dones, costs, 0.37) if fused else None
        tot, h, t = ctl.params_update(xs, dones, costs, 0.37)
        res[fused] = (float(tot), float(h), float(t), grads or [p.grad.detach().clone() for p in ctl.value_function_approximator.parameters()])
    a, b = res[True], res[False]
    for k in range(3):
        assert abs(a[k] - b[k]) <= 1e-5 * abs(b[k]) + 1e-7
    for ga, gb in zip(a[3], b[3]):
        assert float((ga - gb).abs().max()) <= 1e-4 * float(gb.abs().max())


def test_value_loss_grad_rejects_what_it_cannot_do():
    d, ctl = controller("nearhover", activation="sin")          # sin: state dimensions <= 4 only
    xs, dones, costs = _batch(d, ctl, 64, 1)
    with pytest.raises(NotImplementedError):
        _ops.value_loss_grad(d.system, ctl._task, ctl.value_function_approximator.descriptor(), xs, costs, dones)
    assert not ctl.fused_param_grad                              # the controller falls back to autograd there on its own
    with pytest.raises(NotImplementedError):
        controller("nearhover", activation="sin", fused_param_grad=True)
    d, ctl = controller("cartpole", activation="tanh")
    xs, dones, costs = _batch(d, ctl, 64, 1)
    prev = _abi.set_option(_abi.OPT_TRAIN_KERNEL, 1)             # tanh exists in the cooperative kernel only: the option does not apply to it
    try:
        flat = _ops.value_loss_grad(d.system, ctl._task, ctl.value_function_approximator.descriptor(), xs, costs, dones)
        assert torch.isfinite(flat).all()
    finally:
        _abi.set_option(_abi.OPT_TRAIN_KERNEL, prev)
    d, ctl = controller("cartpole")
    flat = _ops.value_loss_grad(d.system, ctl._task, ctl.value_function_approximator.descriptor(), xs[:0].contiguous(), costs[:0].contiguous(),
                                dones[:0].contiguous())
    assert float(flat.abs().max()) == 0.0


def test_mix_gradients_kernel_vs_formula():
    """hjbx_mix_gradients_f32 == g_h / (#interior + eps) + reg g_t / (#done + eps) and the three losses (vhjb.py:241, 253, 284-288), with the
    regularisation weight as a host value and as a device scalar (the hipGraph form)."""
    from q_learning_with_hjb_amd.controller.vhjb import mix_flat
    gen = torch.Generator(device="cuda").manual_seed(4)
    P = 4 * 128 + 128 * 128 + 128 * 64
    flat = torch.randn(2 * P + 4, generator=gen, device="cuda")
    flat[2 * P:] = torch.tensor([37.5, 12.25, 187.0, 69.0], device="cuda")
    shapes = [torch.empty(4, 128), torch.empty(128, 128), torch.empty(128, 64)]
    want, wh, wt = mix_flat(flat.double(), shapes, 0.37, 1e-10)
    for reg in (0.37, torch.tensor(0.37, device="cuda")):
        mixed, losses = _ops.mix_gradients(flat, P, reg, 1e-10)
        got = mixed.double()
        ref = torch.cat([w.reshape(-1) for w in want])
        assert float((got - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
        assert abs(float(losses[1]) - float(wh)) < 1e-6 * abs(float(wh)) and abs(float(losses[2]) - float(wt)) < 1e-6 * abs(float(wt))
        assert abs(float(losses[0]) - float(wh + 0.37 * wt)) < 1e-6 * abs(float(wh + 0.37 * wt))


def test_mix_adam_follows_torch_adam_step_for_step():
    """hjbx_mix_adam_f32 = hjbx_mix_gradients_f32 + optax.adam (vhjb.py:120, 262-263) in one launch: against torch.optim.Adam on the mixed
    gradient of the same flat buffers, ten steps from a non-trivial state; moments to float32 rounding, weights to a few ulp of the step."""
    g = torch.Generator(device="cuda").manual_seed(9)
    shapes = [(4, 128), (128, 128), (128, 64)]
    P = sum(a * b for a, b in shapes)
    ours = [torch.randn(sh, generator=g, device="cuda") * 0.1 for sh in shapes]
    theirs = [torch.nn.Parameter(w.clone()) for w in ours]
    lr, b1, b2, eps_adam, eps = 3e-3, 0.9, 0.999, 1e-8, 1e-7
    opt = torch.optim.Adam(theirs, lr=lr, betas=(b1, b2), eps=eps_adam)
    m = [torch.zeros_like(w) for w in ours]
    v = [torch.zeros_like(w) for w in ours]
    steps = [torch.zeros((), device="cuda") for _ in ours]
    ticket = torch.zeros((1,), dtype=torch.int32, device="cuda")
    acc, counter = torch.zeros(3, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    want_acc = torch.zeros(3, device="cuda")
    for it in range(10):
        flat = torch.randn((2 * P + 4,), generator=g, device="cuda") * (10.0 ** (it % 3 - 1))
        flat[2 * P + 2], flat[2 * P + 3] = 200.0 + it, 56.0 - it
        reg = 0.01 * (it + 1)
        mixed, losses = _ops.mix_gradients(flat, P, reg, eps)
        off = 0
        for p in theirs:
            p.grad = mixed[off:off + p.numel()].view_as(p).clone()
            off += p.numel()
        opt.step()
        got = _ops.mix_adam(flat, reg, eps, ours, m, v, steps, ticket, lr, b1, b2, eps_adam, loss_accum=acc, step_counter=counter)
        want_acc += losses
        assert torch.equal(got, losses) and int(ticket) == 0 and all(float(k) == it + 1 for k in steps)
        for w, p, mm, vv in zip(ours, theirs, m, v):
            st = opt.state[p]
            assert torch.allclose(mm, st["exp_avg"], rtol=1e-5, atol=1e-9) and torch.allclose(vv, st["exp_avg_sq"], rtol=1e-5, atol=1e-12)
            assert (w - p).abs().max().item() <= 2e-6 * lr * (it + 1) + 1e-7 * p.abs().max().item()
    assert int(counter) == 10 and torch.allclose(acc, want_acc, rtol=1e-6)


@pytest.mark.parametrize("name,activation", [("cartpole", "relu"), ("quad2d", "tanh"), ("nearhover", "relu"), ("linear", "sin")])
def test_value_loss_adam_equals_gradient_then_mix_adam(name, activation, impl):
    """hjbx_value_loss_adam_f32 (params_update in one call: on the cooperative path the reduction of the partial sums, the mix and the Adam
    step share one epilogue kernel and the flat buffer is never written) == hjbx_value_loss_grad_f32 followed by hjbx_mix_adam_f32: the same
    sums in the same order, so weights, moments, step counts, losses and the device-side counters agree bit for bit; three steps."""
    if activation != "relu" and impl != "coop/f32":
        pytest.skip("tanh / sin exist in the cooperative kernel only")
    d, ctl = controller(name, activation=activation)
    vf = ctl.value_function_approximator
    xs, dones, costs = _batch(d, ctl, 300, 13)
    state = []
    for fused in (True, False):
        g = torch.Generator(device="cuda").manual_seed(2)
        w = [p.detach().clone().contiguous() for p in vf.parameters()]
        m = [torch.rand(p.shape, generator=g, device="cuda") * 1e-3 for p in w]
        v = [torch.rand(p.shape, generator=g, device="cuda") * 1e-5 for p in w]
        steps = [torch.full((), 4.0, device="cuda") for _ in w]
        ticket, acc, counter = torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(3, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
        desc = vf.descriptor()
        desc.W1, desc.W2, desc.W3 = (t.data_ptr() for t in w)
        out = []
        for it in range(3):
            reg = 0.2 * (it + 1)
            if fused:
                losses = _ops.value_loss_adam(d.system, ctl._task, desc, xs, costs, dones, ctl.residual_mode, reg, ctl.epsilon, w, m, v, steps, ticket, 1e-3, 0.9,
                                              0.999, 1e-8, acc, counter)
            else:
                flat = _ops.value_loss_grad(d.system, ctl._task, desc, xs, costs, dones, mode=ctl.residual_mode)
                losses = _ops.mix_adam(flat, reg, ctl.epsilon, w, m, v, steps, ticket, 1e-3, 0.9, 0.999, 1e-8, acc, counter)
            out.append(losses.clone())
        torch.cuda.synchronize()
        state.append((w, m, v, steps, acc, counter, out, ticket))
    a, b = state
    assert int(a[5]) == int(b[5]) == 3 and int(a[7]) == int(b[7]) == 0 and all(float(x) == float(y) == 7.0 for x, y in zip(a[3], b[3]))
    for k in (0, 1, 2):
        for ta, tb in zip(a[k], b[k]):
            assert torch.equal(ta, tb), (k, float((ta - tb).abs().max()))
    assert torch.equal(a[4], b[4]) and all(torch.equal(x, y) for x, y in zip(a[6], b[6]))
    assert all(float((p - q).abs().max()) > 0 for p, q in zip(a[0], vf.parameters()))              # the weights did move
