#!/usr/bin/env python3
"""Diagnostic (never shipped): in-kernel shader clock of k_value_grad_mfma.
Build with HJBX_EXTRA_FLAGS=-DHJBX_DIAG_CLOCK, then run this on the GPU box. clock = d(s_memtime)/d(s_memrealtime) x 100 MHz."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from q_learning_with_hjb_amd import _ops
from q_learning_with_hjb_amd.configs import defaults as D
from q_learning_with_hjb_amd.controller.vhjb import VHJBController
from q_learning_with_hjb_amd.dynamics.cartpole import Cartpole

dyn = Cartpole(D.cartpole_dynamics_config())
ctl = VHJBController(dyn, D.cartpole_vhjb_config())
ctl.value_function_approximator.load_quadratic(ctl.P, noise=0.05)
B = 1 << 20
x = dyn.get_initial_state(B, generator=torch.Generator(device="cuda").manual_seed(0))
vf = ctl.value_function_approximator
t_end = time.time() + 2.0
while time.time() < t_end:                      # >= 2 s of back-to-back launches so DVFS settles
    for _ in range(50):
        V, g = vf.fused_value_grad(x)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    V, g = vf.fused_value_grad(x)
e1.record()
torch.cuda.synchronize()
print(f"event-timed launch-to-launch, same x: {e0.elapsed_time(e1)/50*1e3:.1f} us per launch")
xs_ring = [x + 1e-3 * k for k in range(48)]      # 48 x 16 MB = 768 MB > Infinity Cache
torch.cuda.synchronize()
e0.record()
for k in range(96):
    V, g = vf.fused_value_grad(xs_ring[k % 48])
e1.record()
torch.cuda.synchronize()
print(f"event-timed launch-to-launch, 48 different x buffers: {e0.elapsed_time(e1)/96*1e3:.1f} us per launch")
dn = torch.full((B,), -1, dtype=torch.int32, device="cuda"); c = torch.empty(B, device="cuda"); d = torch.empty(B, device="cuda")
xn = torch.empty_like(x)
e0.record()
for k in range(96):
    V, g = vf.fused_value_grad(xs_ring[k % 48])
    _ops.vhjb_step(dyn.system, ctl._task, k, 1 << 30, xs_ring[k % 48], g, xn, c, d, dn)
e1.record()
torch.cuda.synchronize()
print(f"event-timed value_grad + vhjb_step pair: {e0.elapsed_time(e1)/96*1e3:.1f} us per pair")
NW = int(os.environ.get("NW", "8"))
st = V[: 4 * 256 * NW].view(-1, 4).cpu()
cyc, rt = st[:, 0].double(), st[:, 1].double()
ok = rt > 0
clk = (cyc[ok] / rt[ok] * 100e6)
print(f"waves {int(ok.sum())}: shader cycles median {cyc[ok].median():.0f}, wall {rt[ok].median()/100:.1f} us, clock median {clk.median()/1e9:.3f} GHz "
      f"(min {clk.min()/1e9:.3f}, max {clk.max()/1e9:.3f})")
fill, start = st[:, 2].double(), st[:, 3].double()
print(f"LDS fill (entry -> loop start): median {fill.median()/100:.1f} us, max {fill.max()/100:.1f} us; "
      f"kernel-entry skew across waves: {(start.max()-start.min())/100:.1f} us; last wave end - first entry: "
      f"{((start+fill+rt).max()-start.min())/100:.1f} us")
print(f"MFMA cycles per wave = 128 tiles per CU x 784 MFMAs x 64 cycles; pipe share per SIMD = {128/4*784*64/cyc[ok].median():.3f}")
