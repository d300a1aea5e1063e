#!/usr/bin/env python3
"""PCIe-inclusive rate of the numpy-in / numpy-out flavour of the Python surface (DESIGN.md section 7 note; never the bench value):
Dynamics.simulate on host arrays = H2D of x and u, one kernel, D2H of x'."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from q_learning_with_hjb_amd.configs import defaults as D
from q_learning_with_hjb_amd.dynamics.cartpole import Cartpole

d = Cartpole(D.cartpole_dynamics_config())
B = 1 << 20
rng = np.random.default_rng(0)
x = (d.x0_mean + rng.uniform(-1, 1, (B, 4)) * d.x0_std).astype(np.float32)
u = rng.uniform(-10, 10, (B, 1)).astype(np.float32)
for _ in range(3):
    y = d.simulate(x, u)
t0 = time.perf_counter()
for _ in range(10):
    y = d.simulate(x, u)
host = (time.perf_counter() - t0) / 10
xd, ud = torch.as_tensor(x, device="cuda"), torch.as_tensor(u, device="cuda")
for _ in range(3):
    d.simulate(xd, ud)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100):
    d.simulate(xd, ud)
torch.cuda.synchronize()
dev = (time.perf_counter() - t0) / 100
print(f"Cartpole.simulate, B=2^20 f32: host arrays in/out {host*1e3:.2f} ms per call = {B/host:.3e} env-steps/s "
      f"({(x.nbytes + u.nbytes + y.nbytes)/host/1e9:.1f} GB/s over PCIe incl. pageable staging); device tensors {dev*1e6:.1f} us = {B/dev:.3e} env-steps/s")
