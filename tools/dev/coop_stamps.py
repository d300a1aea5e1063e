"""Development aid: where the cooperative gradient kernel spends its time at the reference's minibatch (a library built with
-DHJBX_COOP_STAMPS records 100 MHz wall-clock stamps of workgroup 1 at every barrier).
    HJBX_LIBRARY=build/dev/libhjbx_stamps.so python tools/dev/coop_stamps.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from q_learning_with_hjb_amd import _abi, _ops  # noqa: E402
from test_gpu_train import _batch  # noqa: E402
from test_gpu_vhjb import controller  # noqa: E402

d, ctl = controller("cartpole")
vf = ctl.value_function_approximator
xs, dones, costs = _batch(d, ctl, 256, 3)
for _ in range(5):
    flat = _ops.value_loss_grad(d.system, ctl._task, vf.descriptor(), xs, costs, dones)
torch.cuda.synchronize()
ws = list(_ops._tws.values())[0]
grid, n = 32, 4
partial = (grid * 48 * 1024 * 4 + 255) // 256 * 256
rec = ws[partial + 1 * 2 * (2 * n * 128) * 4:][:64 * 8].cpu().numpy().view(np.uint64)
cnt = int(rec[63])
t = rec[:cnt].astype(np.int64)
names = ["entry", "LDS filled"] + [f"barrier {c}" for c in "ABCDEFGIJKLMNOP"] + ["loop-exit barrier", "dW1 tail done", "partial sums stored"]
print("stamps:", cnt)
for i in range(cnt):
    print(f"{names[i] if i < len(names) else i:22s} +{(t[i] - t[0]) * 10:7d} ns   (step {(t[i] - t[i - 1]) * 10 if i else 0:6d} ns)")
