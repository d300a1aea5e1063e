"""Development aid: time hjbx_value_loss_grad_f32 in its three implementations (cooperative single kernel; the round-2 pair in f32 and with
f16x2 chains) at a full batch and at the reference's minibatch, and params_update at 256 under hipGraph replay.
    python tools/dev/time_train.py [system] [B]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from q_learning_with_hjb_amd import _abi, _ops

name = sys.argv[1] if len(sys.argv) > 1 else "nearhover"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
for label, kern, arith in (("coop/f32", 0, 0), ("pair/f32", 1, 0), ("pair/f16x2", 1, 2)):
    _abi.set_option(_abi.OPT_TRAIN_KERNEL, kern)
    _abi.set_option(_abi.OPT_MLP_ARITHMETIC, arith)
    bench.ARITHMETIC = {0: "f32", 2: "f16x2"}[arith]
    for b in (B, 256):
        r = bench.param_gradient_kernels(name, b)
        print(label, b, json.dumps({k: r[k] for k in ("ms", "samples_per_s", "frac")}), flush=True)
    _ops.release_workspaces()
    torch.cuda.empty_cache()
for label, kern, arith in (("coop/f32", 0, 0), ("pair/f32", 1, 0), ("pair/f16x2", 1, 2)):
    _abi.set_option(_abi.OPT_TRAIN_KERNEL, kern)
    _abi.set_option(_abi.OPT_MLP_ARITHMETIC, arith)
    r = bench.optimiser_step(1, None)
    print(label, "params_update 256", json.dumps({k: r[k] for k in ("ms_per_update", "updates_per_s", "gradient", "mode", "fit_phase")}), flush=True)
