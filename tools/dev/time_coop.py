"""Development aid: time the cooperative parameter-gradient kernel (cartpole, the library named by HJBX_LIBRARY) at B = 2^20 and 256."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
for name, b in (("cartpole", 1 << 20), ("nearhover", 1 << 20), ("cartpole", 256), ("nearhover", 256)):
    r = bench.param_gradient_kernels(name, b)
    print(os.environ.get("HJBX_LIBRARY", "default"), name, b, json.dumps({k: r[k] for k in ("ms", "samples_per_s", "frac")}), flush=True)
